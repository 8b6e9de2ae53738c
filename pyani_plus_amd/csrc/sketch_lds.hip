// sketch_lds.hip -- per-genome candidate regions -> sorted, de-duplicated CSR sketches (gfx950).
//
// Same result as the global radix sort + head flags + scan of sketch_build.hip (the `mins` array of
// `sourmash scripts singlesketch`, pyani_plus/methods/sourmash.py:67-83), for the usual case that every
// genome's survivors fit in LDS: a bacterial genome at scaled=1000 keeps ~5 000 hashes = 40 KB of the
// CU's 160 KB.  kmer_hash.hip drops each genome's survivors, unordered, into that genome's own region
// (capacity = expectation + 25 % + 128); then
//
//   genome_sort_kernel  one workgroup per genome: region -> LDS, bitonic sort padded with ~0 keys to a
//                       power of two, adjacent-duplicate flags, workgroup scan, unique hashes written
//                       back to the front of the region, their number to uniq[g]
//   offsets_kernel      one workgroup: exclusive scan of uniq[] -> CSR offsets (u64) and the total
//   gather_kernel       one workgroup per genome: region front -> hashes[off[g] ..)
//
// Six launches instead of the ~45 of the 9-pass global sort.  Regions that overflow (low-complexity
// genomes, tiny `scaled`) or genomes too long for LDS make pa_sketch fall back to sketch_build.hip.
#include <algorithm>

#include "pa_internal.h"

namespace {

constexpr int kSortThreads = 1024;
constexpr int kSortWaves = kSortThreads / 64;

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, uint32_t lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = __shfl_up(v, o, 64);
    if (lane >= (uint32_t)o) v += t;
  }
  return v;
}

constexpr uint32_t kBuckets = kSortThreads;  // one bucket per thread
constexpr uint32_t kBucketCap = 48;          // longest bucket a single thread sorts by insertion

__global__ __launch_bounds__(kSortThreads) void genome_sort_kernel(uint64_t *__restrict__ regions,
                                                                   const uint64_t *__restrict__ region_off,
                                                                   const uint32_t *__restrict__ cursor,
                                                                   uint32_t *__restrict__ uniq, uint32_t key_shift,
                                                                   uint32_t key_mult) {
  extern __shared__ uint64_t s_key[];
  __shared__ uint32_t s_wave[kSortWaves];
  __shared__ uint32_t s_count[kBuckets], s_cursor[kBuckets];
  __shared__ uint32_t s_longest;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint32_t g = blockIdx.x;
  uint64_t *__restrict__ region = regions + region_off[g];
  const uint32_t room = (uint32_t)(region_off[g + 1] - region_off[g]);
  const uint32_t n = min(cursor[g], room);  // cursor > room only on overflow, which voids this pass anyway
  if (n == 0) {
    if (tid == 0) uniq[g] = 0;
    return;
  }
  // Hashes are spread evenly over [0, max_hash], so 1 024 equal value ranges take ~5 keys each: count, scan,
  // scatter into LDS, and every thread finishes its own range with an insertion sort.  A range with more
  // than kBucketCap keys (sequence that is anything but random) sends the genome to the bitonic network
  // instead, which costs the same for any input.
  auto bucket_of = [&](uint64_t key) -> uint32_t {
    const uint32_t b = (uint32_t)(((key >> key_shift) * (uint64_t)key_mult) >> 32);
    return b < kBuckets ? b : kBuckets - 1u;
  };
  s_count[tid] = 0;
  if (tid == 0) s_longest = 0;
  __syncthreads();
  for (uint32_t i = tid; i < n; i += kSortThreads) atomicAdd(&s_count[bucket_of(region[i])], 1u);
  __syncthreads();
  const uint32_t my_count = s_count[tid];
  {
    const uint32_t incl = wave_incl_scan(my_count, lane);
    if (lane == 63) s_wave[wave] = incl;
    uint32_t longest = my_count;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) longest = max(longest, (uint32_t)__shfl_xor(longest, o, 64));
    if (lane == 0) atomicMax(&s_longest, longest);
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < (int)wave; ++w) base += s_wave[w];
    s_cursor[tid] = base + incl - my_count;
  }
  __syncthreads();
  const uint32_t my_start = s_cursor[tid];
  const bool by_buckets = s_longest <= kBucketCap;
  __syncthreads();  // everyone has read its start before the cursors move
  if (by_buckets) {
    for (uint32_t i = tid; i < n; i += kSortThreads) {
      const uint64_t key = region[i];
      s_key[atomicAdd(&s_cursor[bucket_of(key)], 1u)] = key;
    }
    __syncthreads();
    for (uint32_t i = 1; i < my_count; ++i) {
      const uint64_t key = s_key[my_start + i];
      uint32_t j = i;
      while (j > 0 && s_key[my_start + j - 1] > key) {
        s_key[my_start + j] = s_key[my_start + j - 1];
        --j;
      }
      s_key[my_start + j] = key;
    }
    __syncthreads();
  }
  uint32_t np2 = 2;
  while (np2 < n) np2 <<= 1;
  if (!by_buckets) {
    for (uint32_t i = tid; i < np2; i += kSortThreads) s_key[i] = i < n ? region[i] : ~0ULL;
    __syncthreads();
  }
  // bitonic network; pads equal the largest key, so the first n slots end up holding the n real keys
  for (uint32_t k = 2; !by_buckets && k <= np2; k <<= 1) {
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      for (uint32_t t = tid; t < (np2 >> 1); t += kSortThreads) {
        const uint32_t i = 2u * t - (t & (j - 1u));  // bit j clear
        const uint32_t l = i + j;
        const uint64_t a = s_key[i], b = s_key[l];
        const bool up = (i & k) == 0u;
        if ((a > b) == up) {
          s_key[i] = b;
          s_key[l] = a;
        }
      }
      __syncthreads();
    }
  }
  // unique: thread t owns the contiguous slice [t*per, (t+1)*per)
  const uint32_t per = (n + kSortThreads - 1u) / kSortThreads;
  const uint32_t beg = min(tid * per, n), end = min(beg + per, n);
  uint32_t heads = 0;
  for (uint32_t i = beg; i < end; ++i) heads += (i == 0 || s_key[i] != s_key[i - 1]) ? 1u : 0u;
  const uint32_t incl = wave_incl_scan(heads, lane);
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  uint32_t base = 0, total = 0;
  for (int w = 0; w < kSortWaves; ++w) {
    const uint32_t x = s_wave[w];
    if (w < (int)wave) base += x;
    total += x;
  }
  uint32_t pos = base + incl - heads;
  for (uint32_t i = beg; i < end; ++i) {
    const uint64_t v = s_key[i];
    if (i == 0 || v != s_key[i - 1]) region[pos++] = v;
  }
  if (tid == 0) uniq[g] = total;
}

// exclusive scan of uniq[0..n) into u64 offsets; off[n] and *total get the sum
__global__ __launch_bounds__(kSortThreads) void offsets_kernel(const uint32_t *__restrict__ uniq, uint32_t n,
                                                               uint64_t *__restrict__ off, uint64_t *__restrict__ total) {
  __shared__ uint32_t s_wave[kSortWaves];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  uint64_t carry = 0;
  for (uint32_t i0 = 0; i0 < n; i0 += kSortThreads) {
    const uint32_t i = i0 + tid;
    const uint32_t v = i < n ? uniq[i] : 0u;
    const uint32_t incl = wave_incl_scan(v, lane);
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t base = 0, sum = 0;
    for (int w = 0; w < kSortWaves; ++w) {
      const uint32_t x = s_wave[w];
      if (w < (int)wave) base += x;
      sum += x;
    }
    if (i < n) off[i] = carry + base + incl - v;
    carry += sum;
    __syncthreads();
  }
  if (tid == 0) {
    off[n] = carry;
    *total = carry;
  }
}

__global__ __launch_bounds__(256) void gather_kernel(const uint64_t *__restrict__ regions,
                                                     const uint64_t *__restrict__ region_off,
                                                     const uint64_t *__restrict__ off, uint64_t *__restrict__ hashes) {
  const uint32_t g = blockIdx.x;
  const uint64_t *__restrict__ src = regions + region_off[g];
  const uint64_t o = off[g];
  const uint32_t n = (uint32_t)(off[g + 1] - o);
  for (uint32_t i = threadIdx.x; i < n; i += 256) hashes[o + i] = src[i];
}

}  // namespace

int pa_sketch_from_regions(pa_ctx *c, uint64_t *d_regions, const uint64_t *d_region_off, const uint32_t *d_cursor,
                           const uint32_t *d_overflow, uint32_t n_genomes, uint32_t longest_region, uint64_t max_hash,
                           uint64_t *d_hashes, uint64_t cap_hashes, uint64_t *d_off, uint64_t *h_total,
                           bool *h_overflow) {
  PA_REQUIRE(longest_region <= kLdsSortMax, "LDS sort: region of %u candidates exceeds %u", longest_region, kLdsSortMax);
  uint32_t np2 = 2;
  while (np2 < longest_region) np2 <<= 1;
  const uint32_t lds_bytes = np2 * (uint32_t)sizeof(uint64_t);
  PA_TRY(c->flags.reserve((uint64_t)n_genomes * sizeof(uint32_t)));
  uint32_t *d_uniq = c->flags.as<uint32_t>();
  uint64_t *d_total = c->counters.as<uint64_t>() + 1;
  PA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(genome_sort_kernel),
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  // value ranges of the bucket sort: equal shares of [0, max_hash]
  const int key_bits = max_hash ? 64 - __builtin_clzll(max_hash) : 1;
  const uint32_t key_shift = key_bits > 32 ? (uint32_t)key_bits - 32u : 0u;
  const uint64_t top = (max_hash >> key_shift) + 1;  // (key >> shift) < top <= 2^32
  const uint32_t key_mult = (uint32_t)std::min<uint64_t>(((uint64_t)kBuckets << 32) / top, 0xffffffffull);
  hipLaunchKernelGGL(genome_sort_kernel, dim3(n_genomes), dim3(kSortThreads), lds_bytes, c->stream, d_regions,
                     d_region_off, d_cursor, d_uniq, key_shift, key_mult);
  hipLaunchKernelGGL(offsets_kernel, dim3(1), dim3(kSortThreads), 0, c->stream, d_uniq, n_genomes, d_off, d_total);
  PA_HIP(hipGetLastError());
  // one round trip for both scalars: [0] total, [1] overflow flag
  PA_HIP(hipMemcpyAsync(c->h_pinned, d_total, sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
  PA_HIP(hipMemcpyAsync(c->h_pinned + 1, d_overflow, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  PA_HIP(hipStreamSynchronize(c->stream));
  *h_overflow = (uint32_t)c->h_pinned[1] != 0u;
  if (*h_overflow) return PA_OK;  // caller falls back to the global sort
  *h_total = c->h_pinned[0];
  if (*h_total > cap_hashes) {
    pa_set_error("sketch output needs %llu hashes, caller gave room for %llu", (unsigned long long)*h_total,
                 (unsigned long long)cap_hashes);
    return PA_E_CAPACITY;
  }
  if (*h_total)
    hipLaunchKernelGGL(gather_kernel, dim3(n_genomes), dim3(256), 0, c->stream, d_regions, d_region_off, d_off,
                       d_hashes);
  PA_HIP(hipGetLastError());
  return PA_OK;
}
