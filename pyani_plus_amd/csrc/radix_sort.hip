// radix_sort.hip -- stable LSD radix sort of (u64 key, u32 value) pairs and an
// exclusive scan, the two device primitives the sketch and dictionary phases
// share (gfx950).
//
// There is no reference analogue: sourmash sorts each sketch on the CPU
// (`singlesketch`, call site pyani_plus/methods/sourmash.py:67-83) and
// `manysearch` merges sorted lists pairwise (sourmash.py:184-200).  Here one
// global sort orders every (hash, genome) posting at once.
//
// Pass structure (8-bit digits): per-tile digit histogram -> exclusive scan of
// the [digit][tile] table -> stable scatter.  Inside the scatter each wave
// owns a contiguous slice of the tile and ranks its elements with a
// wavefront multi-split: 8 ballots build the mask of lanes holding the same
// digit, popcounts below the lane give the in-wave rank, and a per-wave LDS
// counter row carries the running count from one 64-element row to the next.
#include "pa_internal.h"

namespace {

constexpr int kSortThreads = 256;
constexpr int kWaves = kSortThreads / 64;
constexpr int kItems = 8;                         // 64-element rows per wave
constexpr int kTile = kSortThreads * kItems;      // 2048 elements per workgroup
constexpr int kRadix = 256;

__device__ __forceinline__ uint32_t digit_of(uint64_t key, int shift) { return (uint32_t)(key >> shift) & 0xffu; }
__device__ __forceinline__ uint32_t wave_incl_scan_rs(uint32_t v, uint32_t lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = __shfl_up(v, o, 64);
    if (lane >= (uint32_t)o) v += t;
  }
  return v;
}

// BY_VAL: the digit comes from the 32-bit value instead of the 64-bit key
template <bool BY_VAL>
__global__ __launch_bounds__(kSortThreads) void rs_hist_kernel(const uint64_t *__restrict__ keys,
                                                               const uint32_t *__restrict__ vals, uint64_t n,
                                                               int shift, uint32_t n_tiles,
                                                               uint32_t *__restrict__ hist /*[256][n_tiles]*/) {
  __shared__ uint32_t s_h[kRadix];
  const uint32_t tid = threadIdx.x;
  s_h[tid] = 0;
  __syncthreads();
  const uint64_t base = (uint64_t)blockIdx.x * kTile;
#pragma unroll
  for (int it = 0; it < kItems; ++it) {
    const uint64_t i = base + (uint64_t)it * kSortThreads + tid;
    if (i < n) atomicAdd(&s_h[BY_VAL ? ((vals[i] >> shift) & 0xffu) : digit_of(keys[i], shift)], 1u);
  }
  __syncthreads();
  hist[(uint64_t)tid * n_tiles + blockIdx.x] = s_h[tid];
}

template <bool BY_VAL>
__global__ __launch_bounds__(kSortThreads) void rs_scatter_kernel(
    const uint64_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in, uint64_t *__restrict__ keys_out,
    uint32_t *__restrict__ vals_out, uint64_t n, int shift, uint32_t n_tiles,
    const uint32_t *__restrict__ offs /*[256][n_tiles] exclusive*/) {
  __shared__ uint32_t s_cnt[kWaves][kRadix];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
#pragma unroll
  for (int w = 0; w < kWaves; ++w) s_cnt[w][tid] = 0;
  __syncthreads();

  // wave `wave` owns elements [base + wave*kItems*64, +kItems*64), row-major rows of 64
  const uint64_t wbase = (uint64_t)blockIdx.x * kTile + (uint64_t)wave * (kItems * 64);
  uint64_t key[kItems];
  uint32_t val[kItems];
  uint32_t rank[kItems];
  const uint64_t lt_mask = (lane == 0) ? 0ULL : (~0ULL >> (64 - lane));
#pragma unroll
  for (int it = 0; it < kItems; ++it) {
    const uint64_t i = wbase + (uint64_t)it * 64 + lane;
    const bool live = i < n;
    key[it] = live ? keys_in[i] : ~0ULL;
    val[it] = live ? vals_in[i] : ~0u;
    const uint32_t d = BY_VAL ? ((val[it] >> shift) & 0xffu) : digit_of(key[it], shift);
    // lanes with the same digit (dead lanes only match dead lanes)
    uint64_t peers = __ballot(live);
    if (!live) peers = ~peers;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const bool bit = (d >> b) & 1u;
      const uint64_t bal = __ballot(bit);
      peers &= bit ? bal : ~bal;
    }
    const uint32_t below = __popcll(peers & lt_mask);
    const uint32_t pre = s_cnt[wave][d];
    __builtin_amdgcn_wave_barrier();
    if (live && below == 0) s_cnt[wave][d] = pre + __popcll(peers);
    __builtin_amdgcn_wave_barrier();
    rank[it] = pre + below;
  }
  __syncthreads();
  // The tile is put in order in LDS first and leaves from there: thread t writes the elements t, t + 256, ... of the
  // ordered tile, so the elements of one digit go to consecutive addresses from consecutive lanes (runs of 8 on
  // average: whole 64-byte sectors of keys) instead of one 8-byte store per element wherever its wave's rank put it.
  __shared__ uint64_t s_key[kTile];
  __shared__ uint32_t s_val[kTile];
  __shared__ uint32_t s_delta[kRadix];  // per digit: global start of the tile's run minus its start inside the tile
  __shared__ uint32_t s_w[kWaves];
  {
    // thread = digit: the digit's elements in the tile, the waves' slices inside them, the run's start inside the tile
    uint32_t wave_cnt[kWaves], total = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) { wave_cnt[w] = s_cnt[w][tid]; total += wave_cnt[w]; }
    const uint32_t inc = wave_incl_scan_rs(total, lane);
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    uint32_t local = inc - total;
#pragma unroll
    for (int w = 0; w < kWaves; ++w)
      if ((uint32_t)w < wave) local += s_w[w];
    s_delta[tid] = offs[(uint64_t)tid * n_tiles + blockIdx.x] - local;
    uint32_t run = local;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) { s_cnt[w][tid] = run; run += wave_cnt[w]; }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < kItems; ++it) {
    const uint64_t i = wbase + (uint64_t)it * 64 + lane;
    if (i < n) {
      const uint32_t d = BY_VAL ? ((val[it] >> shift) & 0xffu) : digit_of(key[it], shift);
      const uint32_t at = s_cnt[wave][d] + rank[it];  // position inside the ordered tile
      s_key[at] = key[it];
      s_val[at] = val[it];
    }
  }
  __syncthreads();
  const uint64_t tile_base = (uint64_t)blockIdx.x * kTile;
  const uint32_t live = (uint32_t)(n - tile_base < (uint64_t)kTile ? n - tile_base : (uint64_t)kTile);
#pragma unroll
  for (int it = 0; it < kItems; ++it) {
    const uint32_t at = (uint32_t)it * kSortThreads + tid;
    if (at < live) {
      const uint64_t k = s_key[at];
      const uint32_t v = s_val[at];
      const uint32_t d = BY_VAL ? ((v >> shift) & 0xffu) : digit_of(k, shift);
      const uint32_t dst = s_delta[d] + at;
      keys_out[dst] = k;
      vals_out[dst] = v;
    }
  }
}

// ---- exclusive scan (three-kernel: tile sums -> scan of sums -> apply) -------
constexpr int kScanThreads = 256;
constexpr int kScanItems = 8;
constexpr int kScanTile = kScanThreads * kScanItems;  // 2048

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, uint32_t lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = __shfl_up(v, o, 64);
    if (lane >= (uint32_t)o) v += t;
  }
  return v;
}

// block-wide exclusive scan of one value per thread; returns exclusive prefix, total in *total
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *s_w /*[4]*/, uint32_t *total) {
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint32_t inc = wave_incl_scan(v, lane);
  if (lane == 63) s_w[wave] = inc;
  __syncthreads();
  uint32_t wpre = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < kScanThreads / 64; ++w) {
    const uint32_t x = s_w[w];
    if ((uint32_t)w < wave) wpre += x;
    tot += x;
  }
  __syncthreads();
  *total = tot;
  return wpre + inc - v;
}

__global__ __launch_bounds__(kScanThreads) void scan_tile_sums_kernel(const uint32_t *__restrict__ in, uint64_t n,
                                                                      uint32_t *__restrict__ tile_sums) {
  __shared__ uint32_t s_w[kScanThreads / 64];
  const uint64_t base = (uint64_t)blockIdx.x * kScanTile + (uint64_t)threadIdx.x * kScanItems;
  uint32_t s = 0;
#pragma unroll
  for (int j = 0; j < kScanItems; ++j)
    if (base + j < n) s += in[base + j];
  uint32_t tot;
  (void)block_excl_scan(s, s_w, &tot);
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = tot;
}

// single workgroup: exclusive scan of m values in place (u32), total (u64) out
__global__ __launch_bounds__(kScanThreads) void scan_sums_kernel(uint32_t *__restrict__ sums, uint32_t m,
                                                                 unsigned long long *__restrict__ total_out) {
  __shared__ uint32_t s_w[kScanThreads / 64];
  uint64_t carry = 0;
  for (uint32_t base = 0; base < m; base += kScanThreads) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t v = i < m ? sums[i] : 0u;
    uint32_t tot;
    const uint32_t ex = block_excl_scan(v, s_w, &tot);
    if (i < m) sums[i] = (uint32_t)(carry + ex);
    carry += tot;
  }
  if (threadIdx.x == 0 && total_out) *total_out = carry;
}

__global__ __launch_bounds__(kScanThreads) void scan_apply_kernel(const uint32_t *__restrict__ in,
                                                                  uint32_t *__restrict__ out, uint64_t n,
                                                                  const uint32_t *__restrict__ tile_pre) {
  __shared__ uint32_t s_w[kScanThreads / 64];
  const uint64_t base = (uint64_t)blockIdx.x * kScanTile + (uint64_t)threadIdx.x * kScanItems;
  uint32_t v[kScanItems];
  uint32_t s = 0;
#pragma unroll
  for (int j = 0; j < kScanItems; ++j) {
    v[j] = (base + j < n) ? in[base + j] : 0u;
    s += v[j];
  }
  uint32_t tot;
  uint32_t ex = block_excl_scan(s, s_w, &tot) + tile_pre[blockIdx.x];
#pragma unroll
  for (int j = 0; j < kScanItems; ++j) {
    if (base + j < n) out[base + j] = ex;
    ex += v[j];
  }
}

}  // namespace

int pa_exclusive_scan_u32(pa_ctx *c, const uint32_t *d_in, uint32_t *d_out, uint64_t n, uint64_t *d_total_u64) {
  if (n == 0) {
    if (d_total_u64) PA_HIP(hipMemsetAsync(d_total_u64, 0, sizeof(uint64_t), c->stream));
    return PA_OK;
  }
  const uint32_t tiles = ceil_div_u64(n, kScanTile);
  PA_TRY(c->scan_tmp.reserve((uint64_t)tiles * sizeof(uint32_t)));
  uint32_t *d_sums = c->scan_tmp.as<uint32_t>();
  hipLaunchKernelGGL(scan_tile_sums_kernel, dim3(tiles), dim3(kScanThreads), 0, c->stream, d_in, n, d_sums);
  hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(kScanThreads), 0, c->stream, d_sums, tiles,
                     reinterpret_cast<unsigned long long *>(d_total_u64));
  hipLaunchKernelGGL(scan_apply_kernel, dim3(tiles), dim3(kScanThreads), 0, c->stream, d_in, d_out, n, d_sums);
  PA_HIP(hipGetLastError());
  return PA_OK;
}

int pa_radix_sort_pairs(pa_ctx *c, uint64_t *keys[2], uint32_t *vals[2], uint64_t n, int bit_lo, int bit_hi,
                        bool by_val, int *which) {
  if (n <= 1 || bit_hi <= bit_lo) return PA_OK;
  PA_REQUIRE(n < (1ULL << 32), "radix sort: %llu elements exceed the 32-bit index space", (unsigned long long)n);
  const uint32_t tiles = ceil_div_u64(n, kTile);
  const uint64_t hist_n = (uint64_t)kRadix * tiles;
  PA_TRY(c->hist.reserve(hist_n * sizeof(uint32_t)));
  uint32_t *d_hist = c->hist.as<uint32_t>();
  int cur = *which;
  for (int shift = bit_lo; shift < bit_hi; shift += 8) {
    if (by_val)
      hipLaunchKernelGGL(rs_hist_kernel<true>, dim3(tiles), dim3(kSortThreads), 0, c->stream, keys[cur], vals[cur],
                         n, shift, tiles, d_hist);
    else
      hipLaunchKernelGGL(rs_hist_kernel<false>, dim3(tiles), dim3(kSortThreads), 0, c->stream, keys[cur], vals[cur],
                         n, shift, tiles, d_hist);
    PA_TRY(pa_exclusive_scan_u32(c, d_hist, d_hist, hist_n, nullptr));
    if (by_val)
      hipLaunchKernelGGL(rs_scatter_kernel<true>, dim3(tiles), dim3(kSortThreads), 0, c->stream, keys[cur],
                         vals[cur], keys[cur ^ 1], vals[cur ^ 1], n, shift, tiles, d_hist);
    else
      hipLaunchKernelGGL(rs_scatter_kernel<false>, dim3(tiles), dim3(kSortThreads), 0, c->stream, keys[cur],
                         vals[cur], keys[cur ^ 1], vals[cur ^ 1], n, shift, tiles, d_hist);
    cur ^= 1;
  }
  PA_HIP(hipGetLastError());
  *which = cur;
  return PA_OK;
}
