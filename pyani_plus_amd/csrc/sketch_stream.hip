// sketch_stream.hip -- host arena -> sketches with the upload hidden behind the hash kernel (gfx950).
//
// The boundary hands over genomes in host memory (the reference reads FASTA files,
// pyani_plus/methods/sourmash.py:67-83); copying 1.9 GB of arena and then hashing it costs 35 + 13 ms at
// N = 1000.  Two things cut that:
//   * the invalid-position mask is a third of the bytes and almost all zeros: it crosses the bus as a
//     list of runs (pa_mask_runs on the host, pa_mask_from_runs on the device);
//   * the packed bases go up in chunks on a copy stream while the hash kernel works on the chunks that
//     have arrived -- the kernel only needs its own blocks and one block of look-back, and with
//     per-genome candidate regions (sketch_lds.hip) its launches are independent of each other.
#include <algorithm>
#include <vector>

#include <cstdlib>

#include "pa_internal.h"

namespace {

constexpr int kThreads = 256;

// one thread per 32-position mask word; runs are sorted and disjoint
__global__ __launch_bounds__(kThreads) void mask_from_runs_kernel(const uint64_t *__restrict__ run_start,
                                                                  const uint64_t *__restrict__ run_len, uint32_t n_runs,
                                                                  uint32_t *__restrict__ mask, uint64_t n_words) {
  const uint64_t w = (uint64_t)blockIdx.x * kThreads + threadIdx.x;
  if (w >= n_words) return;
  const uint64_t lo = w * 32u, hi = lo + 32u;
  // first run that ends after lo
  uint32_t a = 0, b = n_runs;
  while (a < b) {
    const uint32_t mid = (a + b) >> 1;
    if (run_start[mid] + run_len[mid] <= lo) a = mid + 1; else b = mid;
  }
  uint32_t bits = 0;
  for (uint32_t r = a; r < n_runs && run_start[r] < hi; ++r) {
    const uint64_t s = run_start[r] > lo ? run_start[r] : lo;
    const uint64_t e = run_start[r] + run_len[r] < hi ? run_start[r] + run_len[r] : hi;
    if (e > s) {
      const uint32_t n = (uint32_t)(e - s);
      bits |= (n >= 32u ? 0xffffffffu : ((1u << n) - 1u)) << (uint32_t)(s - lo);
    }
  }
  mask[w] = bits;
}

// one thread per 64-block word of the bitmap; a block's mask is one uint2
__global__ __launch_bounds__(kThreads) void dirty_kernel(const uint2 *__restrict__ mask, uint64_t n_blocks64,
                                                         uint64_t *__restrict__ dirty) {
  // one wave per bitmap word: lane l looks at block 64*w + l, a ballot makes the word
  const uint64_t w = ((uint64_t)blockIdx.x * kThreads + threadIdx.x) >> 6;
  const uint64_t t = w * 64u + (threadIdx.x & 63u);
  bool d = false;
  if (t < n_blocks64) {
    const uint2 m = mask[t];
    d = (m.x | m.y) != 0u || t == 0 || mask[t - 1].y != 0u;
  }
  const uint64_t word = __ballot(d);
  if ((threadIdx.x & 63u) == 0 && w * 64u < n_blocks64) dirty[w] = word;
}

}  // namespace

int pa_build_dirty(pa_ctx *c, const uint32_t *d_mask, uint64_t n_blocks64, uint64_t *d_dirty, hipStream_t stream) {
  if (n_blocks64 == 0) return PA_OK;
  const uint64_t words = (n_blocks64 + 63) / 64;
  hipLaunchKernelGGL(dirty_kernel, dim3(ceil_div_u64(words * 64, kThreads)), dim3(kThreads), 0, stream ? stream : c->stream,
                     reinterpret_cast<const uint2 *>(d_mask), n_blocks64, d_dirty);
  PA_HIP(hipGetLastError());
  return PA_OK;
}

int pa_dirty_or_build(pa_ctx *c, const uint32_t *d_mask, uint64_t n_blocks64, const uint64_t *d_dirty, const uint64_t **out) {
  if (d_dirty) { *out = d_dirty; return PA_OK; }
  PA_TRY(c->dirty.reserve(((n_blocks64 + 63) / 64 + 1) * 8));
  PA_TRY(pa_build_dirty(c, d_mask, n_blocks64, c->dirty.as<uint64_t>()));
  *out = c->dirty.as<uint64_t>();
  return PA_OK;
}

extern "C" {

int pa_arena_dirty(pa_ctx *c, const uint32_t *d_mask, uint64_t arena_bases, uint64_t *d_dirty) {
  PA_REQUIRE(c && (arena_bases == 0 || (d_mask && d_dirty)), "pa_arena_dirty: null argument");
  PA_REQUIRE((arena_bases % PA_ALIGN_BASES) == 0, "pa_arena_dirty: arena_bases %llu is not a multiple of %u",
             (unsigned long long)arena_bases, PA_ALIGN_BASES);
  PA_HIP(hipSetDevice(c->device));
  return pa_build_dirty(c, d_mask, arena_bases / PA_ALIGN_BASES, d_dirty);
}

int pa_mask_from_runs(pa_ctx *c, const uint64_t *h_run_start, const uint64_t *h_run_len, uint32_t n_runs,
                      uint32_t *d_mask, uint64_t arena_bases) {
  PA_REQUIRE(c && (n_runs == 0 || (h_run_start && h_run_len)), "pa_mask_from_runs: null argument");
  PA_REQUIRE((arena_bases % PA_ALIGN_BASES) == 0, "pa_mask_from_runs: arena_bases %llu is not a multiple of %u",
             (unsigned long long)arena_bases, PA_ALIGN_BASES);
  if (arena_bases == 0) return PA_OK;
  PA_REQUIRE(d_mask, "pa_mask_from_runs: null mask");
  PA_HIP(hipSetDevice(c->device));
  for (uint32_t r = 0; r < n_runs; ++r)
    PA_REQUIRE(h_run_len[r] > 0 && h_run_start[r] + h_run_len[r] <= arena_bases &&
                   (r == 0 || h_run_start[r] >= h_run_start[r - 1] + h_run_len[r - 1]),
               "pa_mask_from_runs: run %u is empty, overlaps its predecessor or leaves the arena", r);
  PA_TRY(c->scan_tmp.reserve((uint64_t)n_runs * 16 + 16));
  uint64_t *d_start = c->scan_tmp.as<uint64_t>(), *d_len = d_start + n_runs;
  if (n_runs) {
    PA_HIP(hipMemcpyAsync(d_start, h_run_start, (uint64_t)n_runs * 8, hipMemcpyHostToDevice, c->stream));
    PA_HIP(hipMemcpyAsync(d_len, h_run_len, (uint64_t)n_runs * 8, hipMemcpyHostToDevice, c->stream));
  }
  const uint64_t n_words = arena_bases / 32;
  hipLaunchKernelGGL(mask_from_runs_kernel, dim3(ceil_div_u64(n_words, kThreads)), dim3(kThreads), 0, c->stream, d_start,
                     d_len, n_runs, d_mask, n_words);
  PA_HIP(hipGetLastError());
  PA_HIP(hipStreamSynchronize(c->stream));  // the caller's run arrays may go away
  return PA_OK;
}

int pa_sketch_streamed(pa_ctx *c, const uint32_t *h_packed, const uint64_t *h_run_start, const uint64_t *h_run_len,
                       uint32_t n_runs, uint64_t arena_bases, const uint64_t *h_genome_start, uint32_t n_genomes,
                       uint32_t k, uint64_t max_hash, uint32_t *d_packed, uint32_t *d_mask, uint64_t *d_dirty,
                       uint64_t *d_hashes, uint64_t cap_hashes, uint64_t *d_off, uint64_t *h_total) {
  PA_REQUIRE(c && d_off && h_total && h_genome_start, "pa_sketch_streamed: null argument");
  PA_REQUIRE((arena_bases % PA_ALIGN_BASES) == 0, "pa_sketch_streamed: arena_bases %llu is not a multiple of %u",
             (unsigned long long)arena_bases, PA_ALIGN_BASES);
  PA_REQUIRE(arena_bases == 0 || (h_packed && d_packed && d_mask), "pa_sketch_streamed: null arena");
  PA_REQUIRE(k >= 1 && k <= PA_MAX_K, "pa_sketch_streamed: k=%u outside [1,%u]", k, PA_MAX_K);
  PA_REQUIRE(h_genome_start[n_genomes] == arena_bases, "pa_sketch_streamed: genome_start[n] must equal arena_bases");
  PA_HIP(hipSetDevice(c->device));
  *h_total = 0;
  PA_TRY(pa_mask_from_runs(c, h_run_start, h_run_len, n_runs, d_mask, arena_bases));

  const uint64_t n_blocks = arena_bases / PA_ALIGN_BASES;
  if (!d_dirty) {
    PA_TRY(c->dirty.reserve(((n_blocks + 63) / 64 + 1) * 8));
    d_dirty = c->dirty.as<uint64_t>();
  }
  PA_TRY(pa_build_dirty(c, d_mask, n_blocks, d_dirty));
  PA_REQUIRE(n_blocks < (1ULL << 32), "pa_sketch_streamed: arena too large");
  const double frac = (max_hash == UINT64_MAX) ? 1.0 : ((double)max_hash + 1.0) / 18446744073709551616.0;
  std::vector<uint64_t> region_off(n_genomes + 1, 0);
  std::vector<uint32_t> blk(n_genomes + 1);
  uint64_t longest_region = 0;
  for (uint32_t g = 0; g <= n_genomes; ++g) {
    const uint64_t s = h_genome_start[g];
    PA_REQUIRE((s % PA_ALIGN_BASES) == 0 && (g == 0 || s >= h_genome_start[g - 1]) && s <= arena_bases,
               "pa_sketch_streamed: genome_start[%u]=%llu must be an ascending multiple of %u inside the arena", g,
               (unsigned long long)s, PA_ALIGN_BASES);
    blk[g] = (uint32_t)(s / PA_ALIGN_BASES);
    if (g < n_genomes) {
      const uint64_t room = (uint64_t)((double)(h_genome_start[g + 1] - s) * frac * 1.25) + 128;
      longest_region = std::max(longest_region, room);
      region_off[g + 1] = region_off[g] + room;
    }
  }
  const bool overlap = n_genomes > 0 && longest_region <= kLdsSortMax && arena_bases > 0;
  if (!overlap) {  // long genomes / tiny scaled: plain upload, general sketch path
    PA_HIP(hipMemcpyAsync(d_packed, h_packed, arena_bases / 4, hipMemcpyHostToDevice, c->stream));
    return pa_sketch(c, d_packed, d_mask, d_dirty, arena_bases, h_genome_start, n_genomes, k, max_hash, d_hashes, cap_hashes,
                     d_off, h_total);
  }
  if (!c->copy_stream) PA_HIP(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
  PA_TRY(c->genome_blk.reserve((uint64_t)(n_genomes + 1) * sizeof(uint32_t)));
  PA_TRY(c->region_off.reserve((uint64_t)(n_genomes + 1) * sizeof(uint64_t)));
  PA_TRY(c->region_cursor.reserve((uint64_t)n_genomes * sizeof(uint32_t)));
  PA_TRY(c->cand_keys[0].reserve(region_off[n_genomes] * sizeof(uint64_t)));
  PA_HIP(hipMemcpyAsync(c->genome_blk.p, blk.data(), (uint64_t)(n_genomes + 1) * sizeof(uint32_t), hipMemcpyHostToDevice,
                        c->stream));
  PA_HIP(hipMemcpyAsync(c->region_off.p, region_off.data(), (uint64_t)(n_genomes + 1) * sizeof(uint64_t),
                        hipMemcpyHostToDevice, c->stream));
  uint32_t *d_overflow = c->counters.as<uint32_t>() + 12;
  PA_HIP(hipMemsetAsync(c->region_cursor.p, 0, (uint64_t)n_genomes * sizeof(uint32_t), c->stream));
  PA_HIP(hipMemsetAsync(d_overflow, 0, sizeof(uint32_t), c->stream));
  PA_HIP(hipStreamSynchronize(c->stream));  // blk / region_off are stack-owned; the copy stream starts after this

  // 64 MB of packed bases (2.7e8 positions) per chunk: ~1.2 ms on the bus, ~0.7 ms of hashing
  uint64_t chunk_blocks = (64ull << 20) / 16;
  if (const char *v = PA_TOOL_ENV("PA_STREAM_CHUNK_BLOCKS"))  // tests: small chunks, so that windows cross chunk boundaries
    chunk_blocks = std::max<uint64_t>(64, (strtoull(v, nullptr, 10) + 63) / 64 * 64);
  std::vector<hipEvent_t> arrived;
  int status = PA_OK;
  {
    ProfScope prof(c, PA_PROF_KMER_HASH);
    for (uint64_t b0 = 0; b0 < n_blocks && status == PA_OK; b0 += chunk_blocks) {
      const uint64_t b1 = std::min(n_blocks, b0 + chunk_blocks);
      hipEvent_t e;
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { status = PA_E_HIP; pa_set_error("hipEventCreate failed"); break; }
      arrived.push_back(e);
      if (hipMemcpyAsync(d_packed + b0 * 4, h_packed + b0 * 4, (b1 - b0) * 16, hipMemcpyHostToDevice, c->copy_stream) != hipSuccess ||
          hipEventRecord(e, c->copy_stream) != hipSuccess || hipStreamWaitEvent(c->stream, e, 0) != hipSuccess) {
        status = PA_E_HIP;
        pa_set_error("pa_sketch_streamed: chunk upload failed");
        break;
      }
      status = pa_launch_kmer_hash(c, d_packed, d_mask, d_dirty, b1, c->genome_blk.as<uint32_t>(), n_genomes, k, max_hash,
                                   c->cand_keys[0].as<uint64_t>(), nullptr, 0, nullptr, c->region_off.as<uint64_t>(),
                                   c->region_cursor.as<uint32_t>(), d_overflow, b0, c->stream);
    }
  }
  bool overflow = false;
  if (status == PA_OK) {
    ProfScope prof(c, PA_PROF_SKETCH_SORT);
    status = pa_sketch_from_regions(c, c->cand_keys[0].as<uint64_t>(), c->region_off.as<uint64_t>(),
                                    c->region_cursor.as<uint32_t>(), d_overflow, n_genomes, (uint32_t)longest_region,
                                    max_hash, d_hashes, cap_hashes, d_off, h_total, &overflow);
  } else {
    (void)hipStreamSynchronize(c->copy_stream);
    (void)hipStreamSynchronize(c->stream);
  }
  for (hipEvent_t e : arrived) (void)hipEventDestroy(e);
  if (status != PA_OK) return status;
  if (!overflow) return PA_OK;
  // a region overflowed (repeats, low-complexity sequence): the arena is resident now, take the general path
  return pa_sketch(c, d_packed, d_mask, d_dirty, arena_bases, h_genome_start, n_genomes, k, max_hash, d_hashes, cap_hashes,
                   d_off, h_total);
}

}  // extern "C"
