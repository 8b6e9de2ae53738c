// bottom_mash.hip -- bottom-m MinHash sketches and the Mash Jaccard -> ANI estimator (gfx950).
//
// BASELINE.json configs[1] and the north-star text name "bottom-m=1000" sketches and a
// "sketch Jaccard -> ANI" step.  The REFERENCE NEVER USES THAT MODE: every fixture `.sig` has
// "num":0 and the only sketch parameter pyani-plus passes is `scaled=N`
// (pyani_plus/methods/sourmash.py:75-76), so this file has no reference call site to replace and its
// parity is UNPINNED -- it is checked against oracle/sourmash_oracle.c's restatement of the published
// Mash estimator only.  It exists so that the mode the baseline names can be measured beside the
// reference's own (scaled) mode.
//
//   pa_sketch_bottom : the scaled pipeline with a threshold sized for ~4m survivors of the shortest
//                      genome, then truncation of every sketch to its m smallest hashes (threshold
//                      raised and the call repeated if a genome came up short)
//   pa_pair_mash     : hashes -> dense ids in hash order (one sort of all postings), then one THREAD per
//                      ordered pair of a (TQ x TS) tile whose TQ+TS id lists sit in LDS (4 bytes per hash,
//                      up to 156 KB of the CU's 160 KB): a two-pointer merge that stops at the m-th union
//                      element.  Lists too long for LDS fall back to one wavefront per pair with a merge
//                      path over the 64-bit lists in HBM.
//   pa_ani_mash      : 1 + ln(2j/(1+j))/k, j = common/denom; common == 0 -> NaN
#include <algorithm>
#include <cmath>
#include <vector>

#include "pa_internal.h"

namespace {

constexpr int kThreads = 256;
constexpr int kWavesPerBlock = kThreads / 64;

__global__ __launch_bounds__(kThreads) void truncated_sizes_kernel(const uint64_t *__restrict__ off, uint32_t n,
                                                                   uint64_t m, uint32_t *__restrict__ sizes) {
  const uint32_t g = blockIdx.x * kThreads + threadIdx.x;
  if (g >= n) return;
  const uint64_t s = off[g + 1] - off[g];
  sizes[g] = (uint32_t)(s < m ? s : m);
}

__global__ __launch_bounds__(kThreads) void truncate_copy_kernel(const uint64_t *__restrict__ in_hashes,
                                                                 const uint64_t *__restrict__ in_off,
                                                                 const uint32_t *__restrict__ out_pos, uint32_t n,
                                                                 uint64_t m, uint64_t *__restrict__ out_hashes,
                                                                 uint64_t *__restrict__ out_off) {
  const uint32_t g = blockIdx.x;  // one workgroup per genome
  const uint64_t src = in_off[g], size = in_off[g + 1] - src;
  const uint64_t keep = size < m ? size : m, dst = out_pos[g];
  for (uint64_t i = threadIdx.x; i < keep; i += kThreads) out_hashes[dst + i] = in_hashes[src + i];
  if (threadIdx.x == 0) {
    out_off[g] = dst;
    if (g == n - 1) out_off[n] = dst + keep;
  }
}

__device__ __forceinline__ uint32_t merge_path(const uint64_t *__restrict__ a, uint32_t na,
                                               const uint64_t *__restrict__ b, uint32_t nb, uint32_t d) {
  uint32_t lo = d > nb ? d - nb : 0, hi = d < na ? d : na;
  while (lo < hi) {
    const uint32_t i = (lo + hi) >> 1;
    if (a[i] <= b[d - i - 1]) lo = i + 1; else hi = i;
  }
  return lo;
}

// Walk `steps` merge steps from (i, j) (ties take A first).  An A step is a new union element and a
// common one if the head of B equals it; a B step is new unless it repeats the A element just taken.
// Stops early once `union_limit` union elements have been seen.
__device__ __forceinline__ void walk(const uint64_t *__restrict__ a, uint32_t na, const uint64_t *__restrict__ b,
                                     uint32_t nb, uint32_t i, uint32_t j, uint32_t steps, uint32_t union_limit,
                                     uint32_t *uni_out, uint32_t *com_out) {
  uint32_t uni = 0, com = 0;
  for (uint32_t t = 0; t < steps && uni < union_limit; ++t) {
    const bool take_a = (j >= nb) || (i < na && a[i] <= b[j]);
    if (take_a) {
      com += (j < nb && a[i] == b[j]) ? 1u : 0u;
      ++uni;
      ++i;
    } else {
      uni += (i > 0 && a[i - 1] == b[j]) ? 0u : 1u;
      ++j;
    }
  }
  *uni_out = uni;
  *com_out = com;
}

__global__ __launch_bounds__(kThreads) void mash_pair_kernel(const uint64_t *__restrict__ hashes,
                                                             const uint64_t *__restrict__ off, uint32_t q0, uint32_t nq,
                                                             uint32_t s0, uint32_t ns, uint32_t m,
                                                             uint32_t *__restrict__ common, uint32_t *__restrict__ denom) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t pair = (uint64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (pair >= (uint64_t)nq * ns) return;
  const uint32_t q = q0 + (uint32_t)(pair / ns), s = s0 + (uint32_t)(pair % ns);
  const uint64_t *__restrict__ a = hashes + off[q];
  const uint64_t *__restrict__ b = hashes + off[s];
  const uint32_t na = (uint32_t)(off[q + 1] - off[q]), nb = (uint32_t)(off[s + 1] - off[s]);
  const uint32_t total = na + nb;
  const uint32_t per = (total + 63u) / 64u;
  const uint32_t d0 = min(lane * per, total), d1 = min(d0 + per, total);
  uint32_t i = 0, j = 0, uni = 0, com = 0;
  if (d0 < d1) {
    i = merge_path(a, na, b, nb, d0);
    j = d0 - i;
    walk(a, na, b, nb, i, j, d1 - d0, 0xffffffffu, &uni, &com);
  }
  // where does the union reach m?
  uint32_t before = uni;  // inclusive scan -> exclusive
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = __shfl_up(before, o, 64);
    if (lane >= (uint32_t)o) before += t;
  }
  const uint32_t incl = before;
  before -= uni;
  uint32_t my_com = com;
  if (before >= m) my_com = 0;  // entirely beyond the m-th union element
  else if (incl > m) {          // the m-th union element falls inside this lane's slice: walk it again, bounded
    uint32_t u2, c2;
    walk(a, na, b, nb, i, j, d1 - d0, m - before, &u2, &c2);
    my_com = c2;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) my_com += __shfl_xor(my_com, o, 64);
  const uint32_t total_union = __shfl(incl, 63, 64);
  if (lane == 0) {
    common[pair] = my_com;
    denom[pair] = total_union < m ? total_union : m;
  }
}

constexpr uint32_t kSentinel = 0xffffffffu;
constexpr uint32_t kLdsBudget = 156u * 1024u;  // of the 160 KB a gfx950 CU has
constexpr uint32_t kMaxTileThreads = 1024;

// Tile kernel.  lds = (tq + ts) lists of `stride` ids: the first min(len, m) ids of the sketch, then a sentinel.
__global__ __launch_bounds__(kMaxTileThreads) void mash_tile_kernel(
    const uint32_t *__restrict__ ids, const uint64_t *__restrict__ off, uint32_t q0, uint32_t nq, uint32_t s0,
    uint32_t ns, uint32_t m, uint32_t tq, uint32_t ts, uint32_t stride, uint32_t tiles_s,
    uint32_t *__restrict__ common, uint32_t *__restrict__ denom) {
  extern __shared__ uint32_t lds[];
  const uint32_t qb = (blockIdx.x / tiles_s) * tq, sb = (blockIdx.x % tiles_s) * ts;
  const uint32_t nqt = min(tq, nq - qb), nst = min(ts, ns - sb);
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
  for (uint32_t l = wave; l < nqt + nst; l += n_waves) {
    const uint32_t g = l < nqt ? q0 + qb + l : s0 + sb + (l - nqt);
    const uint64_t beg = off[g];
    const uint32_t len = (uint32_t)min((uint64_t)m, off[g + 1] - beg);
    uint32_t *dst = lds + (l < nqt ? l : tq + (l - nqt)) * stride;
    for (uint32_t x = lane; x < len; x += 64) dst[x] = ids[beg + x];
    if (lane == 0) dst[len] = kSentinel;
  }
  __syncthreads();
  const uint32_t qi = threadIdx.x / ts, si = threadIdx.x % ts;
  if (qi >= nqt || si >= nst) return;
  const uint32_t *A = lds + qi * stride, *B = lds + (tq + si) * stride;
  uint32_t i = 0, j = 0, uni = 0, com = 0;
  uint32_t x = A[0], y = B[0];
  while (uni < m && (x & y) != kSentinel) {  // ids are < sentinel, so x & y is all ones only when both lists are spent
    const uint32_t adv_a = x <= y ? 1u : 0u, adv_b = y <= x ? 1u : 0u;
    com += adv_a & adv_b;
    ++uni;
    i += adv_a;
    j += adv_b;
    x = A[i];
    y = B[j];
  }
  const uint64_t pair = (uint64_t)(qb + qi) * ns + (sb + si);
  common[pair] = com;
  denom[pair] = uni;
}

__global__ __launch_bounds__(kThreads) void mash_ani_kernel(const uint32_t *__restrict__ common,
                                                            const uint32_t *__restrict__ denom, uint64_t n,
                                                            double inv_k, double *__restrict__ ani) {
  const uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  const uint32_t c = common[i], d = denom[i];
  double v;
  if (c == 0 || d == 0) v = __builtin_nan("");
  else if (c == d) v = 1.0;
  else {
    const double j = (double)c / (double)d;
    v = 1.0 + log(2.0 * j / (1.0 + j)) * inv_k;
  }
  ani[i] = v;
}

}  // namespace

extern "C" {

int pa_sketch_bottom(pa_ctx *c, const uint32_t *d_packed, const uint32_t *d_mask, const uint64_t *d_dirty, uint64_t arena_bases,
                     const uint64_t *h_genome_start, uint32_t n_genomes, uint32_t k, uint32_t m, uint64_t *d_hashes,
                     uint64_t cap_hashes, uint64_t *d_off, uint64_t *h_total) {
  PA_REQUIRE(c && d_off && h_total && h_genome_start, "pa_sketch_bottom: null argument");
  PA_REQUIRE(m >= 1, "pa_sketch_bottom: m must be at least 1");
  PA_REQUIRE(cap_hashes >= (uint64_t)n_genomes * m || n_genomes == 0, "pa_sketch_bottom: room for n*m = %llu hashes needed",
             (unsigned long long)n_genomes * m);
  *h_total = 0;
  if (n_genomes == 0) { PA_HIP(hipMemsetAsync(d_off, 0, sizeof(uint64_t), c->stream)); return PA_OK; }
  // threshold for ~4m survivors of the shortest genome (positions, an upper bound on windows)
  uint64_t shortest = ~0ULL;
  for (uint32_t g = 0; g < n_genomes; ++g) {
    const uint64_t len = h_genome_start[g + 1] - h_genome_start[g];
    if (len && len < shortest) shortest = len;
  }
  if (shortest == ~0ULL) shortest = 1;
  double frac = 4.0 * (double)m / (double)shortest;
  DevBuf tmp_hashes, tmp_off, sizes, pos;
  struct Release { DevBuf *b[4]; ~Release() { for (DevBuf *x : b) x->release(); } } rel{{&tmp_hashes, &tmp_off, &sizes, &pos}};
  PA_TRY(tmp_off.reserve((uint64_t)(n_genomes + 1) * 8));
  PA_TRY(sizes.reserve((uint64_t)n_genomes * 4 + 16));
  PA_TRY(pos.reserve((uint64_t)n_genomes * 4 + 16));
  std::vector<uint64_t> h_off(n_genomes + 1);
  for (;;) {
    const uint64_t max_hash = frac >= 1.0 ? ~0ULL : (uint64_t)(frac * 18446744073709551616.0);
    uint64_t cap = (uint64_t)((double)arena_bases * (frac >= 1.0 ? 1.0 : frac) * 1.25) + 65536;
    if (cap > arena_bases) cap = arena_bases ? arena_bases : 1;
    uint64_t total = 0;
    int st = PA_E_CAPACITY;
    for (int attempt = 0; attempt < 2 && st == PA_E_CAPACITY; ++attempt) {
      PA_TRY(tmp_hashes.reserve(cap * 8));
      st = pa_sketch(c, d_packed, d_mask, d_dirty, arena_bases, h_genome_start, n_genomes, k, max_hash, tmp_hashes.as<uint64_t>(),
                     cap, tmp_off.as<uint64_t>(), &total);
      if (st == PA_E_CAPACITY) cap = total;
    }
    if (st != PA_OK) return st;
    PA_HIP(hipMemcpyAsync(h_off.data(), tmp_off.p, (uint64_t)(n_genomes + 1) * 8, hipMemcpyDeviceToHost, c->stream));
    PA_HIP(hipStreamSynchronize(c->stream));
    bool short_sketch = false;
    for (uint32_t g = 0; g < n_genomes && !short_sketch; ++g) short_sketch = h_off[g + 1] - h_off[g] < m;
    if (!short_sketch || max_hash == ~0ULL) break;  // with the threshold at its maximum a short sketch is simply all there is
    frac *= 8.0;
  }
  hipLaunchKernelGGL(truncated_sizes_kernel, dim3(ceil_div_u64(n_genomes, kThreads)), dim3(kThreads), 0, c->stream,
                     tmp_off.as<uint64_t>(), n_genomes, (uint64_t)m, sizes.as<uint32_t>());
  PA_TRY(pa_exclusive_scan_u32(c, sizes.as<uint32_t>(), pos.as<uint32_t>(), n_genomes, c->counters.as<uint64_t>() + 1));
  hipLaunchKernelGGL(truncate_copy_kernel, dim3(n_genomes), dim3(kThreads), 0, c->stream, tmp_hashes.as<uint64_t>(),
                     tmp_off.as<uint64_t>(), pos.as<uint32_t>(), n_genomes, (uint64_t)m, d_hashes, d_off);
  PA_HIP(hipMemcpyAsync(c->h_pinned, c->counters.as<uint64_t>() + 1, 8, hipMemcpyDeviceToHost, c->stream));
  PA_HIP(hipStreamSynchronize(c->stream));
  *h_total = c->h_pinned[0];
  return PA_OK;
}

int pa_pair_mash(pa_ctx *c, const uint64_t *d_hashes, const uint64_t *d_off, uint32_t n, uint32_t q0, uint32_t q1,
                 uint32_t s0, uint32_t s1, uint32_t m, uint32_t *d_common, uint32_t *d_denom) {
  PA_REQUIRE(c && d_off, "pa_pair_mash: null argument");
  PA_REQUIRE(q0 <= q1 && q1 <= n && s0 <= s1 && s1 <= n && m >= 1, "pa_pair_mash: bad ranges or m");
  PA_HIP(hipSetDevice(c->device));
  const uint64_t pairs = (uint64_t)(q1 - q0) * (s1 - s0);
  if (pairs == 0) return PA_OK;
  PA_REQUIRE(d_hashes && d_common && d_denom, "pa_pair_mash: null buffer");
  PA_REQUIRE(pairs / kWavesPerBlock < (1ULL << 31), "pa_pair_mash: tile of %llu pairs is too large for one launch",
             (unsigned long long)pairs);
  // lengths on the host: the longest staged list decides whether a tile of lists fits in LDS
  std::vector<uint64_t> h_off(n + 1);
  PA_HIP(hipMemcpyAsync(h_off.data(), d_off, (uint64_t)(n + 1) * 8, hipMemcpyDeviceToHost, c->stream));
  PA_HIP(hipStreamSynchronize(c->stream));
  uint64_t longest = 0;
  for (uint32_t g = q0; g < q1; ++g) longest = std::max(longest, std::min<uint64_t>(m, h_off[g + 1] - h_off[g]));
  for (uint32_t g = s0; g < s1; ++g) longest = std::max(longest, std::min<uint64_t>(m, h_off[g + 1] - h_off[g]));
  const uint64_t P = h_off[n];
  const uint32_t stride = (uint32_t)longest + 1u;
  const uint32_t lists = kLdsBudget / (4u * stride);
  if (lists >= 2 && P < (1ULL << 32) && P > 0) {
    uint64_t n_distinct = 0;
    PA_TRY(pa_dense_ids_sorted(c, d_hashes, d_off, n, P, &n_distinct));
    ProfScope prof(c, PA_PROF_PAIR_COUNT);
    const uint32_t nq = q1 - q0, ns = s1 - s0;
    uint32_t tq = std::min({lists / 2u, nq, 32u});
    uint32_t ts = std::min({lists - tq, ns, kMaxTileThreads / tq});
    const uint32_t tiles_q = (nq + tq - 1) / tq, tiles_s = (ns + ts - 1) / ts;
    PA_REQUIRE((uint64_t)tiles_q * tiles_s < (1ULL << 31), "pa_pair_mash: %llu tiles are too many for one launch",
               (unsigned long long)tiles_q * tiles_s);
    const uint32_t threads = ((tq * ts + 63u) / 64u) * 64u;
    const uint32_t lds_bytes = (tq + ts) * stride * 4u;
    PA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(mash_tile_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipLaunchKernelGGL(mash_tile_kernel, dim3(tiles_q * tiles_s), dim3(threads), lds_bytes, c->stream,
                       c->ids.as<uint32_t>(), d_off, q0, nq, s0, ns, m, tq, ts, stride, tiles_s, d_common, d_denom);
    PA_HIP(hipGetLastError());
    return PA_OK;
  }
  ProfScope prof(c, PA_PROF_PAIR_COUNT);
  hipLaunchKernelGGL(mash_pair_kernel, dim3(ceil_div_u64(pairs, kWavesPerBlock)), dim3(kThreads), 0, c->stream, d_hashes,
                     d_off, q0, q1 - q0, s0, s1 - s0, m, d_common, d_denom);
  PA_HIP(hipGetLastError());
  return PA_OK;
}

int pa_ani_mash(pa_ctx *c, const uint32_t *d_common, const uint32_t *d_denom, uint64_t n_pairs, uint32_t k,
                double *d_ani) {
  PA_REQUIRE(c && k >= 1, "pa_ani_mash: null context or k == 0");
  if (n_pairs == 0) return PA_OK;
  PA_REQUIRE(d_common && d_denom && d_ani, "pa_ani_mash: null buffer");
  PA_HIP(hipSetDevice(c->device));
  ProfScope prof(c, PA_PROF_ANI);
  hipLaunchKernelGGL(mash_ani_kernel, dim3(ceil_div_u64(n_pairs, kThreads)), dim3(kThreads), 0, c->stream, d_common,
                     d_denom, n_pairs, 1.0 / (double)k, d_ani);
  PA_HIP(hipGetLastError());
  return PA_OK;
}

}  // extern "C"
