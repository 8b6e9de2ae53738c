// pairs_merge.hip -- per-pair sorted-list intersection (gfx950), second algorithm.
//
// The literal counterpart of `sourmash scripts manysearch`
// (pyani_plus/methods/sourmash.py:184-200): |Q n S| by merging two ascending
// u64 lists.  One wavefront per ordered pair: the merge path of (Q, S) is cut
// into 64 equal diagonals, each lane binary-searches its start point and walks
// its slice, counting the steps where the head of Q equals the head of S; a
// wave reduction adds the 64 partial counts.  Used to cross-check the bit-row
// path at full size and for callers that force PA_PAIRS_MERGE.
#include "pa_internal.h"

namespace {

constexpr int kThreads = 256;
constexpr int kWavesPerBlock = kThreads / 64;

// Merge-path split: number of elements taken from A among the first d merged
// elements, ties taking A first (A[i] <= B[j] -> take A).
__device__ __forceinline__ uint32_t merge_path(const uint64_t *__restrict__ a, uint32_t na,
                                               const uint64_t *__restrict__ b, uint32_t nb, uint32_t d) {
  uint32_t lo = d > nb ? d - nb : 0, hi = d < na ? d : na;
  while (lo < hi) {
    const uint32_t i = (lo + hi) >> 1;  // candidate: i from A, d-i from B
    // too few from A if A[i] <= B[d-i-1]
    if (a[i] <= b[d - i - 1]) lo = i + 1; else hi = i;
  }
  return lo;
}

__global__ __launch_bounds__(kThreads) void merge_count_kernel(const uint64_t *__restrict__ hashes,
                                                               const uint64_t *__restrict__ off, uint32_t q0,
                                                               uint32_t nq, uint32_t s0, uint32_t ns,
                                                               uint32_t *__restrict__ counts) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t pair = (uint64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (pair >= (uint64_t)nq * ns) return;
  const uint32_t q = q0 + (uint32_t)(pair / ns), s = s0 + (uint32_t)(pair % ns);
  const uint64_t *__restrict__ a = hashes + off[q];
  const uint64_t *__restrict__ b = hashes + off[s];
  const uint32_t na = (uint32_t)(off[q + 1] - off[q]), nb = (uint32_t)(off[s + 1] - off[s]);
  uint32_t c = 0;
  if (na && nb) {  // the diagonal is merged like any other pair: this kernel is the independent check of the bit-row path
    const uint32_t total = na + nb;
    const uint32_t per = (total + 63u) / 64u;
    const uint32_t d0 = min(lane * per, total), d1 = min(d0 + per, total);
    if (d0 < d1) {
      uint32_t i = merge_path(a, na, b, nb, d0);
      uint32_t j = d0 - i;
      uint64_t av = i < na ? a[i] : ~0ULL, bv = j < nb ? b[j] : ~0ULL;
      for (uint32_t d = d0; d < d1; ++d) {
        const bool take_a = (j >= nb) || (i < na && av <= bv);
        if (take_a) {
          c += (j < nb && av == bv) ? 1u : 0u;
          ++i;
          av = i < na ? a[i] : ~0ULL;
        } else {
          ++j;
          bv = j < nb ? b[j] : ~0ULL;
        }
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  if (lane == 0) counts[pair] = c;
}

}  // namespace

int pa_pairs_merge(pa_ctx *c, const uint64_t *d_hashes, const uint64_t *d_off, uint32_t n, uint32_t q0, uint32_t q1,
                   uint32_t s0, uint32_t s1, uint32_t *d_counts) {
  (void)n;
  const uint32_t nq = q1 - q0, ns = s1 - s0;
  const uint64_t pairs = (uint64_t)nq * ns;
  if (pairs == 0) return PA_OK;
  PA_REQUIRE(pairs / kWavesPerBlock < (1ULL << 31), "pa_pairs_merge: tile of %llu pairs is too large for one launch",
             (unsigned long long)pairs);
  ProfScope prof(c, PA_PROF_PAIR_COUNT);
  hipLaunchKernelGGL(merge_count_kernel, dim3(ceil_div_u64(pairs, kWavesPerBlock)), dim3(kThreads), 0, c->stream,
                     d_hashes, d_off, q0, nq, s0, ns, d_counts);
  PA_HIP(hipGetLastError());
  return PA_OK;
}
