// sketch_build.hip -- candidates sorted (genome-major, hash-minor) -> CSR sketches.
//
// Replaces the "sorted, de-duplicated mins" step of `sourmash scripts
// singlesketch` (pyani_plus/methods/sourmash.py:67-83; the result is the
// `mins` array of the `.sig` fixtures).  Duplicates (a k-mer seen at several
// positions of one genome) are adjacent after the sort; head flags + one
// exclusive scan compact them and give every genome its CSR offset.
#include "pa_internal.h"

namespace {

constexpr int kThreads = 256;

__global__ __launch_bounds__(kThreads) void head_flags_kernel(const uint64_t *__restrict__ hash,
                                                              const uint32_t *__restrict__ genome, uint64_t n,
                                                              uint32_t *__restrict__ flags) {
  const uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  flags[i] = (i == 0 || hash[i] != hash[i - 1] || genome[i] != genome[i - 1]) ? 1u : 0u;
}

// pos = exclusive scan of flags.  Writes unique hashes and the CSR offsets:
// off[g] = number of unique postings whose genome index is < g.
__global__ __launch_bounds__(kThreads) void compact_csr_kernel(
    const uint64_t *__restrict__ hash, const uint32_t *__restrict__ genome, const uint32_t *__restrict__ flags,
    const uint32_t *__restrict__ pos, uint64_t n, uint32_t n_genomes, uint64_t cap, uint64_t *__restrict__ out_hash,
    uint64_t *__restrict__ off) {
  const uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  const uint32_t g = genome[i];
  const uint32_t p = pos[i];
  if (flags[i]) {
    if (p < cap) out_hash[p] = hash[i];
    const int64_t gp = (i == 0) ? -1 : (int64_t)genome[i - 1];
    for (int64_t x = gp + 1; x <= (int64_t)g; ++x) off[x] = p;  // first posting of genome g (and empty ones before it)
  }
  if (i == n - 1) {
    const uint64_t total = (uint64_t)p + flags[i];
    for (uint32_t x = g + 1; x <= n_genomes; ++x) off[x] = total;
  }
}

}  // namespace

int pa_build_sketch_csr(pa_ctx *c, const uint64_t *d_sorted_hash, const uint32_t *d_sorted_genome, uint64_t n_cand,
                        uint32_t n_genomes, uint64_t *d_hashes, uint64_t cap_hashes, uint64_t *d_off,
                        uint64_t *h_total) {
  if (n_cand == 0) {
    PA_HIP(hipMemsetAsync(d_off, 0, (uint64_t)(n_genomes + 1) * sizeof(uint64_t), c->stream));
    *h_total = 0;
    return PA_OK;
  }
  PA_TRY(c->flags.reserve(2 * n_cand * sizeof(uint32_t)));
  uint32_t *d_flags = c->flags.as<uint32_t>();
  uint32_t *d_pos = d_flags + n_cand;
  uint64_t *d_total = c->counters.as<uint64_t>() + 1;
  const uint32_t grid = ceil_div_u64(n_cand, kThreads);
  hipLaunchKernelGGL(head_flags_kernel, dim3(grid), dim3(kThreads), 0, c->stream, d_sorted_hash, d_sorted_genome,
                     n_cand, d_flags);
  PA_TRY(pa_exclusive_scan_u32(c, d_flags, d_pos, n_cand, d_total));
  PA_HIP(hipMemcpyAsync(c->h_pinned, d_total, sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
  PA_HIP(hipStreamSynchronize(c->stream));
  *h_total = c->h_pinned[0];
  // offsets are always produced (they are valid even when the payload does not fit)
  hipLaunchKernelGGL(compact_csr_kernel, dim3(grid), dim3(kThreads), 0, c->stream, d_sorted_hash, d_sorted_genome,
                     d_flags, d_pos, n_cand, n_genomes, *h_total <= cap_hashes ? cap_hashes : 0ULL, d_hashes, d_off);
  PA_HIP(hipGetLastError());
  if (*h_total > cap_hashes) {
    pa_set_error("sketch output needs %llu hashes, caller gave room for %llu", (unsigned long long)*h_total,
                 (unsigned long long)cap_hashes);
    return PA_E_CAPACITY;
  }
  return PA_OK;
}
