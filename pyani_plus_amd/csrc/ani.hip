// ani.hip -- intersection counts -> (identity, cov_query) on the device (gfx950).
//
// Replaces the manysearch CSV columns the reference reads
// (pyani_plus/methods/sourmash.py:107-110) and their mapping at
// pyani_plus/private_cli.py:1879-1880:
//   cov_query = query_containment_ani = (I/|Q|)^(1/k)
//   identity  = max_containment_ani   = max(cov_query, (I/|S|)^(1/k))
// I == 0 -> the pair is absent from the CSV -> NULL (sourmash.py:141-144);
// encoded here as NaN in both outputs.
#include "pa_internal.h"

namespace {
constexpr int kThreads = 256;

__global__ __launch_bounds__(kThreads) void ani_kernel(const uint32_t *__restrict__ counts,
                                                       const uint64_t *__restrict__ off, uint32_t q0, uint32_t nq,
                                                       uint32_t s0, uint32_t ns, double inv_k,
                                                       double *__restrict__ identity, double *__restrict__ cov_query) {
  const uint64_t idx = (uint64_t)blockIdx.x * kThreads + threadIdx.x;
  if (idx >= (uint64_t)nq * ns) return;
  const uint32_t q = (uint32_t)(idx / ns), s = (uint32_t)(idx % ns);
  const uint32_t c = counts[idx];
  if (c == 0) {
    const double nan = __builtin_nan("");
    identity[idx] = nan;
    cov_query[idx] = nan;
    return;
  }
  const double qs = (double)(off[q0 + q + 1] - off[q0 + q]);
  const double ss = (double)(off[s0 + s + 1] - off[s0 + s]);
  const double qa = pow((double)c / qs, inv_k);
  const double ma = pow((double)c / ss, inv_k);
  identity[idx] = qa > ma ? qa : ma;
  cov_query[idx] = qa;
}
}  // namespace

int pa_launch_ani(pa_ctx *c, const uint32_t *d_counts, const uint64_t *d_off, uint32_t q0, uint32_t q1, uint32_t s0,
                  uint32_t s1, uint32_t k, double *d_identity, double *d_cov_query) {
  const uint32_t nq = q1 - q0, ns = s1 - s0;
  const uint64_t total = (uint64_t)nq * ns;
  if (total == 0) return PA_OK;
  ProfScope prof(c, PA_PROF_ANI);
  hipLaunchKernelGGL(ani_kernel, dim3(ceil_div_u64(total, kThreads)), dim3(kThreads), 0, c->stream, d_counts, d_off,
                     q0, nq, s0, ns, 1.0 / (double)k, d_identity, d_cov_query);
  PA_HIP(hipGetLastError());
  return PA_OK;
}
