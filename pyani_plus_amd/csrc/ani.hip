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

// One row of the matrix per blockIdx.y, two neighbouring columns per thread: no 64-bit division to find the pair, the
// counts arrive as one 8-byte load and each output leaves as one 16-byte store where the row is so aligned (10^8 pairs
// at N = 10^4: the kernel moves 2 GB and took 1.2 ms as one thread per pair, a division and three 4/8-byte accesses each).
__global__ __launch_bounds__(kThreads) void ani_kernel(const uint32_t *__restrict__ counts,
                                                       const uint64_t *__restrict__ off, uint32_t q0, uint32_t nq,
                                                       uint32_t s0, uint32_t ns, double inv_k,
                                                       double *__restrict__ identity, double *__restrict__ cov_query) {
  const uint32_t s = 2u * (blockIdx.x * kThreads + threadIdx.x);
  if (s >= ns) return;
  for (uint32_t q = blockIdx.y; q < nq; q += gridDim.y) {  // a grid has at most 65 535 rows of blocks
  const uint64_t idx = (uint64_t)q * ns + s;
  const bool two = s + 1u < ns;
  // the pair of columns starts on an 8-byte (counts) / 16-byte (outputs) boundary
  const bool aligned = ((reinterpret_cast<uintptr_t>(counts + idx) & 7u) | (reinterpret_cast<uintptr_t>(identity + idx) & 15u) |
                        (reinterpret_cast<uintptr_t>(cov_query + idx) & 15u)) == 0u;
  uint32_t c[2];
  if (two && aligned) {
    const uint2 v = *reinterpret_cast<const uint2 *>(counts + idx);
    c[0] = v.x;
    c[1] = v.y;
  } else {
    c[0] = counts[idx];
    c[1] = two ? counts[idx + 1] : 0u;
  }
  const double nan = __builtin_nan("");
  double ident[2] = {nan, nan}, cov[2] = {nan, nan};
  if (c[0] | c[1]) {
    const double qs = (double)(off[q0 + q + 1] - off[q0 + q]);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (c[j] == 0) continue;
      const double ss = (double)(off[s0 + s + j + 1] - off[s0 + s + j]);
      const double qa = pow((double)c[j] / qs, inv_k);
      const double ma = pow((double)c[j] / ss, inv_k);
      ident[j] = qa > ma ? qa : ma;
      cov[j] = qa;
    }
  }
  if (two && aligned) {
    *reinterpret_cast<double2 *>(identity + idx) = make_double2(ident[0], ident[1]);
    *reinterpret_cast<double2 *>(cov_query + idx) = make_double2(cov[0], cov[1]);
  } else {
    identity[idx] = ident[0];
    cov_query[idx] = cov[0];
    if (two) { identity[idx + 1] = ident[1]; cov_query[idx + 1] = cov[1]; }
  }
  }
}
}  // namespace

int pa_launch_ani(pa_ctx *c, const uint32_t *d_counts, const uint64_t *d_off, uint32_t q0, uint32_t q1, uint32_t s0,
                  uint32_t s1, uint32_t k, double *d_identity, double *d_cov_query) {
  const uint32_t nq = q1 - q0, ns = s1 - s0;
  const uint64_t total = (uint64_t)nq * ns;
  if (total == 0) return PA_OK;
  ProfScope prof(c, PA_PROF_ANI);
  hipLaunchKernelGGL(ani_kernel, dim3(ceil_div_u64((ns + 1u) / 2u, kThreads), nq < 65535u ? nq : 65535u), dim3(kThreads), 0, c->stream, d_counts,
                     d_off, q0, nq, s0, ns, 1.0 / (double)k, d_identity, d_cov_query);
  PA_HIP(hipGetLastError());
  return PA_OK;
}
