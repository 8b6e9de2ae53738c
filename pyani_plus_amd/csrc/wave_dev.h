// wave_dev.h -- wave64 reductions and scans on the DPP path (gfx950).
//
// `__shfl_*` lowers to ds_bpermute_b32: every step is a round trip through the LDS pipeline (~100 cycles
// of latency, shared with the kernel's real LDS traffic).  Row shifts, mirrors and the row broadcasts of
// the GFX9 DPP encoding do the same data movement inside the VALU in a few cycles.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace pa_dev {

// DPP controls (GFX9 encoding)
constexpr int kDppQuadSwap1 = 0xB1;       // quad_perm:[1,0,3,2]
constexpr int kDppQuadSwap2 = 0x4E;       // quad_perm:[2,3,0,1]
constexpr int kDppRowShr = 0x110;         // + n
constexpr int kDppRowMirror = 0x140;
constexpr int kDppRowHalfMirror = 0x141;
constexpr int kDppRowBcast15 = 0x142;
constexpr int kDppRowBcast31 = 0x143;

template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ uint32_t dpp_or_zero(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, BANK_MASK, true);
}

// sum over the 64 lanes, uniform result
__device__ __forceinline__ uint32_t wave_sum_dpp(uint32_t v) {
  v += dpp_or_zero<kDppQuadSwap1>(v);
  v += dpp_or_zero<kDppQuadSwap2>(v);
  v += dpp_or_zero<kDppRowHalfMirror>(v);
  v += dpp_or_zero<kDppRowMirror>(v);  // every lane: the sum of its row of 16
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 0) + (uint32_t)__builtin_amdgcn_readlane((int)v, 16) +
         (uint32_t)__builtin_amdgcn_readlane((int)v, 32) + (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
}

// maximum over the 64 lanes, uniform result (unsigned)
__device__ __forceinline__ uint32_t wave_max_dpp(uint32_t v) {
  v = max(v, dpp_or_zero<kDppQuadSwap1>(v));
  v = max(v, dpp_or_zero<kDppQuadSwap2>(v));
  v = max(v, dpp_or_zero<kDppRowHalfMirror>(v));
  v = max(v, dpp_or_zero<kDppRowMirror>(v));  // every lane: the maximum of its row of 16
  return max(max((uint32_t)__builtin_amdgcn_readlane((int)v, 0), (uint32_t)__builtin_amdgcn_readlane((int)v, 16)),
             max((uint32_t)__builtin_amdgcn_readlane((int)v, 32), (uint32_t)__builtin_amdgcn_readlane((int)v, 48)));
}
__device__ __forceinline__ uint32_t wave_min_dpp(uint32_t v) { return ~wave_max_dpp(~v); }

// inclusive prefix sum over the 64 lanes
__device__ __forceinline__ uint32_t wave_incl_scan_dpp(uint32_t v) {
  uint32_t x = v;
  x += dpp_or_zero<kDppRowShr + 1>(v);
  x += dpp_or_zero<kDppRowShr + 2>(v);
  x += dpp_or_zero<kDppRowShr + 3>(v);              // 4 consecutive lanes
  x += dpp_or_zero<kDppRowShr + 4, 0xf, 0xe>(x);    // 8
  x += dpp_or_zero<kDppRowShr + 8, 0xf, 0xc>(x);    // the row of 16
  x += dpp_or_zero<kDppRowBcast15, 0xa, 0xf>(x);    // rows 1 and 3 take the total of the row before
  x += dpp_or_zero<kDppRowBcast31, 0xc, 0xf>(x);    // rows 2 and 3 take the total of the first half
  return x;
}

// inclusive prefix sums over each row of 16 lanes (four independent scans)
__device__ __forceinline__ uint32_t row_incl_scan_dpp(uint32_t v) {
  uint32_t x = v;
  x += dpp_or_zero<kDppRowShr + 1>(v);
  x += dpp_or_zero<kDppRowShr + 2>(v);
  x += dpp_or_zero<kDppRowShr + 3>(v);
  x += dpp_or_zero<kDppRowShr + 4, 0xf, 0xe>(x);
  x += dpp_or_zero<kDppRowShr + 8, 0xf, 0xc>(x);
  return x;
}

// inclusive prefix maximum over the 64 lanes (unsigned; lanes a step does not reach take 0, the identity)
__device__ __forceinline__ uint32_t wave_incl_max_scan_dpp(uint32_t v) {
  uint32_t x = v;
  x = max(x, dpp_or_zero<kDppRowShr + 1>(v));
  x = max(x, dpp_or_zero<kDppRowShr + 2>(v));
  x = max(x, dpp_or_zero<kDppRowShr + 3>(v));
  x = max(x, dpp_or_zero<kDppRowShr + 4, 0xf, 0xe>(x));
  x = max(x, dpp_or_zero<kDppRowShr + 8, 0xf, 0xc>(x));
  x = max(x, dpp_or_zero<kDppRowBcast15, 0xa, 0xf>(x));
  x = max(x, dpp_or_zero<kDppRowBcast31, 0xc, 0xf>(x));
  return x;
}

}  // namespace pa_dev
