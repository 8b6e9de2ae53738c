// murmur_dev.h -- device helpers shared by the k-mer hashing kernels (gfx950).
//
// MurmurHash3_x64_128 (Appleby, public-domain algorithm), seed 42, first 64-bit word: the hash
// sourmash applies to canonical k-mers (call site pyani_plus/methods/sourmash.py:67-83) and, in its
// low 32 bits, the one fastANI/Mashmap applies to both strands of every k-mer
// (call site pyani_plus/private_cli.py:1044-1063).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace pa_dev {

__device__ __forceinline__ uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t sh) {
  return __builtin_amdgcn_alignbit(hi, lo, sh);
}

// 4 bases (8 bits, base j at bits 2j) -> 4 ASCII bytes (base j in byte j).
// spread the 2-bit codes to one per byte, then use v_perm_b32 as a 4-entry LUT.
__device__ __forceinline__ uint32_t ascii4(uint32_t b8) {
  uint32_t u = (b8 | (b8 << 12)) & 0x000F000Fu;
  uint32_t v = (u | (u << 6)) & 0x03030303u;
  // selector byte value 0..3 picks that byte of the second source: "ACGT" little-endian
  return __builtin_amdgcn_perm(0u, 0x54474341u, v);
}

__device__ __forceinline__ uint64_t u64_of(uint32_t lo, uint32_t hi) { return ((uint64_t)hi << 32) | lo; }

// 64-bit rotate as two funnel shifts (v_alignbit_b32); r is a compile-time constant in 1..63
__device__ __forceinline__ uint64_t rotl64(uint64_t x, int r) {
  const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
  if (r == 32) return u64_of(hi, lo);
  if (r < 32) return u64_of(alignbit(lo, hi, 32 - r), alignbit(hi, lo, 32 - r));
  return u64_of(alignbit(hi, lo, 64 - r), alignbit(lo, hi, 64 - r));
}

__device__ __forceinline__ uint64_t fmix64(uint64_t k) {
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdULL;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ULL;
  k ^= k >> 33;
  return k;
}

// ---- MurmurHash3_x64_128(seed 42).h1, split at the first multiply -----------------
// The K ASCII bytes form up to four little-endian 64-bit words; word j is a "k1" word
// (j even: k*c1, rotl 31, *c2, xor into h1) or a "k2" word (j odd: k*c2, rotl 33, *c1,
// xor into h2), in the 16-byte blocks and in the tail alike.  P[j] is the FIRST product
// (word * c1 or c2); everything after it is computed here.
constexpr uint64_t kC1 = 0x87c37b91114253d5ULL, kC2 = 0x4cf5ad432745937fULL;

// h*5 + c.  Left to itself hipcc turns this into two v_mad_u64_u32 plus fix-ups; gfx950 has a 64-bit
// shift-and-add (v_lshl_add_u64, shift 0..4), so (h << 2) + h is one instruction and + c a second.
__device__ __forceinline__ uint64_t times5_plus(uint64_t h, uint64_t c) {
  uint64_t t;
  asm("v_lshl_add_u64 %0, %1, 2, %1" : "=v"(t) : "v"(h));
  return t + c;
}

constexpr uint64_t kF1 = 0xff51afd7ed558ccdULL, kF2 = 0xc4ceb9fe1a85ec53ULL;  // fmix64 multipliers

// Everything up to, but not including, the LAST multiply of the two fmix64 calls: with U and V the values this
// returns, the hash is h = fin(U*kF2) + fin(V*kF2), fin(x) = x ^ (x >> 33).  fin only touches the low 31 bits, so
// the high word of h is hi32(U*kF2) + hi32(V*kF2) + at most one carry -- and that sum of two high words is linear:
//   hi32(U*c) + hi32(V*c) = mulhi(U.lo, c.lo) + mulhi(V.lo, c.lo) + (U.lo + V.lo)*c.hi + (U.hi + V.hi)*c.lo  (mod 2^32)
// which is what the FracMinHash screen of kmer_hash.hip tests before anything else of the last step is computed.
// (The 64-bit multiplies are left to hipcc: v_mad_u64_u32 + 2 v_mul_lo_u32 + v_add3_u32.  A hand-written
// chain of three v_mad_u64_u32 + one add was measured at the same speed.)
template <int K, int N>
__device__ __forceinline__ void murmur3_pre_last_mul(const uint64_t (&P)[N], uint64_t &U, uint64_t &V) {
  static_assert(N >= (K + 7) / 8, "one first product per 8-byte word of the k-mer");
  uint64_t h1 = 42, h2 = 42;
  constexpr int nblocks = K / 16;
  constexpr int tail = K % 16;
#pragma unroll
  for (int i = 0; i < nblocks; ++i) {
    const uint64_t k1 = rotl64(P[2 * i], 31) * kC2;
    h1 ^= k1;
    if (i == 0) {  // h2 is still the seed: (rotl(h1) + 42) * 5 + c = rotl(h1) * 5 + (5 * 42 + c), one 64-bit add less
      h1 = times5_plus(rotl64(h1, 27), 5ULL * 42ULL + 0x52dce729ULL);
    } else {
      h1 = rotl64(h1, 27) + h2;
      h1 = times5_plus(h1, 0x52dce729ULL);
    }
    const uint64_t k2 = rotl64(P[2 * i + 1], 33) * kC1;
    h2 ^= k2;
    h2 = rotl64(h2, 31) + h1;
    h2 = times5_plus(h2, 0x38495ab5ULL);
  }
  if constexpr (tail > 8) h2 ^= rotl64(P[2 * nblocks + 1], 33) * kC1;
  if constexpr (tail > 0) h1 ^= rotl64(P[2 * nblocks], 31) * kC2;
  h1 ^= (uint64_t)K; h2 ^= (uint64_t)K;
  h1 += h2; h2 += h1;
  h1 ^= h1 >> 33; h1 *= kF1; h1 ^= h1 >> 33;
  h2 ^= h2 >> 33; h2 *= kF1; h2 ^= h2 >> 33;
  U = h1;
  V = h2;
}

// X.hi + Y.hi + 1 (mod 2^32) for X = U*kF2, Y = V*kF2, without forming X and Y (see above)
__device__ __forceinline__ uint32_t last_mul_high_sum_plus1(uint64_t U, uint64_t V) {
  constexpr uint32_t c0 = (uint32_t)kF2, c1 = (uint32_t)(kF2 >> 32);
  const uint32_t u0 = (uint32_t)U, u1 = (uint32_t)(U >> 32), v0 = (uint32_t)V, v1 = (uint32_t)(V >> 32);
  return (__umulhi(u0, c0) + __umulhi(v0, c0) + 1u) + ((u0 + v0) * c1 + (u1 + v1) * c0);
}

template <int K, int N>
__device__ __forceinline__ uint64_t murmur3_from_products(const uint64_t (&P)[N]) {
  uint64_t U, V;
  murmur3_pre_last_mul<K, N>(P, U, V);
  const uint64_t X = U * kF2, Y = V * kF2;
  return (X ^ (X >> 33)) + (Y ^ (Y >> 33));
}

// the 4-base group `b8` (base j at bits 2j) as ASCII, keeping only its first `nv` bytes
__device__ __forceinline__ uint32_t ascii_group(uint32_t b8, int nv) {
  const uint32_t a = ascii4(b8);
  return nv >= 4 ? a : (nv <= 0 ? 0u : (a & ((1u << (8 * nv)) - 1u)));
}

}  // namespace pa_dev
