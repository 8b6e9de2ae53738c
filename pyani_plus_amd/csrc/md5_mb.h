// md5_mb.h -- MD5 of many buffers at once (host side of the FASTA front-end).
//
// Genome identity in pyani-plus is the md5 of the decompressed FASTA bytes (pyani_plus/utils.py:142-196), and
// md5 is a strictly serial chain inside one message: ~5 cycles per step whatever the core can issue.  It is the
// largest single cost of the front-end (5 GB of text per 1 000 genomes), and the boxes this runs on give a
// container 16 CPUs' worth of time.  Sixteen files are independent messages, so they go through the rounds side
// by side, one 32-bit lane of a 512-bit register each (AVX-512F: vprold, vpternlogd), which costs the same
// latency chain once for sixteen blocks.  Without AVX-512 the unrolled scalar form below is used.
#pragma once
#include <immintrin.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <vector>

namespace md5mb {

constexpr uint32_t kInit[4] = {0x67452301u, 0xefcdab89u, 0x98badcfeu, 0x10325476u};
alignas(64) constexpr uint32_t kK[64] = {
    0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501,
    0x698098d8, 0x8b44f7af, 0xffff5bb1, 0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821,
    0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa, 0xd62f105d, 0x02441453, 0xd8a1e681, 0xe7d3fbc8,
    0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8, 0x676f02d9, 0x8d2a4c8a,
    0xfffa3942, 0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70,
    0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05, 0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665,
    0xf4292244, 0x432aff97, 0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92, 0xffeff47d, 0x85845dd1,
    0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1, 0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391};

// ---- one message, unrolled (the compiler keeps the sixteen words and the state in registers)
inline void blocks_scalar(uint32_t st[4], const uint8_t *p, size_t n_blocks) {
  uint32_t a = st[0], b = st[1], c = st[2], d = st[3];
  for (size_t blk = 0; blk < n_blocks; ++blk, p += 64) {
    uint32_t m[16];
    memcpy(m, p, 64);  // little-endian host
    const uint32_t a0 = a, b0 = b, c0 = c, d0 = d;
#define PA_MD5_ROL(x, s) (((x) << (s)) | ((x) >> (32 - (s))))
#define PA_MD5_F(x, y, z) ((z) ^ ((x) & ((y) ^ (z))))
#define PA_MD5_G(x, y, z) ((y) ^ ((z) & ((x) ^ (y))))
#define PA_MD5_H(x, y, z) ((x) ^ (y) ^ (z))
#define PA_MD5_I(x, y, z) ((y) ^ ((x) | ~(z)))
#define PA_MD5_STEP(f, w, x, y, z, g, i, s) \
  w += f(x, y, z) + m[g] + kK[i];           \
  w = PA_MD5_ROL(w, s) + x;
#define PA_MD5_ROUND4(f, i, g0, g1, g2, g3, s0, s1, s2, s3) \
  PA_MD5_STEP(f, a, b, c, d, g0, i, s0)                     \
  PA_MD5_STEP(f, d, a, b, c, g1, i + 1, s1)                 \
  PA_MD5_STEP(f, c, d, a, b, g2, i + 2, s2)                 \
  PA_MD5_STEP(f, b, c, d, a, g3, i + 3, s3)
    PA_MD5_ROUND4(PA_MD5_F, 0, 0, 1, 2, 3, 7, 12, 17, 22)
    PA_MD5_ROUND4(PA_MD5_F, 4, 4, 5, 6, 7, 7, 12, 17, 22)
    PA_MD5_ROUND4(PA_MD5_F, 8, 8, 9, 10, 11, 7, 12, 17, 22)
    PA_MD5_ROUND4(PA_MD5_F, 12, 12, 13, 14, 15, 7, 12, 17, 22)
    PA_MD5_ROUND4(PA_MD5_G, 16, 1, 6, 11, 0, 5, 9, 14, 20)
    PA_MD5_ROUND4(PA_MD5_G, 20, 5, 10, 15, 4, 5, 9, 14, 20)
    PA_MD5_ROUND4(PA_MD5_G, 24, 9, 14, 3, 8, 5, 9, 14, 20)
    PA_MD5_ROUND4(PA_MD5_G, 28, 13, 2, 7, 12, 5, 9, 14, 20)
    PA_MD5_ROUND4(PA_MD5_H, 32, 5, 8, 11, 14, 4, 11, 16, 23)
    PA_MD5_ROUND4(PA_MD5_H, 36, 1, 4, 7, 10, 4, 11, 16, 23)
    PA_MD5_ROUND4(PA_MD5_H, 40, 13, 0, 3, 6, 4, 11, 16, 23)
    PA_MD5_ROUND4(PA_MD5_H, 44, 9, 12, 15, 2, 4, 11, 16, 23)
    PA_MD5_ROUND4(PA_MD5_I, 48, 0, 7, 14, 5, 6, 10, 15, 21)
    PA_MD5_ROUND4(PA_MD5_I, 52, 12, 3, 10, 1, 6, 10, 15, 21)
    PA_MD5_ROUND4(PA_MD5_I, 56, 8, 15, 6, 13, 6, 10, 15, 21)
    PA_MD5_ROUND4(PA_MD5_I, 60, 4, 11, 2, 9, 6, 10, 15, 21)
#undef PA_MD5_ROUND4
#undef PA_MD5_STEP
#undef PA_MD5_I
#undef PA_MD5_H
#undef PA_MD5_G
#undef PA_MD5_F
#undef PA_MD5_ROL
    a += a0; b += b0; c += c0; d += d0;
  }
  st[0] = a; st[1] = b; st[2] = c; st[3] = d;
}

// the last (partial) block plus padding and length, then the digest as 32 hex characters
inline void finish_hex(uint32_t st[4], const uint8_t *tail, size_t n_tail, uint64_t total_bytes, char out[33]) {
  uint8_t buf[128] = {0};
  memcpy(buf, tail, n_tail);
  buf[n_tail] = 0x80;
  const size_t n = n_tail < 56 ? 64 : 128;
  const uint64_t bits = total_bytes * 8;
  for (int i = 0; i < 8; ++i) buf[n - 8 + i] = (uint8_t)(bits >> (8 * i));
  blocks_scalar(st, buf, n / 64);
  for (int i = 0; i < 16; ++i) snprintf(out + 2 * i, 3, "%02x", (st[i / 4] >> (8 * (i % 4))) & 0xffu);
  out[32] = 0;
}

inline void md5_hex_scalar(const uint8_t *p, size_t n, char out[33]) {
  uint32_t st[4] = {kInit[0], kInit[1], kInit[2], kInit[3]};
  blocks_scalar(st, p, n / 64);
  finish_hex(st, p + (n / 64) * 64, n % 64, n, out);
}

inline bool have_avx512() {
  static const bool ok = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw");
  return ok;
}

// ---- sixteen messages side by side: n_blocks blocks of each, states in st[lane][4].  Lane l reads
// ptr[l] + blk * stride[l] (idle lanes: stride 0 on a scratch block); only the lanes of `live` get their state back.
__attribute__((target("avx512f,avx512bw"))) inline void blocks_x16(uint32_t st[16][4], const uint8_t *const ptr[16],
                                                                    const size_t stride[16], size_t n_blocks,
                                                                    uint32_t live) {
  alignas(64) uint32_t tmp[4][16];
  for (int l = 0; l < 16; ++l)
    for (int w = 0; w < 4; ++w) tmp[w][l] = st[l][w];
  __m512i a = _mm512_load_si512(tmp[0]), b = _mm512_load_si512(tmp[1]), c = _mm512_load_si512(tmp[2]),
          d = _mm512_load_si512(tmp[3]);
  for (size_t blk = 0; blk < n_blocks; ++blk) {
    // sixteen 64-byte blocks -> sixteen registers of "word j of every lane": a 16 x 16 transpose of 32-bit words
    __m512i r[16], t[16];
    for (int l = 0; l < 16; ++l) r[l] = _mm512_loadu_si512(ptr[l] + stride[l] * blk);
    for (int i = 0; i < 16; i += 2) {
      t[i] = _mm512_unpacklo_epi32(r[i], r[i + 1]);
      t[i + 1] = _mm512_unpackhi_epi32(r[i], r[i + 1]);
    }
    for (int i = 0; i < 16; i += 4) {
      r[i] = _mm512_unpacklo_epi64(t[i], t[i + 2]);
      r[i + 1] = _mm512_unpackhi_epi64(t[i], t[i + 2]);
      r[i + 2] = _mm512_unpacklo_epi64(t[i + 1], t[i + 3]);
      r[i + 3] = _mm512_unpackhi_epi64(t[i + 1], t[i + 3]);
    }
    // r[4q + j] holds, in 128-bit group g, word 4g + j of lanes 4q .. 4q + 3
    for (int j = 0; j < 4; ++j) {
      t[j] = _mm512_shuffle_i32x4(r[j], r[4 + j], 0x88);       // groups 0, 2 of lanes 0-3 | 0, 2 of lanes 4-7
      t[4 + j] = _mm512_shuffle_i32x4(r[j], r[4 + j], 0xdd);   // groups 1, 3
      t[8 + j] = _mm512_shuffle_i32x4(r[8 + j], r[12 + j], 0x88);
      t[12 + j] = _mm512_shuffle_i32x4(r[8 + j], r[12 + j], 0xdd);
    }
    __m512i m[16];
    for (int j = 0; j < 4; ++j) {
      m[j] = _mm512_shuffle_i32x4(t[j], t[8 + j], 0x88);           // word j      (group 0)
      m[8 + j] = _mm512_shuffle_i32x4(t[j], t[8 + j], 0xdd);       // word 8 + j  (group 2)
      m[4 + j] = _mm512_shuffle_i32x4(t[4 + j], t[12 + j], 0x88);  // word 4 + j  (group 1)
      m[12 + j] = _mm512_shuffle_i32x4(t[4 + j], t[12 + j], 0xdd); // word 12 + j (group 3)
    }
    const __m512i a0 = a, b0 = b, c0 = c, d0 = d;
#define PA_MB_STEP(imm, w, x, y, z, g, i, s)                                                      \
  w = _mm512_add_epi32(_mm512_add_epi32(w, _mm512_ternarylogic_epi32(x, y, z, imm)),              \
                       _mm512_add_epi32(m[g], _mm512_set1_epi32((int)kK[i])));                    \
  w = _mm512_add_epi32(_mm512_rol_epi32(w, s), x);
#define PA_MB_ROUND4(imm, i, g0, g1, g2, g3, s0, s1, s2, s3) \
  PA_MB_STEP(imm, a, b, c, d, g0, i, s0)                     \
  PA_MB_STEP(imm, d, a, b, c, g1, i + 1, s1)                 \
  PA_MB_STEP(imm, c, d, a, b, g2, i + 2, s2)                 \
  PA_MB_STEP(imm, b, c, d, a, g3, i + 3, s3)
    // truth tables over (x, y, z): F = (x & y) | (~x & z) = 0xca, G = (x & z) | (y & ~z) = 0xe4,
    // H = x ^ y ^ z = 0x96, I = y ^ (x | ~z) = 0x39
    PA_MB_ROUND4(0xca, 0, 0, 1, 2, 3, 7, 12, 17, 22)
    PA_MB_ROUND4(0xca, 4, 4, 5, 6, 7, 7, 12, 17, 22)
    PA_MB_ROUND4(0xca, 8, 8, 9, 10, 11, 7, 12, 17, 22)
    PA_MB_ROUND4(0xca, 12, 12, 13, 14, 15, 7, 12, 17, 22)
    PA_MB_ROUND4(0xe4, 16, 1, 6, 11, 0, 5, 9, 14, 20)
    PA_MB_ROUND4(0xe4, 20, 5, 10, 15, 4, 5, 9, 14, 20)
    PA_MB_ROUND4(0xe4, 24, 9, 14, 3, 8, 5, 9, 14, 20)
    PA_MB_ROUND4(0xe4, 28, 13, 2, 7, 12, 5, 9, 14, 20)
    PA_MB_ROUND4(0x96, 32, 5, 8, 11, 14, 4, 11, 16, 23)
    PA_MB_ROUND4(0x96, 36, 1, 4, 7, 10, 4, 11, 16, 23)
    PA_MB_ROUND4(0x96, 40, 13, 0, 3, 6, 4, 11, 16, 23)
    PA_MB_ROUND4(0x96, 44, 9, 12, 15, 2, 4, 11, 16, 23)
    PA_MB_ROUND4(0x39, 48, 0, 7, 14, 5, 6, 10, 15, 21)
    PA_MB_ROUND4(0x39, 52, 12, 3, 10, 1, 6, 10, 15, 21)
    PA_MB_ROUND4(0x39, 56, 8, 15, 6, 13, 6, 10, 15, 21)
    PA_MB_ROUND4(0x39, 60, 4, 11, 2, 9, 6, 10, 15, 21)
#undef PA_MB_ROUND4
#undef PA_MB_STEP
    a = _mm512_add_epi32(a, a0);
    b = _mm512_add_epi32(b, b0);
    c = _mm512_add_epi32(c, c0);
    d = _mm512_add_epi32(d, d0);
  }
  _mm512_store_si512(tmp[0], a);
  _mm512_store_si512(tmp[1], b);
  _mm512_store_si512(tmp[2], c);
  _mm512_store_si512(tmp[3], d);
  for (int l = 0; l < 16; ++l)
    if ((live >> l) & 1u)
      for (int w = 0; w < 4; ++w) st[l][w] = tmp[w][l];
}

// md5 of n buffers (hex digests into out[i]).  With AVX-512 the buffers are taken sixteen at a time, longest first
// so that the lanes of a group have similar lengths; the lanes run side by side as far as the shortest one still in
// the group reaches, which drops out there, and so on -- a lane never waits for more than its own length.
inline void md5_many(const uint8_t *const *data, const size_t *len, uint32_t n, char (*out)[33]) {
  if (!have_avx512() || n < 4) {
    for (uint32_t i = 0; i < n; ++i) md5_hex_scalar(data[i], len[i], out[i]);
    return;
  }
  std::vector<uint32_t> order(n);
  std::iota(order.begin(), order.end(), 0u);
  std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return len[x] > len[y]; });
  static const uint8_t idle_block[64] = {0};
  for (uint32_t g0 = 0; g0 < n; g0 += 16) {
    const uint32_t lanes = std::min<uint32_t>(16u, n - g0);
    uint32_t st[16][4];
    size_t done[16];  // full blocks processed per lane
    for (int l = 0; l < 16; ++l) {
      memcpy(st[l], kInit, sizeof(kInit));
      done[l] = 0;
    }
    // lanes are sorted longest first: lane `live - 1` is the shortest one still running.  With fewer than four
    // lanes left the scalar form is as fast.
    for (uint32_t live = lanes; live >= 4;) {
      const size_t upto = len[order[g0 + live - 1]] / 64;  // every live lane has at least this many full blocks
      const size_t at = done[0];
      if (upto > at) {
        const uint8_t *ptr[16];
        size_t stride[16];
        for (uint32_t l = 0; l < 16; ++l) {
          ptr[l] = l < live ? data[order[g0 + l]] + 64 * at : idle_block;
          stride[l] = l < live ? 64 : 0;
        }
        blocks_x16(st, ptr, stride, upto - at, (1u << live) - 1u);
        for (uint32_t l = 0; l < live; ++l) done[l] = upto;
      }
      while (live > 0 && len[order[g0 + live - 1]] / 64 == done[live - 1]) --live;  // the lanes that end here
    }
    for (uint32_t l = 0; l < lanes; ++l) {
      const uint32_t i = order[g0 + l];
      const size_t full = len[i] / 64;
      if (done[l] < full) blocks_scalar(st[l], data[i] + 64 * done[l], full - done[l]);
      finish_hex(st[l], data[i] + 64 * full, len[i] % 64, len[i], out[i]);
    }
  }
}

}  // namespace md5mb
