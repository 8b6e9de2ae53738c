// md5.h -- MD5 (RFC 1321), host side: genome identity is the md5 of the decompressed FASTA bytes
// (pyani_plus/utils.py:142-196) and a sketch's `md5sum` is the md5 of its k and decimal hashes.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>

namespace {

// ---- MD5 (RFC 1321) ---------------------------------------------------------
struct Md5 {
  uint32_t a = 0x67452301u, b = 0xefcdab89u, c = 0x98badcfeu, d = 0x10325476u;
  uint64_t total = 0;
  uint8_t buf[64];
  size_t fill = 0;

  static inline uint32_t rol(uint32_t x, int s) { return (x << s) | (x >> (32 - s)); }

  void block(const uint8_t *p) {
    static const uint32_t K[64] = {
        0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501,
        0x698098d8, 0x8b44f7af, 0xffff5bb1, 0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821,
        0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa, 0xd62f105d, 0x02441453, 0xd8a1e681, 0xe7d3fbc8,
        0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8, 0x676f02d9, 0x8d2a4c8a,
        0xfffa3942, 0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70,
        0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05, 0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665,
        0xf4292244, 0x432aff97, 0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92, 0xffeff47d, 0x85845dd1,
        0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1, 0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391};
    static const int S[64] = {7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 5, 9,  14, 20, 5, 9,
                              14, 20, 5, 9,  14, 20, 5, 9,  14, 20, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23,
                              4, 11, 16, 23, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21};
    uint32_t m[16];
    for (int i = 0; i < 16; ++i)
      m[i] = (uint32_t)p[4 * i] | ((uint32_t)p[4 * i + 1] << 8) | ((uint32_t)p[4 * i + 2] << 16) |
             ((uint32_t)p[4 * i + 3] << 24);
    uint32_t A = a, B = b, C = c, D = d;
    for (int i = 0; i < 64; ++i) {
      uint32_t f;
      int g;
      if (i < 16) { f = (B & C) | (~B & D); g = i; }
      else if (i < 32) { f = (D & B) | (~D & C); g = (5 * i + 1) & 15; }
      else if (i < 48) { f = B ^ C ^ D; g = (3 * i + 5) & 15; }
      else { f = C ^ (B | ~D); g = (7 * i) & 15; }
      const uint32_t t = D;
      D = C;
      C = B;
      B = B + rol(A + f + K[i] + m[g], S[i]);
      A = t;
    }
    a += A; b += B; c += C; d += D;
  }

  void update(const uint8_t *p, size_t n) {
    total += n;
    if (fill) {
      const size_t take = n < 64 - fill ? n : 64 - fill;
      memcpy(buf + fill, p, take);
      fill += take; p += take; n -= take;
      if (fill == 64) { block(buf); fill = 0; }
    }
    while (n >= 64) { block(p); p += 64; n -= 64; }
    if (n) { memcpy(buf, p, n); fill = n; }
  }

  void hex(char out[33]) {
    const uint64_t bits = total * 8;
    uint8_t pad[72] = {0x80};
    const size_t padlen = (fill < 56) ? 56 - fill : 120 - fill;
    uint8_t len[8];
    for (int i = 0; i < 8; ++i) len[i] = (uint8_t)(bits >> (8 * i));
    const uint64_t keep = total;
    update(pad, padlen);
    update(len, 8);
    total = keep;
    const uint32_t w[4] = {a, b, c, d};
    for (int i = 0; i < 16; ++i) snprintf(out + 2 * i, 3, "%02x", (w[i / 4] >> (8 * (i % 4))) & 0xffu);
    out[32] = 0;
  }
};


}  // namespace
