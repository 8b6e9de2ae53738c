// inflate_fast.h -- gzip members inflated with 64-bit bit buffers and two-level Huffman tables.
//
// The reference reads compressed genomes through Python's gzip module (pyani_plus/utils.py:178-196); zlib's
// inflate delivers about 0.2 GB/s of FASTA text per core, which on a container with 16 CPUs' worth of time makes
// decompression THE cost of the front-end for .gz input (a 5 Mb genome: 25-30 ms against 1 ms for its checksum and
// 1 ms for packing).  This is the usual modern formulation of RFC 1951 decoding (the design libdeflate made common):
// a bit buffer refilled eight bytes at a time, an 11-bit first-level table for literals/lengths whose entries carry
// the base value and the number of extra bits, word-wise match copies, and a margin-checked fast loop with a fully
// checked loop for the ends of the buffers.
//
// Safety and exactness do not rest on this file alone: every member's CRC-32 and length are checked against its
// trailer (RFC 1952), and the caller falls back to zlib for the whole file on ANY failure here -- malformed input,
// an unusual code, a mismatch.  So a wrong answer needs a CRC-32 collision, and an odd but legal stream still loads.
#pragma once
#include <immintrin.h>

#include <cstdint>
#include <cstring>
#include <vector>

namespace pa_inflate {

constexpr int kLitBits = 11, kDistBits = 8;
constexpr uint32_t kLitTableSize = (1u << kLitBits) + 1024u, kDistTableSize = (1u << kDistBits) + 512u;
// literal kinds first: kind + 1 literals in one entry (see pair_literals)
enum : uint32_t { kLiteral = 0, kLiteral2 = 1, kLiteral3 = 2, kLength = 3, kEnd = 4, kSub = 5, kInvalid = 6 };

// entry: value (literal, length/offset base, or subtable start) << 16 | extra bits (or subtable index bits) << 8
//        | kind << 4 | code length
inline constexpr uint32_t entry(uint32_t value, uint32_t extra, uint32_t kind, uint32_t len) {
  return (value << 16) | (extra << 8) | (kind << 4) | len;
}
inline uint32_t e_value(uint32_t e) { return e >> 16; }
inline uint32_t e_extra(uint32_t e) { return (e >> 8) & 0xffu; }
inline uint32_t e_kind(uint32_t e) { return (e >> 4) & 0xfu; }
inline uint32_t e_len(uint32_t e) { return e & 0xfu; }

constexpr uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
constexpr uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
constexpr uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
constexpr uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

inline uint32_t symbol_entry(bool litlen, uint32_t sym, uint32_t len) {
  if (litlen) {
    if (sym < 256) return entry(sym, 0, kLiteral, len);
    if (sym == 256) return entry(0, 0, kEnd, len);
    if (sym <= 285) return entry(kLenBase[sym - 257], kLenExtra[sym - 257], kLength, len);
    return entry(0, 0, kInvalid, len);
  }
  if (sym < 30) return entry(kDistBase[sym], kDistExtra[sym], kLength, len);
  return entry(0, 0, kInvalid, len);
}

// canonical Huffman code of `n` symbols with the given lengths (0 = unused) -> two-level decode table.
// Over-subscribed codes fail; slots of an incomplete code stay kInvalid (decoding one fails the member).
inline bool build_table(const uint8_t *lens, uint32_t n, bool litlen, uint32_t *table, uint32_t table_bits, uint32_t table_size) {
  uint32_t count[16] = {0};
  for (uint32_t i = 0; i < n; ++i) ++count[lens[i]];
  count[0] = 0;
  uint32_t kraft = 0;
  for (int l = 1; l <= 15; ++l) kraft += count[l] << (15 - l);
  if (kraft > (1u << 15)) return false;
  uint32_t next_code[16], code = 0;
  for (int l = 1; l <= 15; ++l) {
    next_code[l] = code;
    code = (code + count[l]) << 1;
  }
  const uint32_t primary = 1u << table_bits;
  for (uint32_t i = 0; i < primary; ++i) table[i] = entry(0, 0, kInvalid, 0);
  uint16_t rev[288];
  uint8_t sub_max[1u << kLitBits];  // longest code behind each first-level slot (0 = none)
  memset(sub_max, 0, primary);
  for (uint32_t s = 0; s < n; ++s) {
    const uint32_t l = lens[s];
    if (!l) continue;
    uint32_t c = next_code[l]++, r = 0;
    for (uint32_t b = 0; b < l; ++b) { r = (r << 1) | (c & 1u); c >>= 1; }
    rev[s] = (uint16_t)r;
    if (l > table_bits) {
      uint8_t &m = sub_max[r & (primary - 1)];
      if (l > m) m = (uint8_t)l;
    }
  }
  uint32_t used = primary;
  for (uint32_t i = 0; i < primary; ++i) {
    if (!sub_max[i]) continue;
    const uint32_t bits = sub_max[i] - table_bits, size = 1u << bits;
    if (used + size > table_size) return false;
    table[i] = entry(used, bits, kSub, 0);
    for (uint32_t j = 0; j < size; ++j) table[used + j] = entry(0, 0, kInvalid, 0);
    used += size;
  }
  for (uint32_t s = 0; s < n; ++s) {
    const uint32_t l = lens[s];
    if (!l) continue;
    const uint32_t r = rev[s], e = symbol_entry(litlen, s, l);
    if (l <= table_bits) {
      for (uint32_t i = r; i < primary; i += 1u << l) table[i] = e;
    } else {
      const uint32_t ptr = table[r & (primary - 1)], start = e_value(ptr), bits = e_extra(ptr);
      for (uint32_t i = r >> table_bits; i < (1u << bits); i += 1u << (l - table_bits)) table[start + i] = e;
    }
  }
  return true;
}

// Sequences are nearly all literals with codes of two or three bits (A, C, G, T and the line feed), and a decoder
// that looks up one symbol at a time spends a load-to-use latency per base.  The first-level index holds eleven
// bits, so wherever one short literal leaves room for a second and a third the entry is replaced by one that
// carries all of them: value = first | second << 8, the `extra` field = third, length = the sum.
inline void pair_literals(uint32_t *table) {
  constexpr uint32_t primary = 1u << kLitBits;
  static thread_local uint32_t single[primary];
  memcpy(single, table, sizeof(single));
  for (uint32_t i = 0; i < primary; ++i) {
    const uint32_t e1 = single[i];
    if (e_kind(e1) != kLiteral) continue;
    const uint32_t l1 = e_len(e1);
    if (l1 >= (uint32_t)kLitBits) continue;
    const uint32_t i2 = i >> l1, e2 = single[i2];  // the bits above the known ones read as zero: fine while the code fits the known ones
    if (e_kind(e2) != kLiteral || l1 + e_len(e2) > (uint32_t)kLitBits) continue;
    const uint32_t l2 = e_len(e2), i3 = i2 >> l2, e3 = single[i3];
    if (e_kind(e3) == kLiteral && l1 + l2 + e_len(e3) <= (uint32_t)kLitBits)
      table[i] = entry(e_value(e1) | (e_value(e2) << 8), e_value(e3), kLiteral3, l1 + l2 + e_len(e3));
    else
      table[i] = entry(e_value(e1) | (e_value(e2) << 8), 0, kLiteral2, l1 + l2);
  }
}

struct Tables {
  uint32_t lit[kLitTableSize], dist[kDistTableSize];
};

struct BitReader {
  const uint8_t *p, *end;
  uint64_t bits = 0;
  uint32_t n = 0;  // valid bits in `bits`
  inline void refill() {
    if (end - p >= 8) {
      uint64_t w;
      memcpy(&w, p, 8);
      bits |= w << n;
      p += (63 - n) >> 3;
      n |= 56;
    } else {
      while (n <= 56 && p < end) { bits |= (uint64_t)*p++ << n; n += 8; }
    }
  }
  inline bool need(uint32_t k) {  // checked form: false when the input ends first
    if (n < k) refill();
    return n >= k;
  }
  inline uint32_t take(uint32_t k) {
    const uint32_t v = (uint32_t)(bits & ((1ull << k) - 1ull));
    bits >>= k;
    n -= k;
    return v;
  }
  inline void byte_align() {  // give back the whole bytes still in the buffer
    p -= n >> 3;
    bits = 0;
    n = 0;
  }
};

inline uint32_t lookup(const uint32_t *table, uint32_t table_bits, uint64_t bits) {
  uint32_t e = table[bits & ((1u << table_bits) - 1u)];
  if (e_kind(e) == kSub) e = table[e_value(e) + ((uint32_t)(bits >> table_bits) & ((1u << e_extra(e)) - 1u))];
  return e;
}

// the symbols of one Huffman-coded block, appended to out[0 .. pos).  0 = end of block reached, -1 = failure.
// The reader's state and the output cursor live in locals for the length of the block: byte stores may alias
// anything, so state kept behind references would be reloaded after every literal.
inline int decode_block(BitReader &reader, const Tables &t, std::vector<uint8_t> &out, size_t &out_pos, size_t member_start) {
  const uint8_t *p = reader.p, *const end = reader.end;
  uint64_t bits = reader.bits;
  uint32_t n = reader.n;
  size_t pos = out_pos, cap = out.size();
  uint8_t *o = out.data();
  const uint32_t *const lit = t.lit, *const dist = t.dist;
  constexpr uint32_t kLitMask = (1u << kLitBits) - 1u, kDistMask = (1u << kDistBits) - 1u;
  int status = -1;
#define PA_INF_REFILL8()                \
  do {                                  \
    uint64_t w_;                        \
    memcpy(&w_, p, 8);                  \
    bits |= w_ << n;                    \
    p += (63 - n) >> 3;                 \
    n |= 56;                            \
  } while (0)
#define PA_INF_LOOKUP(e, table, mask, tbits)                                                              \
  do {                                                                                                    \
    e = table[bits & mask];                                                                               \
    if (e_kind(e) == kSub) e = table[e_value(e) + ((uint32_t)(bits >> tbits) & ((1u << e_extra(e)) - 1u))]; \
  } while (0)
#define PA_INF_CONSUME(e) \
  do {                    \
    bits >>= e_len(e);    \
    n -= e_len(e);        \
  } while (0)
  for (;;) {
    // ---- fast loop: sixteen bytes of input ahead (both refills take the eight-byte form, so at least 56 bits are
    // valid after each) and room for three literals or the longest match plus the slack of its word-wise copy
    while (end - p >= 16) {
      if (cap - pos < 320) {
        out.resize(cap * 2 + 4096);
        o = out.data();
        cap = out.size();
      }
      PA_INF_REFILL8();
      uint32_t e;
      PA_INF_LOOKUP(e, lit, kLitMask, kLitBits);
      // literals: one to three per entry (kind + 1), written as one four-byte store; three entries per refill (<= 45 bits)
#define PA_INF_LITERALS(e)                                              \
  do {                                                                  \
    const uint32_t w4_ = e_value(e) | (e_extra(e) << 16);               \
    memcpy(o + pos, &w4_, 4);                                           \
    pos += e_kind(e) + 1u;                                              \
  } while (0)
      if ((e & 0xf0u) <= 0x20u) {
        PA_INF_CONSUME(e);
        PA_INF_LITERALS(e);
        PA_INF_LOOKUP(e, lit, kLitMask, kLitBits);
        if ((e & 0xf0u) <= 0x20u) {
          PA_INF_CONSUME(e);
          PA_INF_LITERALS(e);
          PA_INF_LOOKUP(e, lit, kLitMask, kLitBits);
          if ((e & 0xf0u) <= 0x20u) {
            PA_INF_CONSUME(e);
            PA_INF_LITERALS(e);
            continue;
          }
        }
      }
      // not a literal; at most 30 bits used since the refill, 20 more for a length and its extra bits
      if (e_kind(e) == kLength) {
        PA_INF_CONSUME(e);
        const uint32_t length = e_value(e) + (uint32_t)(bits & ((1ull << e_extra(e)) - 1ull));
        bits >>= e_extra(e);
        n -= e_extra(e);
        PA_INF_REFILL8();
        uint32_t d;
        PA_INF_LOOKUP(d, dist, kDistMask, kDistBits);
        if (e_kind(d) != kLength) goto done;
        PA_INF_CONSUME(d);
        const size_t offset = e_value(d) + (size_t)(bits & ((1ull << e_extra(d)) - 1ull));
        bits >>= e_extra(d);
        n -= e_extra(d);
        if (offset > pos - member_start) goto done;  // a match never reaches back past its own member
        uint8_t *dst = o + pos;
        const uint8_t *src = dst - offset;
        if (offset >= 16) {
          memcpy(dst, src, 16);  // may write up to 15 bytes past the match: inside the 320 bytes of room
          for (uint32_t i = 16; i < length; i += 16) memcpy(dst + i, src + i, 16);
        } else if (offset == 1) {
          memset(dst, *src, length);
        } else {
          for (uint32_t i = 0; i < length; ++i) dst[i] = src[i];
        }
        pos += length;
        continue;
      }
      if (e_kind(e) == kEnd) {
        PA_INF_CONSUME(e);
        status = 0;
      }
      goto done;  // end of block, or an invalid code
    }
    // ---- checked loop: the last bytes of the input, every bit accounted for
    for (;;) {
      if (cap - pos < 320) {
        out.resize(cap * 2 + 4096);
        o = out.data();
        cap = out.size();
      }
      while (n <= 56 && p < end) { bits |= (uint64_t)*p++ << n; n += 8; }
      uint32_t e;
      PA_INF_LOOKUP(e, lit, kLitMask, kLitBits);
      if (e_kind(e) == kInvalid || e_kind(e) == kSub || e_len(e) > n) goto done;
      PA_INF_CONSUME(e);
      if (e_kind(e) <= kLiteral3) { PA_INF_LITERALS(e); continue; }
      if (e_kind(e) == kEnd) { status = 0; goto done; }
      if (e_extra(e) > n) goto done;  // (at most 5 bits; the refill above left 57 or the input is exhausted)
      const uint32_t length = e_value(e) + (uint32_t)(bits & ((1ull << e_extra(e)) - 1ull));
      bits >>= e_extra(e);
      n -= e_extra(e);
      while (n <= 56 && p < end) { bits |= (uint64_t)*p++ << n; n += 8; }
      uint32_t d;
      PA_INF_LOOKUP(d, dist, kDistMask, kDistBits);
      if (e_kind(d) != kLength || e_len(d) + e_extra(d) > n) goto done;
      PA_INF_CONSUME(d);
      const size_t offset = e_value(d) + (size_t)(bits & ((1ull << e_extra(d)) - 1ull));
      bits >>= e_extra(d);
      n -= e_extra(d);
      if (offset > pos - member_start) goto done;  // a match never reaches back past its own member
      for (uint32_t i = 0; i < length; ++i) o[pos + i] = o[pos + i - offset];
      pos += length;
    }
  }
done:
#undef PA_INF_LITERALS
#undef PA_INF_CONSUME
#undef PA_INF_LOOKUP
#undef PA_INF_REFILL8
  reader.p = p;
  reader.bits = bits;
  reader.n = n;
  out_pos = pos;
  return status;
}

// raw DEFLATE stream at br -> out[0 .. pos), through the final block.  false = failure (caller falls back to zlib).
inline bool inflate_stream(BitReader &br, std::vector<uint8_t> &out, size_t &pos, Tables &t) {
  const size_t member_start = pos;
  static const uint8_t kOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
  for (;;) {
    if (!br.need(3)) return false;
    const uint32_t final_block = br.take(1), type = br.take(2);
    if (type == 0) {  // stored
      br.byte_align();
      if (br.end - br.p < 4) return false;
      const uint32_t len = br.p[0] | (br.p[1] << 8), nlen = br.p[2] | (br.p[3] << 8);
      if ((len ^ nlen) != 0xffffu) return false;
      br.p += 4;
      if ((size_t)(br.end - br.p) < len) return false;
      if (out.size() - pos < len) out.resize((out.size() + len) * 2);
      memcpy(out.data() + pos, br.p, len);
      pos += len;
      br.p += len;
    } else if (type == 1 || type == 2) {
      uint8_t lens[288 + 32];
      uint32_t n_lit, n_dist;
      if (type == 1) {
        n_lit = 288;
        n_dist = 32;
        for (uint32_t i = 0; i < 144; ++i) lens[i] = 8;
        for (uint32_t i = 144; i < 256; ++i) lens[i] = 9;
        for (uint32_t i = 256; i < 280; ++i) lens[i] = 7;
        for (uint32_t i = 280; i < 288; ++i) lens[i] = 8;
        for (uint32_t i = 0; i < 32; ++i) lens[288 + i] = 5;
      } else {
        if (!br.need(14)) return false;
        n_lit = br.take(5) + 257;
        n_dist = br.take(5) + 1;
        const uint32_t n_pre = br.take(4) + 4;
        if (n_lit > 286 || n_dist > 30) return false;
        uint8_t pre_lens[19] = {0};
        for (uint32_t i = 0; i < n_pre; ++i) {
          if (!br.need(3)) return false;
          pre_lens[kOrder[i]] = (uint8_t)br.take(3);
        }
        uint32_t pre[(1u << 7) + 8];
        if (!build_table(pre_lens, 19, true, pre, 7, 1u << 7)) return false;  // lengths <= 7: no subtables; symbols 0..18 come out as literals
        uint32_t i = 0;
        while (i < n_lit + n_dist) {
          if (!br.need(7 + 7)) {  // code + the longest repeat count, unless the stream ends here
            br.refill();
          }
          const uint32_t e = pre[br.bits & 127u];
          if (e_kind(e) != kLiteral || e_len(e) > br.n) return false;
          br.bits >>= e_len(e);
          br.n -= e_len(e);
          const uint32_t sym = e_value(e);
          if (sym < 16) { lens[i++] = (uint8_t)sym; continue; }
          uint32_t rep, val = 0;
          if (sym == 16) {
            if (i == 0 || !br.need(2)) return false;
            val = lens[i - 1];
            rep = 3 + br.take(2);
          } else if (sym == 17) {
            if (!br.need(3)) return false;
            rep = 3 + br.take(3);
          } else {
            if (!br.need(7)) return false;
            rep = 11 + br.take(7);
          }
          if (i + rep > n_lit + n_dist) return false;
          memset(lens + i, (int)val, rep);
          i += rep;
        }
        if (lens[256] == 0) return false;  // no end-of-block code
        // the two codes are built from separate arrays
        memmove(lens + 288, lens + n_lit, n_dist);
        memset(lens + n_lit, 0, 288 - n_lit);
        memset(lens + 288 + n_dist, 0, 32 - n_dist);
        n_lit = 288;
        n_dist = 32;
      }
      if (!build_table(lens, n_lit, true, t.lit, kLitBits, kLitTableSize)) return false;
      pair_literals(t.lit);
      if (!build_table(lens + 288, n_dist, false, t.dist, kDistBits, kDistTableSize)) return false;
      if (decode_block(br, t, out, pos, member_start) != 0) return false;
    } else {
      return false;
    }
    if (final_block) return true;
  }
}

// ---- CRC-32 (reflected 0xEDB88320), eight bytes per step
struct CrcTables {
  uint32_t t[8][256];
  CrcTables() {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xedb88320u ^ (c >> 1) : c >> 1;
      t[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
      for (int j = 1; j < 8; ++j) t[j][i] = (t[j - 1][i] >> 8) ^ t[0][t[j - 1][i] & 0xffu];
  }
};
// carry-less-multiply folding (Gopal et al., "Fast CRC computation for generic polynomials using PCLMULQDQ"; the
// constants are x^k mod P for the reflected gzip polynomial): 64 bytes per step, state in and out without the final
// inversion.  n >= 64 and a multiple of 16.  Held to the table form in tests/test_host_logic.py.
__attribute__((target("pclmul,sse4.1"))) inline uint32_t crc32_fold(uint32_t crc, const uint8_t *buf, size_t n) {
  const __m128i r2r1 = _mm_set_epi64x(0x00000001c6e41596LL, 0x0000000154442bd4LL);
  const __m128i r4r3 = _mm_set_epi64x(0x00000000ccaa009eLL, 0x00000001751997d0LL);
  const __m128i r5 = _mm_set_epi64x(0, 0x0000000163cd6124LL);
  const __m128i mask32 = _mm_set_epi32(0, 0, 0, -1);
  const __m128i ru_poly = _mm_set_epi64x(0x00000001F7011641LL, 0x00000001DB710641LL);
#define PA_CRC_LD(q) _mm_loadu_si128(reinterpret_cast<const __m128i *>(q))
#define PA_CRC_FOLD(x, k, next) \
  _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x, k, 0x00), _mm_clmulepi64_si128(x, k, 0x11)), next)
  __m128i x1 = PA_CRC_LD(buf), x2 = PA_CRC_LD(buf + 16), x3 = PA_CRC_LD(buf + 32), x4 = PA_CRC_LD(buf + 48);
  x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)crc));
  buf += 64;
  n -= 64;
  while (n >= 64) {
    x1 = PA_CRC_FOLD(x1, r2r1, PA_CRC_LD(buf));
    x2 = PA_CRC_FOLD(x2, r2r1, PA_CRC_LD(buf + 16));
    x3 = PA_CRC_FOLD(x3, r2r1, PA_CRC_LD(buf + 32));
    x4 = PA_CRC_FOLD(x4, r2r1, PA_CRC_LD(buf + 48));
    buf += 64;
    n -= 64;
  }
  x1 = PA_CRC_FOLD(x1, r4r3, x2);
  x1 = PA_CRC_FOLD(x1, r4r3, x3);
  x1 = PA_CRC_FOLD(x1, r4r3, x4);
  while (n >= 16) {
    x1 = PA_CRC_FOLD(x1, r4r3, PA_CRC_LD(buf));
    buf += 16;
    n -= 16;
  }
#undef PA_CRC_FOLD
#undef PA_CRC_LD
  x1 = _mm_xor_si128(_mm_srli_si128(x1, 8), _mm_clmulepi64_si128(r4r3, x1, 0x01));  // 128 -> 64 bits
  const __m128i upper = _mm_srli_si128(x1, 4);                                         // 64 -> 32 bits
  x1 = _mm_xor_si128(_mm_clmulepi64_si128(_mm_and_si128(x1, mask32), r5, 0x00), upper);
  const __m128i keep = x1;  // Barrett reduction
  x1 = _mm_and_si128(_mm_clmulepi64_si128(_mm_and_si128(x1, mask32), ru_poly, 0x10), mask32);
  x1 = _mm_xor_si128(_mm_clmulepi64_si128(x1, ru_poly, 0x00), keep);
  return (uint32_t)_mm_extract_epi32(x1, 1);
}

// CRC-32 as gzip defines it.  `use_clmul` < 0: decide from the CPU.
inline uint32_t crc32(const uint8_t *p, size_t n, int use_clmul = -1) {
  static const CrcTables tab;
  static const bool have_clmul = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
  uint32_t c = 0xffffffffu;
  if ((use_clmul < 0 ? have_clmul : (use_clmul > 0 && have_clmul)) && n >= 64) {
    const size_t head = n & ~(size_t)15;
    c = crc32_fold(c, p, head);
    p += head;
    n -= head;
  }
  while (n >= 8) {
    uint64_t w;
    memcpy(&w, p, 8);
    w ^= c;
    c = tab.t[7][w & 0xff] ^ tab.t[6][(w >> 8) & 0xff] ^ tab.t[5][(w >> 16) & 0xff] ^ tab.t[4][(w >> 24) & 0xff] ^
        tab.t[3][(w >> 32) & 0xff] ^ tab.t[2][(w >> 40) & 0xff] ^ tab.t[1][(w >> 48) & 0xff] ^ tab.t[0][w >> 56];
    p += 8;
    n -= 8;
  }
  while (n--) c = tab.t[0][(c ^ *p++) & 0xffu] ^ (c >> 8);
  return ~c;
}

// One or more gzip members at [in, in + n) (zero padding after the last accepted, as Python's gzip module does) ->
// `out` (replaced).  false = anything unexpected: the caller then runs zlib over the same bytes.
inline bool gunzip_all(const uint8_t *in, size_t n, std::vector<uint8_t> &out) {
  size_t pos = 0;
  const uint8_t *p = in, *const end = in + n;
  // the trailer of the last member holds its length mod 2^32: a good first size for single-member files
  size_t guess = n * 4 + 65536;
  if (n >= 18) {
    const uint8_t *t = end - 4;
    const size_t isize = (size_t)t[0] | ((size_t)t[1] << 8) | ((size_t)t[2] << 16) | ((size_t)t[3] << 24);
    if (isize > n / 2 && isize < n * 1100) guess = isize + 512;
  }
  if (out.size() < guess) out.resize(guess);
  std::vector<Tables> tables(1);  // 14 KB: off the stack
  bool any = false;
  while (p < end) {
    if (any) {
      bool only_zeros = true;
      for (const uint8_t *q = p; q < end && only_zeros; ++q) only_zeros = *q == 0;
      if (only_zeros) break;
    }
    if (end - p < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8) return false;
    const uint32_t flags = p[3];
    if (flags & 0xe0u) return false;
    p += 10;
    if (flags & 4u) {
      if (end - p < 2) return false;
      const size_t xlen = p[0] | (p[1] << 8);
      if ((size_t)(end - p) < 2 + xlen) return false;
      p += 2 + xlen;
    }
    for (uint32_t bit : {8u, 16u}) {
      if (!(flags & bit)) continue;
      while (p < end && *p) ++p;
      if (p == end) return false;
      ++p;
    }
    if (flags & 2u) p += 2;
    if (end - p < 8) return false;
    BitReader br{p, end};
    const size_t start = pos;
    if (!inflate_stream(br, out, pos, tables[0])) return false;
    br.byte_align();
    p = br.p;
    if (end - p < 8) return false;
    const uint32_t want_crc = p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24);
    const uint32_t want_len = p[4] | (p[5] << 8) | (p[6] << 16) | ((uint32_t)p[7] << 24);
    p += 8;
    if ((uint32_t)(pos - start) != want_len || crc32(out.data() + start, pos - start) != want_crc) return false;
    any = true;
  }
  if (!any) return false;
  out.resize(pos);
  return true;
}

}  // namespace pa_inflate
