// pack_host.cpp -- host side of the boundary: FASTA text -> 2-bit arena.
//
// Replaces the FASTA reader inside `sourmash scripts singlesketch`
// (pyani_plus/methods/sourmash.py:67-83).  Record and whitespace semantics are
// those of pyani_plus/utils.py:67-90 (fasta_bytes_iterator): text before the
// first '>' line is ignored, a '>' only starts a record at the start of a line,
// " \t\r\n" are dropped from sequence lines.  Residues are case-insensitive;
// anything outside ACGT becomes an invalid position, and one invalid position
// is written between records so that no k-mer window spans two records.
#include <immintrin.h>

#include <cmath>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>
#include <unistd.h>

#include "../../include/pyani_hip.h"
#include "host_pool.h"

void pa_set_error(const char *fmt, ...);

namespace {

struct ArenaWriter {
  uint32_t *packed, *mask;
  uint64_t cap, pos = 0;
  uint32_t pw = 0, mw = 0;
  bool overflow = false;
  // Residues that are neither ACGT nor N (IUPAC codes and anything else a file holds), as (position, upper-cased byte):
  // the arena keeps one "not ACGT" bit per residue, which stands for N; fastANI hashes the characters as they are
  // (pyani_plus/private_cli.py:1044-1063 hands it the FASTA text), so the fragment-ANI kernels take these from the list.
  std::vector<uint64_t> *amb_pos = nullptr;
  std::vector<uint8_t> *amb_byte = nullptr;
  inline void put_other(uint8_t ch) {
    if (amb_pos) {
      const uint8_t up = (ch >= 'a' && ch <= 'z') ? (uint8_t)(ch - 'a' + 'A') : ch;
      if (up != 'N') { amb_pos->push_back(pos); amb_byte->push_back(up); }
    }
    put(0, 1);
  }
  inline void put(uint32_t code, uint32_t invalid) {
    if (pos >= cap) { overflow = true; ++pos; return; }
    pw |= code << (2 * (pos & 15));
    mw |= invalid << (pos & 31);
    ++pos;
    if ((pos & 15) == 0) { packed[(pos >> 4) - 1] = pw; pw = 0; }
    if ((pos & 31) == 0) { mask[(pos >> 5) - 1] = mw; mw = 0; }
  }
  // Always at least one invalid position, then up to the next multiple of 64: the kernel's
  // 32-base look-back into the previous block must never see a neighbouring genome's bases.
  inline void pad64() {
    put(0, 1);
    while (pos & 63) put(0, 1);
  }
};

struct Lut {
  uint8_t v[256];
  constexpr Lut() : v() {
    for (int i = 0; i < 256; ++i) v[i] = 4;  // invalid residue
    v[(int)'A'] = v[(int)'a'] = 0;
    v[(int)'C'] = v[(int)'c'] = 1;
    v[(int)'G'] = v[(int)'g'] = 2;
    v[(int)'T'] = v[(int)'t'] = 3;
    v[(int)' '] = v[(int)'\t'] = v[(int)'\r'] = v[(int)'\n'] = 5;  // dropped
  }
};
constexpr Lut kLut;

}  // namespace

extern "C" uint32_t pa_host_cpu_budget(void) { return pa_cpu_budget(); }

extern "C" uint64_t pa_pack_bound(uint64_t n_text_bytes) { return (n_text_bytes / 64 + 1) * 64 + 64; }

// Runs of set bits of an invalid-position mask, ascending (word scan: zero words cost one compare).  The words are
// scanned in chunks on the host pool (625 MB of mask per 5 Gb arena: 0.1 s on one core); a run that crosses a chunk
// boundary comes out of both chunks and is joined when the lists are put together.
namespace {
void mask_runs_range(const uint32_t *h_mask, uint64_t w0, uint64_t w1, std::vector<uint64_t> &starts, std::vector<uint64_t> &lens) {
  uint64_t start = 0;
  bool in_run = false;
  for (uint64_t w = w0; w < w1; ++w) {
    const uint32_t x = h_mask[w];
    if (!in_run && x == 0) continue;
    if (in_run && x == 0xffffffffu) continue;
    for (uint32_t b = 0; b < 32; ++b) {
      const bool bit = (x >> b) & 1u;
      if (bit && !in_run) { in_run = true; start = w * 32 + b; }
      else if (!bit && in_run) {
        in_run = false;
        starts.push_back(start);
        lens.push_back(w * 32 + b - start);
      }
    }
  }
  if (in_run) {
    starts.push_back(start);
    lens.push_back(w1 * 32 - start);
  }
}
}  // namespace

static int64_t mask_runs_impl(const uint32_t *h_mask, uint64_t arena_bases, uint64_t *h_run_start,
                             uint64_t *h_run_len, uint64_t cap) {
  if (!h_mask && arena_bases) return -1;
  const uint64_t n_words = arena_bases / 32;
  const uint32_t nt = pa_host_threads(n_words, 4u << 20, 0);
  std::vector<std::vector<uint64_t>> starts(nt), lens(nt);
  HostPool::get().run(nt, [&](uint32_t t, uint32_t n_workers) {
    const uint64_t w0 = n_words * t / n_workers, w1 = n_words * (t + 1) / n_workers;
    mask_runs_range(h_mask, w0, w1, starts[t], lens[t]);
  });
  uint64_t n = 0, last_end = ~0ULL;  // end of the last run emitted (to join runs across chunk boundaries)
  uint64_t last_slot = 0;
  for (uint32_t t = 0; t < nt; ++t) {
    for (size_t i = 0; i < starts[t].size(); ++i) {
      const uint64_t s0 = starts[t][i], l0 = lens[t][i];
      if (n && s0 == last_end) {  // continues the previous run
        if (last_slot < cap && h_run_start && h_run_len) h_run_len[last_slot] += l0;
        last_end += l0;
        continue;
      }
      if (n < cap && h_run_start && h_run_len) { h_run_start[n] = s0; h_run_len[n] = l0; }
      last_slot = n;
      last_end = s0 + l0;
      ++n;
    }
  }
  return (int64_t)n;
}

extern "C" int64_t pa_mask_runs(const uint32_t *h_mask, uint64_t arena_bases, uint64_t *h_run_start,
                                uint64_t *h_run_len, uint64_t cap) {
  int64_t n = -1;
  const int st = pa_host_guard("pa_mask_runs", pa_set_error, [&] {
    n = mask_runs_impl(h_mask, arena_bases, h_run_start, h_run_len, cap);
    return 0;
  });
  return st == 0 ? n : -1;  // a negative count is this entry point's failure value
}

extern "C" uint64_t pa_max_hash(uint64_t scaled) {
  if (scaled == 0) return 0;
  if (scaled == 1) return UINT64_MAX;
  return (uint64_t)(18446744073709551616.0 / (double)scaled);
}

namespace {

// ---- runs of plain ACGT/acgt, 32 or 16 characters at a time (AVX2) -------------------------------------------
// A chunk whose characters are all ACGT in either case becomes 2 bits per base with a handful of vector
// instructions (compare against the four letters, combine the codes with two multiply-adds); any other chunk --
// N runs, IUPAC codes, blanks inside a line -- goes through the per-character table below, as every chunk does
// on a CPU without AVX2.  The appended bits are what ArenaWriter::put would have produced one base at a time.
__attribute__((target("avx2"))) inline bool codes32(const uint8_t *p, uint64_t *bits) {
  const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(p));
  const __m256i u = _mm256_and_si256(v, _mm256_set1_epi8((char)0xdf));  // fold case: only a letter or letter|0x20 folds to it
  const __m256i is_c = _mm256_cmpeq_epi8(u, _mm256_set1_epi8('C')), is_g = _mm256_cmpeq_epi8(u, _mm256_set1_epi8('G')),
                is_t = _mm256_cmpeq_epi8(u, _mm256_set1_epi8('T')), is_a = _mm256_cmpeq_epi8(u, _mm256_set1_epi8('A'));
  const __m256i valid = _mm256_or_si256(_mm256_or_si256(is_a, is_c), _mm256_or_si256(is_g, is_t));
  if ((uint32_t)_mm256_movemask_epi8(valid) != 0xffffffffu) return false;
  const __m256i code = _mm256_or_si256(_mm256_or_si256(_mm256_and_si256(is_c, _mm256_set1_epi8(1)), _mm256_and_si256(is_g, _mm256_set1_epi8(2))),
                                       _mm256_and_si256(is_t, _mm256_set1_epi8(3)));
  const __m256i pairs = _mm256_maddubs_epi16(code, _mm256_set1_epi16(0x0401));   // c0 + 4 c1 per 16-bit lane
  const __m256i quads = _mm256_madd_epi16(pairs, _mm256_set1_epi32(0x00100001));  // + 16 (c2 + 4 c3): one byte per 4 bases
  const __m256i gather = _mm256_shuffle_epi8(quads, _mm256_setr_epi8(0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1,
                                                                     0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1));
  *bits = (uint64_t)(uint32_t)_mm256_extract_epi32(gather, 0) | ((uint64_t)(uint32_t)_mm256_extract_epi32(gather, 4) << 32);
  return true;
}
__attribute__((target("avx2"))) inline bool codes16(const uint8_t *p, uint32_t *bits) {
  const __m128i v = _mm_loadu_si128(reinterpret_cast<const __m128i *>(p));
  const __m128i u = _mm_and_si128(v, _mm_set1_epi8((char)0xdf));
  const __m128i is_c = _mm_cmpeq_epi8(u, _mm_set1_epi8('C')), is_g = _mm_cmpeq_epi8(u, _mm_set1_epi8('G')),
                is_t = _mm_cmpeq_epi8(u, _mm_set1_epi8('T')), is_a = _mm_cmpeq_epi8(u, _mm_set1_epi8('A'));
  const __m128i valid = _mm_or_si128(_mm_or_si128(is_a, is_c), _mm_or_si128(is_g, is_t));
  if (_mm_movemask_epi8(valid) != 0xffff) return false;
  const __m128i code = _mm_or_si128(_mm_or_si128(_mm_and_si128(is_c, _mm_set1_epi8(1)), _mm_and_si128(is_g, _mm_set1_epi8(2))),
                                    _mm_and_si128(is_t, _mm_set1_epi8(3)));
  const __m128i pairs = _mm_maddubs_epi16(code, _mm_set1_epi16(0x0401));
  const __m128i quads = _mm_madd_epi16(pairs, _mm_set1_epi32(0x00100001));
  const __m128i gather = _mm_shuffle_epi8(quads, _mm_setr_epi8(0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1));
  *bits = (uint32_t)_mm_cvtsi128_si32(gather);
  return true;
}

// 32 (16) valid bases at once: the bits go where put() would have put them, the mask gets zeros
inline void put_valid32(ArenaWriter &w, uint64_t bits) {
  const uint32_t sh = 2u * (uint32_t)(w.pos & 15), word = (uint32_t)(w.pos >> 4);
  w.packed[word] = w.pw | (uint32_t)(bits << sh);
  w.packed[word + 1] = (uint32_t)(bits >> (32u - sh));
  w.pw = sh ? (uint32_t)(bits >> (64u - sh)) : 0u;
  w.mask[w.pos >> 5] = w.mw;  // the 32 new positions complete this mask word and leave the next one empty so far
  w.mw = 0;
  w.pos += 32;
}
inline void put_valid16(ArenaWriter &w, uint32_t bits) {
  const uint32_t sh = 2u * (uint32_t)(w.pos & 15), word = (uint32_t)(w.pos >> 4);
  w.packed[word] = w.pw | (bits << sh);  // sh == 0: the word is complete and pw restarts at 0
  w.pw = sh ? bits >> (32u - sh) : 0u;
  if ((w.pos & 31) >= 16) { w.mask[w.pos >> 5] = w.mw; w.mw = 0; }
  w.pos += 16;
}

// the leading run of whole clean chunks of [p, e): appended; returns how many characters that was
__attribute__((target("avx2"))) size_t clean_run_avx2(ArenaWriter &w, const uint8_t *p, const uint8_t *e) {
  const uint8_t *const p0 = p;
  while (e - p >= 32 && w.pos + 32 <= w.cap) {
    uint64_t bits;
    if (!codes32(p, &bits)) break;
    put_valid32(w, bits);
    p += 32;
  }
  while (e - p >= 16 && w.pos + 16 <= w.cap) {
    uint32_t bits;
    if (!codes16(p, &bits)) break;
    put_valid16(w, bits);
    p += 16;
  }
  return (size_t)(p - p0);
}

// the residues of one sequence line [p, e) (no line feed inside): the per-character rules of pa_pack_fasta
template <bool kVector>
inline void pack_line(ArenaWriter &w, const uint8_t *p, const uint8_t *e, uint64_t &residues, uint64_t &invalid) {
  while (p < e) {
    if constexpr (kVector) {
      if (e - p >= 16) {
        const size_t n = clean_run_avx2(w, p, e);
        residues += n;
        p += n;
      }
    }
    // up to the end of the line, or (vector form) past the next character that is not a plain base: an N run or a
    // blank interrupts a clean stretch, after it whole chunks are tried again
    for (; p < e; ++p) {
      const uint8_t code = kLut.v[*p];
      if (code < 4) { w.put(code, 0); ++residues; continue; }
      if (code == 4) { w.put_other(*p); ++residues; ++invalid; }
      if (kVector && e - p > 32 && kLut.v[p[1]] < 4) { ++p; break; }
    }
  }
}

template <bool kVector>
int pack_fasta_impl(const uint8_t *h_text, uint64_t n_text, uint32_t *h_packed, uint32_t *h_mask, uint64_t cap_bases,
                    uint64_t *n_bases, uint64_t *n_residues, uint64_t *n_records, uint64_t *n_invalid,
                    std::vector<uint64_t> *rec_start, std::vector<uint64_t> *rec_len, std::vector<uint64_t> *amb_pos,
                    std::vector<uint8_t> *amb_byte) {
  ArenaWriter w{h_packed, h_mask, cap_bases};
  if (amb_pos && amb_byte) { w.amb_pos = amb_pos; w.amb_byte = amb_byte; }
  uint64_t residues = 0, records = 0, invalid = 0, record_first_residue = 0;
  const uint8_t *p = h_text, *const end = h_text + n_text;
  bool in_record = false;
  while (p < end) {
    const uint8_t *nl = static_cast<const uint8_t *>(memchr(p, '\n', (size_t)(end - p)));
    const uint8_t *e = nl ? nl : end;
    if (*p == '>') {  // title line (we are at the start of a line)
      if (in_record) {
        if (rec_len) rec_len->push_back(residues - record_first_residue);
        w.put(0, 1);  // separator: windows never span records
      }
      in_record = true;
      ++records;
      if (rec_start) rec_start->push_back(w.pos);  // positions count the one-position separators
      record_first_residue = residues;
    } else if (in_record) {  // a sequence line; text before the first record is ignored
      pack_line<kVector>(w, p, e, residues, invalid);
    }
    p = nl ? nl + 1 : end;
  }
  if (in_record && rec_len) rec_len->push_back(residues - record_first_residue);
  w.pad64();
  if (w.overflow) {
    pa_set_error("pa_pack_fasta: arena capacity %llu bases is too small (need %llu)", (unsigned long long)cap_bases,
                 (unsigned long long)w.pos);
    if (n_bases) *n_bases = w.pos;
    return PA_E_CAPACITY;
  }
  if (n_bases) *n_bases = w.pos;
  if (n_residues) *n_residues = residues;
  if (n_records) *n_records = records;
  if (n_invalid) *n_invalid = invalid;
  return PA_OK;
}

}  // namespace

// pa_pack_fasta plus the record table of pa_fasta_records from the same pass (the FASTA loader's form), and the list of
// residues that are neither ACGT nor N (positions relative to the genome's first)
int pa_pack_fasta_records(const uint8_t *h_text, uint64_t n_text, uint32_t *h_packed, uint32_t *h_mask, uint64_t cap_bases,
                          uint64_t *n_bases, uint64_t *n_residues, uint64_t *n_records, uint64_t *n_invalid,
                          std::vector<uint64_t> *rec_start, std::vector<uint64_t> *rec_len, std::vector<uint64_t> *amb_pos,
                          std::vector<uint8_t> *amb_byte) {
  if ((!h_text && n_text) || !h_packed || !h_mask || (cap_bases & 63)) {
    pa_set_error("pa_pack_fasta: null buffer or capacity %llu not a multiple of 64", (unsigned long long)cap_bases);
    return PA_E_INVALID;
  }
  static const bool avx2 = __builtin_cpu_supports("avx2") && getenv("PA_PACK_SCALAR") == nullptr;
  return avx2 ? pack_fasta_impl<true>(h_text, n_text, h_packed, h_mask, cap_bases, n_bases, n_residues, n_records, n_invalid, rec_start, rec_len, amb_pos, amb_byte)
              : pack_fasta_impl<false>(h_text, n_text, h_packed, h_mask, cap_bases, n_bases, n_residues, n_records, n_invalid, rec_start, rec_len, amb_pos, amb_byte);
}

extern "C" int pa_pack_fasta(const uint8_t *h_text, uint64_t n_text, uint32_t *h_packed, uint32_t *h_mask,
                             uint64_t cap_bases, uint64_t *n_bases, uint64_t *n_residues, uint64_t *n_records,
                             uint64_t *n_invalid) {
  return pa_pack_fasta_records(h_text, n_text, h_packed, h_mask, cap_bases, n_bases, n_residues, n_records, n_invalid,
                               nullptr, nullptr, nullptr, nullptr);
}

// The residues of a text that are neither ACGT nor N, as the packers above place them: (arena position relative to the
// genome's first, upper-cased byte), ascending.  `fasta`: the text is FASTA (pa_pack_fasta's rules) or a bare sequence
// (pa_pack_seq's).  Returns how many there are (more than `cap`: only the first `cap` were written), negative on failure.
extern "C" int64_t pa_text_ambiguous(const uint8_t *h_text, uint64_t n_text, int fasta, uint64_t *h_pos, uint8_t *h_byte,
                                     uint64_t cap) {
  if ((!h_text && n_text) || (cap && (!h_pos || !h_byte))) { pa_set_error("pa_text_ambiguous: null argument"); return -1; }
  int64_t found = -1;
  const int st = pa_host_guard("pa_text_ambiguous", pa_set_error, [&] {
    std::vector<uint64_t> pos;
    std::vector<uint8_t> bytes;
    if (fasta) {
      const uint64_t room = pa_pack_bound(n_text);
      std::vector<uint32_t> packed(room / 16 + 2), mask(room / 32 + 2);
      uint64_t nb = 0, nr = 0, nrec = 0, ninv = 0;
      const int rc = pa_pack_fasta_records(h_text, n_text, packed.data(), mask.data(), room, &nb, &nr, &nrec, &ninv, nullptr, nullptr, &pos, &bytes);
      if (rc != PA_OK) return rc;
    } else {
      for (uint64_t i = 0; i < n_text; ++i) {
        if (kLut.v[h_text[i]] < 4) continue;  // (a bare sequence: every byte is a residue, blanks included -- pa_pack_seq)
        const uint8_t ch = h_text[i], up = (ch >= 'a' && ch <= 'z') ? (uint8_t)(ch - 'a' + 'A') : ch;
        if (up != 'N') { pos.push_back(i); bytes.push_back(up); }
      }
    }
    for (size_t i = 0; i < pos.size() && i < cap; ++i) { h_pos[i] = pos[i]; h_byte[i] = bytes[i]; }
    found = (int64_t)pos.size();
    return (int)PA_OK;
  });
  return st == PA_OK ? found : -1;
}

extern "C" int64_t pa_fasta_records(const uint8_t *h_text, uint64_t n_text, uint64_t *h_rec_start,
                                    uint64_t *h_rec_len, uint64_t cap) {
  // same line/record rules as pa_pack_fasta; positions count the one-position separators
  uint64_t pos = 0, i = 0, n_rec = 0, cur_len = 0;
  bool in_record = false;
  while (i < n_text) {
    if (h_text[i] == '>') {
      if (in_record) {
        if (n_rec - 1 < cap && h_rec_len) h_rec_len[n_rec - 1] = cur_len;
        ++pos;  // separator
      }
      in_record = true;
      if (n_rec < cap && h_rec_start) h_rec_start[n_rec] = pos;
      ++n_rec;
      cur_len = 0;
      while (i < n_text && h_text[i] != '\n') ++i;
      if (i < n_text) ++i;
      continue;
    }
    if (!in_record) {
      while (i < n_text && h_text[i] != '\n') ++i;
      if (i < n_text) ++i;
      continue;
    }
    while (i < n_text) {
      const uint8_t ch = h_text[i++];
      if (kLut.v[ch] <= 4) { ++pos; ++cur_len; }
      else if (ch == '\n') break;
    }
  }
  if (in_record && n_rec - 1 < cap && h_rec_len) h_rec_len[n_rec - 1] = cur_len;
  return (int64_t)n_rec;
}

extern "C" int pa_pack_seq(const uint8_t *h_seq, uint64_t n_seq, uint32_t *h_packed, uint32_t *h_mask,
                           uint64_t cap_bases, uint64_t *n_bases, uint64_t *n_invalid) {
  if ((!h_seq && n_seq) || !h_packed || !h_mask || (cap_bases & 63)) {
    pa_set_error("pa_pack_seq: null buffer or capacity %llu not a multiple of 64", (unsigned long long)cap_bases);
    return PA_E_INVALID;
  }
  ArenaWriter w{h_packed, h_mask, cap_bases};
  uint64_t invalid = 0;
  for (uint64_t i = 0; i < n_seq; ++i) {
    const uint8_t code = kLut.v[h_seq[i]];
    if (code < 4) w.put(code, 0);
    else { w.put(0, 1); ++invalid; }
  }
  w.pad64();
  if (w.overflow) {
    pa_set_error("pa_pack_seq: arena capacity %llu bases is too small (need %llu)", (unsigned long long)cap_bases,
                 (unsigned long long)w.pos);
    if (n_bases) *n_bases = w.pos;
    return PA_E_CAPACITY;
  }
  if (n_bases) *n_bases = w.pos;
  if (n_invalid) *n_invalid = invalid;
  return PA_OK;
}

namespace {
// eight pairs with no common hash: identity and cov_query NaN, is_null 1 -- or false when any of the eight counts is not zero
__attribute__((target("avx2"))) inline bool null_run8(const uint32_t *counts, double *identity, double *cov, uint8_t *is_null) {
  const __m256i c = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(counts));
  if (!_mm256_testz_si256(c, c)) return false;
  const __m256d nan = _mm256_set1_pd(NAN);
  _mm256_storeu_pd(identity, nan);
  _mm256_storeu_pd(identity + 4, nan);
  _mm256_storeu_pd(cov, nan);
  _mm256_storeu_pd(cov + 4, nan);
  if (is_null) *reinterpret_cast<uint64_t *>(is_null) = 0x0101010101010101ULL;  // (unaligned store of eight bytes: fine on x86)
  return true;
}
}  // namespace

// Strict containment-ANI transform: host libm `pow`, the arithmetic that reproduces every reference fixture
// bit for bit.  Rows are split over host threads; with `symmetric` (queries and
// subjects are the same genomes in the same order) the match-side value (I/|S|)^(1/k) of pair (q, s) is the
// query-side value of pair (s, q) -- same integers, same division, same pow -- so one pow per ordered pair
// is computed and identity = max(cov[q][s], cov[s][q]) is taken in a second pass.
extern "C" int pa_ani_host(const uint32_t *h_counts, const uint64_t *h_q_sizes, const uint64_t *h_s_sizes,
                           uint32_t nq, uint32_t ns, uint32_t k, double *h_identity, double *h_cov_query,
                           uint8_t *h_is_null, int symmetric, uint32_t n_threads) {
  if (!h_counts || !h_q_sizes || !h_s_sizes || !h_identity || !h_cov_query || k == 0) {
    pa_set_error("pa_ani_host: null argument or k == 0");
    return PA_E_INVALID;
  }
  if (symmetric && nq != ns) {
    pa_set_error("pa_ani_host: symmetric needs a square block, got %u x %u", nq, ns);
    return PA_E_INVALID;
  }
  const double inv_k = 1.0 / (double)k;
  static const bool avx2 = __builtin_cpu_supports("avx2") && getenv("PA_PACK_SCALAR") == nullptr;
  uint32_t nt = n_threads ? n_threads : std::min<uint32_t>(pa_cpu_budget(), 64u);
  nt = std::max<uint32_t>(1u, std::min<uint32_t>(nt, (uint32_t)(((uint64_t)nq * ns) / 8192u + 1u)));
  nt = std::min(nt, std::max(1u, nq));
  // rows are dealt in small blocks through a shared counter: NULL-heavy rows cost nothing, dense ones a pow each
  constexpr uint32_t kRowBlock = 4;
  std::atomic<uint32_t> next1{0}, next2{0};
  // symmetric saves one pow per ordered pair and pays for it with a second pass over the matrix.  An all-against-all matrix of
  // many species is NULL almost everywhere (25 000 non-NULL pairs in 10^6 at 40 species): there the second scan costs more
  // than the pows it saves, and the two forms give the same doubles (same integers, same division, same pow) -- so a matrix
  // that is sparse in a sample of its rows is done in one pass.
  if (symmetric) {
    uint64_t seen = 0, non_null = 0;
    const uint32_t stride = std::max(1u, nq / 64u);
    for (uint32_t q = 0; q < nq; q += stride)
      for (uint32_t s = 0; s < ns; ++s) { ++seen; non_null += h_counts[(uint64_t)q * ns + s] != 0u; }
    if (non_null * 8u < seen) symmetric = 0;
  }
  auto pass1 = [&](uint32_t, uint32_t) {
    for (;;) {
      const uint32_t r0 = next1.fetch_add(kRowBlock), r1 = std::min(nq, r0 + kRowBlock);
      if (r0 >= nq) break;
      for (uint32_t q = r0; q < r1; ++q) {
        const double qs = (double)h_q_sizes[q];
        for (uint32_t s = 0; s < ns; ++s) {
          const uint64_t idx = (uint64_t)q * ns + s;
          // runs of empty intersections (nearly every pair of an all-against-all matrix of many species): eight NULL pairs at a time
          if (avx2 && s + 8 <= ns && null_run8(h_counts + idx, h_identity + idx, h_cov_query + idx, h_is_null ? h_is_null + idx : nullptr)) {
            s += 7;
            continue;
          }
          const uint32_t c = h_counts[idx];
          if (c == 0) {
            h_identity[idx] = NAN;
            h_cov_query[idx] = NAN;
            if (h_is_null) h_is_null[idx] = 1;
            continue;
          }
          const double qa = std::pow((double)c / qs, inv_k);
          h_cov_query[idx] = qa;
          if (!symmetric) {
            const double ma = std::pow((double)c / (double)h_s_sizes[s], inv_k);
            h_identity[idx] = qa > ma ? qa : ma;
          }
          if (h_is_null) h_is_null[idx] = 0;
        }
      }
    }
  };
  auto pass2 = [&](uint32_t, uint32_t) {  // symmetric only: identity = max(cov[q][s], cov[s][q]), blocked for the transposed reads
    constexpr uint32_t kB = 64;
    for (;;) {
      const uint32_t qb = next2.fetch_add(kB), qe = std::min(nq, qb + kB);
      if (qb >= nq) break;
      for (uint32_t sb = 0; sb < ns; sb += kB)
        for (uint32_t q = qb; q < qe; ++q)
          for (uint32_t s = sb; s < std::min(sb + kB, ns); ++s) {
            const uint64_t idx = (uint64_t)q * ns + s;
            if (h_counts[idx] == 0) continue;
            const double qa = h_cov_query[idx], ma = h_cov_query[(uint64_t)s * ns + q];
            h_identity[idx] = qa > ma ? qa : ma;
          }
    }
  };
  return pa_host_guard("pa_ani_host", pa_set_error, [&] {
    HostPool &pool = HostPool::get();
    pool.run(nt, pass1);
    if (symmetric) pool.run(std::min<uint32_t>(nt, (nq + 63u) / 64u), pass2);
    return (int)PA_OK;
  });
}
