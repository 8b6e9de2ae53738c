// pack_host.cpp -- host side of the boundary: FASTA text -> 2-bit arena.
//
// Replaces the FASTA reader inside `sourmash scripts singlesketch`
// (pyani_plus/methods/sourmash.py:67-83).  Record and whitespace semantics are
// those of pyani_plus/utils.py:67-90 (fasta_bytes_iterator): text before the
// first '>' line is ignored, a '>' only starts a record at the start of a line,
// " \t\r\n" are dropped from sequence lines.  Residues are case-insensitive;
// anything outside ACGT becomes an invalid position, and one invalid position
// is written between records so that no k-mer window spans two records.
#include <cmath>
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

// Runs of set bits of an invalid-position mask, ascending (word scan: zero words cost one compare).
extern "C" int64_t pa_mask_runs(const uint32_t *h_mask, uint64_t arena_bases, uint64_t *h_run_start,
                                uint64_t *h_run_len, uint64_t cap) {
  if (!h_mask && arena_bases) return -1;
  const uint64_t n_words = arena_bases / 32;
  uint64_t n = 0, start = 0;
  bool in_run = false;
  for (uint64_t w = 0; w < n_words; ++w) {
    uint32_t x = h_mask[w];
    if (!in_run && x == 0) continue;
    if (in_run && x == 0xffffffffu) continue;
    for (uint32_t b = 0; b < 32; ++b) {
      const bool bit = (x >> b) & 1u;
      if (bit && !in_run) { in_run = true; start = w * 32 + b; }
      else if (!bit && in_run) {
        in_run = false;
        if (n < cap && h_run_start && h_run_len) { h_run_start[n] = start; h_run_len[n] = w * 32 + b - start; }
        ++n;
      }
    }
  }
  if (in_run) {
    if (n < cap && h_run_start && h_run_len) { h_run_start[n] = start; h_run_len[n] = n_words * 32 - start; }
    ++n;
  }
  return (int64_t)n;
}

extern "C" uint64_t pa_max_hash(uint64_t scaled) {
  if (scaled == 0) return 0;
  if (scaled == 1) return UINT64_MAX;
  return (uint64_t)(18446744073709551616.0 / (double)scaled);
}

extern "C" int pa_pack_fasta(const uint8_t *h_text, uint64_t n_text, uint32_t *h_packed, uint32_t *h_mask,
                             uint64_t cap_bases, uint64_t *n_bases, uint64_t *n_residues, uint64_t *n_records,
                             uint64_t *n_invalid) {
  if ((!h_text && n_text) || !h_packed || !h_mask || (cap_bases & 63)) {
    pa_set_error("pa_pack_fasta: null buffer or capacity %llu not a multiple of 64", (unsigned long long)cap_bases);
    return PA_E_INVALID;
  }
  ArenaWriter w{h_packed, h_mask, cap_bases};
  uint64_t residues = 0, records = 0, invalid = 0;
  uint64_t i = 0;
  bool in_record = false;
  while (i < n_text) {
    if (h_text[i] == '>') {  // title line (we are at the start of a line)
      if (in_record) w.put(0, 1);  // separator: windows never span records
      in_record = true;
      ++records;
      while (i < n_text && h_text[i] != '\n') ++i;
      if (i < n_text) ++i;
      continue;
    }
    // a sequence line (or junk before the first record): consume to end of line
    if (!in_record) {
      while (i < n_text && h_text[i] != '\n') ++i;
      if (i < n_text) ++i;
      continue;
    }
    while (i < n_text) {
      const uint8_t ch = h_text[i++];
      const uint8_t code = kLut.v[ch];
      if (code < 4) { w.put(code, 0); ++residues; }
      else if (code == 4) { w.put(0, 1); ++residues; ++invalid; }
      else if (ch == '\n') break;
    }
  }
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
  uint32_t nt = n_threads ? n_threads : std::min<uint32_t>(pa_cpu_budget(), 64u);
  nt = std::max<uint32_t>(1u, std::min<uint32_t>(nt, (uint32_t)(((uint64_t)nq * ns) / 8192u + 1u)));
  nt = std::min(nt, std::max(1u, nq));
  // rows are dealt in small blocks through a shared counter: NULL-heavy rows cost nothing, dense ones a pow each
  constexpr uint32_t kRowBlock = 4;
  std::atomic<uint32_t> next1{0}, next2{0};
  auto pass1 = [&](uint32_t, uint32_t) {
    for (;;) {
      const uint32_t r0 = next1.fetch_add(kRowBlock), r1 = std::min(nq, r0 + kRowBlock);
      if (r0 >= nq) break;
      for (uint32_t q = r0; q < r1; ++q) {
        const double qs = (double)h_q_sizes[q];
        for (uint32_t s = 0; s < ns; ++s) {
          const uint64_t idx = (uint64_t)q * ns + s;
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
  HostPool &pool = HostPool::get();
  pool.run(nt, pass1);
  if (symmetric) pool.run(std::min<uint32_t>(nt, (nq + 63u) / 64u), pass2);
  return PA_OK;
}
