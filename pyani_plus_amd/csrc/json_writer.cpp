// json_writer.cpp -- bulk writer of the reference's JSON column file.
//
// "Next" row 1 of SURVEY.md section 8(f).  The reference builds one Python dict per
// comparison and re-serialises the whole list after every 100 000 rows
// (pyani_plus/private_cli.py:1863-1888).  For N^2 = 10^6 rows that is tens of seconds of
// interpreter time against milliseconds of GPU time, so the rows are formatted here,
// byte-identical to json.dumps of the same dicts (pyani_plus/private_cli.py:454-504):
//   {"query_hash": "Q", "subject_hash": "S", "identity": 0.99, "cov_query": 0.98}
// with ", " between rows, `null` for NULL, and floats in Python's repr() form (shortest
// round-trip digits; fixed notation for 1e-4 <= |x| < 1e16, otherwise d.ddde-XX).
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <algorithm>

#include "../../include/pyani_hip.h"
#include "host_pool.h"

void pa_set_error(const char *fmt, ...);

namespace {

// Python float.__repr__ for finite doubles
inline char *put_double(char *p, char *end, double v) {
  if (v == 0.0) {
    if (std::signbit(v)) *p++ = '-';
    memcpy(p, "0.0", 3);
    return p + 3;
  }
  const double a = std::fabs(v);
  // shortest round-trip digits in scientific form (Ryu inside libstdc++), then place the point
  char sci[40];
  auto r = std::to_chars(sci, sci + sizeof(sci), v, std::chars_format::scientific);
  if (!(a >= 1e-4 && a < 1e16)) {  // Python repr keeps d[.ddd]e[+-]XX outside this range
    const size_t n = (size_t)(r.ptr - sci);
    memcpy(p, sci, n);
    return p + n;
  }
  const char *q = sci;
  if (*q == '-') *p++ = *q++;
  char digits[24];
  int nd = 0;
  digits[nd++] = *q++;
  if (*q == '.') {
    ++q;
    while (*q != 'e') digits[nd++] = *q++;
  }
  ++q;  // 'e'
  const bool neg = (*q == '-');
  ++q;  // sign
  int ex = 0;
  while (q < r.ptr) ex = ex * 10 + (*q++ - '0');
  if (neg) ex = -ex;
  if (ex >= 0) {  // ddd[.ddd] with ex+1 integer digits
    for (int i = 0; i <= ex; ++i) *p++ = i < nd ? digits[i] : '0';
    *p++ = '.';
    if (nd > ex + 1) for (int i = ex + 1; i < nd; ++i) *p++ = digits[i];
    else *p++ = '0';
  } else {  // 0.000ddd
    *p++ = '0';
    *p++ = '.';
    for (int i = 0; i < -ex - 1; ++i) *p++ = '0';
    for (int i = 0; i < nd; ++i) *p++ = digits[i];
  }
  (void)end;
  return p;
}

// rows [q0, q1) x all subjects of one block, formatted into `out`; `first` = no ", " before the first row
void format_rows(std::vector<char> &out, bool first, const char *const *q_hashes, const size_t *qlen, uint32_t q0,
                 uint32_t q1, const char *const *s_hashes, const size_t *slen, uint32_t ns, const double *identity,
                 const double *cov_query, const uint8_t *is_null, const int64_t *aln_length, const int64_t *sim_errors) {
  size_t fill = 0;
  for (uint32_t q = q0; q < q1; ++q) {
    for (uint32_t s = 0; s < ns; ++s) {
      if (fill + qlen[q] + slen[s] + 256 > out.size()) out.resize(out.size() * 2 + qlen[q] + slen[s] + 256);
      char *p = out.data() + fill;
      char *const end = out.data() + out.size();
      if (!first) { memcpy(p, ", ", 2); p += 2; }
      first = false;
      memcpy(p, "{\"query_hash\": \"", 16); p += 16;
      memcpy(p, q_hashes[q], qlen[q]); p += qlen[q];
      memcpy(p, "\", \"subject_hash\": \"", 20); p += 20;
      memcpy(p, s_hashes[s], slen[s]); p += slen[s];
      memcpy(p, "\", \"identity\": ", 15); p += 15;
      const uint64_t idx = (uint64_t)q * ns + s;
      if (is_null[idx]) { memcpy(p, "null", 4); p += 4; } else p = put_double(p, end, identity[idx]);
      if (aln_length) {  // the fastANI worker's proxy columns, in the key order of pyani_plus/private_cli.py:1066-1080
        memcpy(p, ", \"aln_length\": ", 16); p += 16;
        if (is_null[idx]) { memcpy(p, "null", 4); p += 4; } else p = std::to_chars(p, end, aln_length[idx]).ptr;
        memcpy(p, ", \"sim_errors\": ", 16); p += 16;
        if (is_null[idx]) { memcpy(p, "null", 4); p += 4; } else p = std::to_chars(p, end, sim_errors[idx]).ptr;
      }
      memcpy(p, ", \"cov_query\": ", 15); p += 15;
      if (is_null[idx]) { memcpy(p, "null", 4); p += 4; } else p = put_double(p, end, cov_query[idx]);
      *p++ = '}';
      fill = (size_t)(p - out.data());
    }
  }
  out.resize(fill);
}

// all rows of a block through `f`: host threads format runs of query rows, the caller writes them in order
bool write_block(FILE *f, bool first, const char *const *q_hashes, uint32_t nq, const char *const *s_hashes, uint32_t ns,
                 const double *identity, const double *cov_query, const uint8_t *is_null, const int64_t *aln_length = nullptr,
                 const int64_t *sim_errors = nullptr) {
  if (nq == 0 || ns == 0) return true;
  std::vector<size_t> qlen(nq), slen(ns);
  for (uint32_t q = 0; q < nq; ++q) qlen[q] = strlen(q_hashes[q]);
  for (uint32_t s = 0; s < ns; ++s) slen[s] = strlen(s_hashes[s]);
  const uint32_t rows_per_chunk = std::max<uint32_t>(1u, (uint32_t)(16384u / ns));  // ~16k comparisons = ~2.5 MB of text
  const uint32_t n_chunks = (nq + rows_per_chunk - 1) / rows_per_chunk;
  const uint32_t nt = pa_host_threads(n_chunks, 1, 0);
  std::vector<std::vector<char>> bufs(nt);
  bool ok = true;
  for (uint32_t c0 = 0; c0 < n_chunks && ok; c0 += nt) {
    const uint32_t in_round = std::min(nt, n_chunks - c0);
    HostPool::get().run(in_round, [&](uint32_t w, uint32_t) {
      const uint32_t c = c0 + w, q0 = c * rows_per_chunk, q1 = std::min(nq, q0 + rows_per_chunk);
      bufs[w].resize(std::max<size_t>(bufs[w].capacity(), 1 << 20));
      format_rows(bufs[w], first && c == 0, q_hashes, qlen.data(), q0, q1, s_hashes, slen.data(), ns, identity, cov_query,
                  is_null, aln_length, sim_errors);
    });
    for (uint32_t w = 0; w < in_round && ok; ++w) ok = fwrite(bufs[w].data(), 1, bufs[w].size(), f) == bufs[w].size();
  }
  return ok;
}

}  // namespace

static int write_comparisons_json(const char *path, const char *prefix, const char *suffix,
                                  const char *const *q_hashes, uint32_t nq, const char *const *s_hashes,
                                  uint32_t ns, const double *identity, const double *cov_query,
                                  const uint8_t *is_null) {
  if (!path || !prefix || !suffix || (nq && !q_hashes) || (ns && !s_hashes) || ((uint64_t)nq * ns && (!identity || !cov_query || !is_null))) {
    pa_set_error("pa_write_comparisons_json: null argument");
    return PA_E_INVALID;
  }
  FILE *f = fopen(path, "wb");
  if (!f) { pa_set_error("cannot open %s for writing", path); return PA_E_INVALID; }
  bool ok = fwrite(prefix, 1, strlen(prefix), f) == strlen(prefix);
  try {
    ok = ok && write_block(f, true, q_hashes, nq, s_hashes, ns, identity, cov_query, is_null);
  } catch (...) { fclose(f); throw; }
  ok = ok && fwrite(suffix, 1, strlen(suffix), f) == strlen(suffix);
  ok = (fclose(f) == 0) && ok;
  if (!ok) { pa_set_error("short write to %s", path); return PA_E_INVALID; }
  return PA_OK;
}

// Progressive form (the reference re-serialises its whole list every 100 000 rows so that an interrupted
// worker leaves the completed comparisons behind, pyani_plus/private_cli.py:1863-1894): the file written by
// pa_write_comparisons_json ends with `suffix`; this call moves the suffix back by one block of rows, so the
// file is a complete JSON document after every call and no row is ever formatted twice.
static int append_comparisons_json(const char *path, const char *suffix, int file_has_rows,
                                   const char *const *q_hashes, uint32_t nq, const char *const *s_hashes,
                                   uint32_t ns, const double *identity, const double *cov_query,
                                   const uint8_t *is_null, const int64_t *aln_length = nullptr,
                                   const int64_t *sim_errors = nullptr) {
  if (!path || !suffix || (nq && !q_hashes) || (ns && !s_hashes) || ((uint64_t)nq * ns && (!identity || !cov_query || !is_null))) {
    pa_set_error("pa_append_comparisons_json: null argument");
    return PA_E_INVALID;
  }
  FILE *f = fopen(path, "r+b");
  if (!f) { pa_set_error("cannot open %s for appending", path); return PA_E_INVALID; }
  const size_t ls = strlen(suffix);
  std::vector<char> tail(ls + 1, 0);
  bool ok = fseeko(f, -(off_t)ls, SEEK_END) == 0 && fread(tail.data(), 1, ls, f) == ls && memcmp(tail.data(), suffix, ls) == 0;
  if (!ok) { fclose(f); pa_set_error("%s does not end with the expected JSON suffix", path); return PA_E_INVALID; }
  ok = fseeko(f, -(off_t)ls, SEEK_END) == 0;
  try {
    ok = ok && write_block(f, !file_has_rows, q_hashes, nq, s_hashes, ns, identity, cov_query, is_null, aln_length, sim_errors);
  } catch (...) { fclose(f); throw; }
  ok = ok && fwrite(suffix, 1, ls, f) == ls;
  ok = (fclose(f) == 0) && ok;
  if (!ok) { pa_set_error("short write to %s", path); return PA_E_INVALID; }
  return PA_OK;
}

// C++ exceptions (std::bad_alloc from a formatting buffer, on the caller's or a pool thread) never cross the C ABI
extern "C" int pa_write_comparisons_json(const char *path, const char *prefix, const char *suffix,
                                         const char *const *q_hashes, uint32_t nq, const char *const *s_hashes,
                                         uint32_t ns, const double *identity, const double *cov_query,
                                         const uint8_t *is_null) {
  return pa_host_guard("pa_write_comparisons_json", pa_set_error, [&] {
    return write_comparisons_json(path, prefix, suffix, q_hashes, nq, s_hashes, ns, identity, cov_query, is_null);
  });
}

extern "C" int pa_append_comparisons_json(const char *path, const char *suffix, int file_has_rows,
                                          const char *const *q_hashes, uint32_t nq, const char *const *s_hashes,
                                          uint32_t ns, const double *identity, const double *cov_query,
                                          const uint8_t *is_null) {
  return pa_host_guard("pa_append_comparisons_json", pa_set_error, [&] {
    return append_comparisons_json(path, suffix, file_has_rows, q_hashes, nq, s_hashes, ns, identity, cov_query, is_null);
  });
}

extern "C" int pa_append_comparisons_json_ex(const char *path, const char *suffix, int file_has_rows,
                                             const char *const *q_hashes, uint32_t nq, const char *const *s_hashes,
                                             uint32_t ns, const double *identity, const double *cov_query,
                                             const uint8_t *is_null, const int64_t *aln_length, const int64_t *sim_errors) {
  if ((aln_length == nullptr) != (sim_errors == nullptr)) {
    pa_set_error("pa_append_comparisons_json_ex: aln_length and sim_errors go together");
    return PA_E_INVALID;
  }
  return pa_host_guard("pa_append_comparisons_json_ex", pa_set_error, [&] {
    return append_comparisons_json(path, suffix, file_has_rows, q_hashes, nq, s_hashes, ns, identity, cov_query, is_null, aln_length,
                                   sim_errors);
  });
}

// fastANI prints its identity through a C++ stream with the default precision (six significant digits) and the
// reference parses that text (pyani_plus/methods/fastani.py:98-120): value -> "%.6g" -> value, in place; NaN stays.
extern "C" int pa_round_sig6(double *h_values, uint64_t n) {
  if (!h_values && n) { pa_set_error("pa_round_sig6: null argument"); return PA_E_INVALID; }
  char buf[40];
  for (uint64_t i = 0; i < n; ++i) {
    if (std::isnan(h_values[i])) continue;
    snprintf(buf, sizeof buf, "%.6g", h_values[i]);
    h_values[i] = strtod(buf, nullptr);
  }
  return PA_OK;
}
