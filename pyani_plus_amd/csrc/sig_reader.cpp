// sig_reader.cpp -- bulk reader of sourmash-format signature files (the read half).
//
// In the reference's real flow `prepare-genomes` and `compute-column` are separate processes
// (pyani_plus/private_cli.py:714-754, 1803-1902): the column worker hands the N cached `<md5>.sig` files to
// `sourmash sig collect` and `manysearch` re-reads and JSON-parses every one of them
// (pyani_plus/methods/sourmash.py:160-200).  Here the N files are read on the host pool: locate the one `mins`
// list, parse its decimals, check the sketch parameters the caller asked for and the file's own `md5sum`
// (md5 of str(ksize) + the concatenated decimals, sourmash's checksum of a sketch).  Only the layout that
// `sourmash scripts singlesketch` and this backend write -- one DNA sketch per file -- is taken here; anything
// else (several sketches, other molecules) is reported as "not handled" and the caller parses that file with the
// general JSON reader.
#include <algorithm>
#include <atomic>
#include <charconv>
#include <cstdio>
#include <cstring>
#include <string>
#include <sys/stat.h>
#include <vector>

#include "../../include/pyani_hip.h"
#include "host_pool.h"
#include "md5.h"

void pa_set_error(const char *fmt, ...);

struct pa_sig_batch {
  struct File {
    int status = PA_OK;  // PA_OK, PA_SIG_UNHANDLED (> 0) or a negative pa_status
    std::string message;
    std::vector<uint64_t> mins;
  };
  std::vector<File> files;
};

namespace {

inline bool is_ws(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r'; }

// position just past `"key"` <ws> ':' <ws>, for a key that occurs exactly once as a JSON string token; npos otherwise.
// Inside a JSON string value a quote is written \" , so the byte sequence "key" followed by a colon can only be a key.
size_t value_of_key(const std::string &text, const char *key, int *occurrences) {
  const std::string token = std::string("\"") + key + "\"";
  size_t found = std::string::npos, at = 0;
  int n = 0;
  while ((at = text.find(token, at)) != std::string::npos) {
    size_t p = at + token.size();
    while (p < text.size() && is_ws(text[p])) ++p;
    if (p < text.size() && text[p] == ':' && (at == 0 || text[at - 1] != '\\')) {
      ++p;
      while (p < text.size() && is_ws(text[p])) ++p;
      found = p;
      ++n;
    }
    at += token.size();
  }
  *occurrences = n;
  return n == 1 ? found : std::string::npos;
}

bool parse_u64_at(const std::string &text, size_t p, uint64_t *out, size_t *end) {
  const char *b = text.data() + p, *e = text.data() + text.size();
  auto r = std::from_chars(b, e, *out);
  if (r.ec != std::errc() || r.ptr == b) return false;
  *end = (size_t)(r.ptr - text.data());
  return true;
}

void read_one(const char *path, uint32_t ksize, uint64_t max_hash, pa_sig_batch::File &out) {
  auto fail = [&](int status, const std::string &why) { out.status = status; out.message = why; out.mins.clear(); };
  FILE *f = fopen(path, "rb");
  if (!f) { fail(PA_E_IO, std::string("cannot open ") + path); return; }
  std::string text;
  {
    struct stat sb;
    if (fstat(fileno(f), &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size > 0) text.reserve((size_t)sb.st_size + 1);
    char buf[1 << 16];
    size_t got;
    while ((got = fread(buf, 1, sizeof buf, f)) > 0) text.append(buf, got);
    const bool bad = ferror(f) != 0;
    fclose(f);
    if (bad) { fail(PA_E_IO, std::string("read error on ") + path); return; }
  }
  int n_mins = 0, n_sigs = 0, n_k = 0, n_mh = 0, n_num = 0, n_mol = 0, n_md5 = 0;
  const size_t p_mins = value_of_key(text, "mins", &n_mins);
  (void)value_of_key(text, "signatures", &n_sigs);
  const size_t p_k = value_of_key(text, "ksize", &n_k), p_mh = value_of_key(text, "max_hash", &n_mh);
  const size_t p_num = value_of_key(text, "num", &n_num), p_mol = value_of_key(text, "molecule", &n_mol);
  const size_t p_md5 = value_of_key(text, "md5sum", &n_md5);
  if (n_mins == 0 || n_sigs == 0) { fail(PA_E_INVALID, "not a sourmash signature file (no sketch in it)"); return; }
  if (n_mins != 1 || n_sigs != 1 || n_k != 1 || n_mh != 1 || n_num > 1 || n_mol > 1 || n_md5 > 1 ||
      p_mins == std::string::npos || text[p_mins] != '[') {
    fail(PA_SIG_UNHANDLED, "not the one-sketch layout");
    return;
  }
  uint64_t v = 0;
  size_t end = 0;
  if (!parse_u64_at(text, p_k, &v, &end)) { fail(PA_E_INVALID, "ksize is not a number"); return; }
  if (v != ksize) { fail(PA_SIG_UNHANDLED, "sketch has another ksize"); return; }
  if (!parse_u64_at(text, p_mh, &v, &end)) { fail(PA_E_INVALID, "max_hash is not a number"); return; }
  if (v != max_hash) { fail(PA_SIG_UNHANDLED, "sketch has another max_hash"); return; }
  if (n_num == 1 && (!parse_u64_at(text, p_num, &v, &end) || v != 0)) { fail(PA_SIG_UNHANDLED, "not a scaled (num=0) sketch"); return; }
  if (n_mol == 1 && text.compare(p_mol, 5, "\"DNA\"") != 0) { fail(PA_SIG_UNHANDLED, "not a DNA sketch"); return; }
  // the hash list
  std::vector<uint64_t> &mins = out.mins;
  mins.clear();
  mins.reserve((text.size() - p_mins) / 16 + 16);  // a hash below 2^64 / scaled prints as 17 to 20 characters with its comma
  size_t p = p_mins + 1;
  bool sorted = true;
  std::string digits;  // str(ksize) + the decimals as listed, for the checksum
  digits.reserve(text.size() - p + 8);
  digits += std::to_string(ksize);
  for (;;) {
    while (p < text.size() && is_ws(text[p])) ++p;
    if (p >= text.size()) { fail(PA_E_INVALID, "hash list is not closed"); return; }
    if (text[p] == ']') break;
    if (!mins.empty()) {
      if (text[p] != ',') { fail(PA_E_INVALID, "unexpected character in the hash list"); return; }
      ++p;
      while (p < text.size() && is_ws(text[p])) ++p;
    }
    uint64_t h = 0;
    if (!parse_u64_at(text, p, &h, &end)) { fail(PA_E_INVALID, "hash list holds something that is not an unsigned 64-bit integer"); return; }
    if (!mins.empty() && h <= mins.back()) sorted = false;
    mins.push_back(h);
    if (text[p] == '0' && end - p > 1) { fail(PA_E_INVALID, "hash list holds a number with leading zeros"); return; }
    digits.append(text, p, end - p);  // canonical decimals: from_chars took digits only, and no leading zero
    p = end;
  }
  // the file's own checksum of the sketch, over the hashes as they are listed
  if (n_md5 == 1) {
    if (p_md5 + 34 > text.size() || text[p_md5] != '"' || text[p_md5 + 33] != '"') { fail(PA_E_INVALID, "md5sum is not a 32-character string"); return; }
    Md5 md5;
    md5.update(reinterpret_cast<const uint8_t *>(digits.data()), digits.size());
    char hex[33];
    md5.hex(hex);
    if (memcmp(hex, text.data() + p_md5 + 1, 32) != 0) {
      fail(PA_E_INVALID, std::string("md5sum ") + text.substr(p_md5 + 1, 32) + " does not match the hashes listed (" + hex + ")");
      return;
    }
  }
  if (!sorted) {  // sourmash writes them ascending; a hand-made file need not
    std::sort(mins.begin(), mins.end());
    mins.erase(std::unique(mins.begin(), mins.end()), mins.end());
  }
  out.status = PA_OK;
}

int read_sigs(const char *const *paths, uint32_t n, uint32_t ksize, uint64_t max_hash, uint32_t n_threads, pa_sig_batch *b) {
  b->files.resize(n);
  uint32_t nt = n_threads ? n_threads : pa_cpu_budget();
  nt = std::max<uint32_t>(1u, std::min<uint32_t>(nt, n));
  std::atomic<uint32_t> next{0};
  HostPool::get().run(nt, [&](uint32_t, uint32_t) {
    for (;;) {
      const uint32_t i = next.fetch_add(1);
      if (i >= n) break;
      if (!paths[i]) { b->files[i].status = PA_E_INVALID; b->files[i].message = "null path"; continue; }
      read_one(paths[i], ksize, max_hash, b->files[i]);
    }
  });
  return PA_OK;
}

}  // namespace

extern "C" {

int pa_read_sigs(const char *const *paths, uint32_t n, uint32_t ksize, uint64_t max_hash, uint32_t n_threads,
                 pa_sig_batch **out) {
  if (!out || (n && !paths)) { pa_set_error("pa_read_sigs: null argument"); return PA_E_INVALID; }
  *out = nullptr;
  pa_sig_batch *b = new (std::nothrow) pa_sig_batch();
  if (!b) { pa_set_error("pa_read_sigs: out of host memory"); return PA_E_NOMEM; }
  const int st = pa_host_guard("pa_read_sigs", pa_set_error, [&] { return read_sigs(paths, n, ksize, max_hash, n_threads, b); });
  if (st != PA_OK) { delete b; return st; }
  *out = b;
  return PA_OK;
}

int pa_sig_batch_info(const pa_sig_batch *b, uint32_t i, uint64_t *n_mins, const char **message) {
  if (!b || i >= b->files.size()) { pa_set_error("pa_sig_batch_info: index out of range"); return PA_E_INVALID; }
  if (n_mins) *n_mins = b->files[i].status == PA_OK ? b->files[i].mins.size() : 0;
  if (message) *message = b->files[i].message.c_str();
  return b->files[i].status;
}

int pa_sig_batch_copy(const pa_sig_batch *b, uint64_t *h_mins, uint64_t *h_off) {
  if (!b || !h_off) { pa_set_error("pa_sig_batch_copy: null argument"); return PA_E_INVALID; }
  uint64_t pos = 0;
  for (size_t i = 0; i < b->files.size(); ++i) {
    h_off[i] = pos;
    if (b->files[i].status == PA_OK) pos += b->files[i].mins.size();
  }
  h_off[b->files.size()] = pos;
  if (pos && !h_mins) { pa_set_error("pa_sig_batch_copy: null hash buffer"); return PA_E_INVALID; }
  for (size_t i = 0; i < b->files.size(); ++i)
    if (b->files[i].status == PA_OK && !b->files[i].mins.empty())
      memcpy(h_mins + h_off[i], b->files[i].mins.data(), b->files[i].mins.size() * sizeof(uint64_t));
  return PA_OK;
}

void pa_sig_batch_free(pa_sig_batch *b) { delete b; }

}  // extern "C"
