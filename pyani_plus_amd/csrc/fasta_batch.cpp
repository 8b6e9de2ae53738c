// fasta_batch.cpp -- multi-threaded host front-end: files -> (md5, length, title, 2-bit arena).
//
// "Next" row 2 of SURVEY.md section 8(f).  The reference reads every genome twice, serially,
// in Python: once for the md5 of the decompressed bytes (pyani_plus/utils.py:142-196), once
// for length/description (pyani_plus/db_orm.py:832-866), and a third time inside
// `sourmash scripts singlesketch` (pyani_plus/methods/sourmash.py:67-83).  Here one pass per
// file on a pool of host threads does all of it: read, gunzip (zlib), md5, parse, pack.
//
// Error conventions follow the reference: a `.gz` suffix that disagrees with the content
// (db_orm.py:846-854), a file without any FASTA record (db_orm.py:839-843) and unreadable
// files are reported per file through `status`/`message`, never by aborting the batch.
#include <sys/mman.h>
#include <sys/stat.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pyani_hip.h"
#include "host_pool.h"
#include "inflate_fast.h"
#include "md5.h"
#include "md5_mb.h"

void pa_set_error(const char *fmt, ...);
int pa_pack_fasta_records(const uint8_t *h_text, uint64_t n_text, uint32_t *h_packed, uint32_t *h_mask, uint64_t cap_bases,
                          uint64_t *n_bases, uint64_t *n_residues, uint64_t *n_records, uint64_t *n_invalid,
                          std::vector<uint64_t> *rec_start, std::vector<uint64_t> *rec_len, std::vector<uint64_t> *amb_pos,
                          std::vector<uint8_t> *amb_byte);

namespace {

struct FileResult {
  int status = PA_OK;
  std::string message, description;
  char md5[33] = {0};
  bool gz = false;
  uint64_t n_text = 0, n_bases = 0, n_residues = 0, n_records = 0, n_invalid = 0;
  // The packed bases and the mask live in the batch's slab (plain files: the room a file can need is known from
  // its size) or in vectors of the file's own (gzip: the text size is only known after inflating).
  const uint32_t *packed = nullptr, *mask = nullptr;
  std::vector<uint32_t> own_packed, own_mask;
  std::vector<uint64_t> rec_start, rec_len;  // FASTA records, positions relative to the genome start
  // residues that are neither ACGT nor N: position relative to the genome start, upper-cased byte (pa_fasta_batch_ambiguous)
  std::vector<uint64_t> amb_pos;
  std::vector<uint8_t> amb_byte;
};

// Anonymous mapping, huge pages asked for: one per batch for the packed genomes, one for their masks, one for the bytes
// of the files being read, instead of allocations per file.  A run loads several batches of the same shape one after
// the other and a fresh mapping costs a page fault and a zeroed huge page per 2 MB touched (2 GB per batch of 400
// genomes), so released mappings are kept for the next batch -- at most three, at most 3 GiB, PA_HOST_SLAB_CACHE=0
// turns it off.  (Nothing in a reused mapping is read before it is written: the packer writes every word it reports,
// the arena copy takes only those.)
struct SlabCache {
  struct Item { uint8_t *p; size_t bytes; };
  std::mutex m;
  std::vector<Item> items;
  static SlabCache &get() {
    static SlabCache *c = new SlabCache();  // never destroyed: mappings die with the process
    return *c;
  }
  static bool enabled() {
    static const bool on = [] {
      const char *v = getenv("PA_HOST_SLAB_CACHE");
      return !(v && v[0] == '0');
    }();
    return on;
  }
  uint8_t *take(size_t n, size_t *bytes) {
    std::lock_guard<std::mutex> lock(m);
    size_t best = items.size();
    for (size_t i = 0; i < items.size(); ++i)
      if (items[i].bytes >= n && items[i].bytes <= 2 * n + (64u << 20) && (best == items.size() || items[i].bytes < items[best].bytes)) best = i;
    if (best == items.size()) return nullptr;
    uint8_t *p = items[best].p;
    *bytes = items[best].bytes;
    items.erase(items.begin() + (long)best);
    return p;
  }
  void give(uint8_t *p, size_t bytes) {
    uint8_t *drop = p;
    size_t drop_bytes = bytes;
    if (enabled()) {
      std::lock_guard<std::mutex> lock(m);
      items.push_back({p, bytes});
      drop = nullptr;
      size_t total = 0;
      for (const Item &it : items) total += it.bytes;
      if (items.size() > 3 || total > (3ull << 30)) {  // let the oldest go
        drop = items.front().p;
        drop_bytes = items.front().bytes;
        items.erase(items.begin());
      }
    }
    if (drop) munmap(drop, drop_bytes);
  }
};

struct Slab {
  uint8_t *p = nullptr;
  size_t bytes = 0;
  bool alloc(size_t n) {
    release();
    if (n == 0) return true;
    if (SlabCache::enabled() && (p = SlabCache::get().take(n, &bytes))) return true;
    void *m = mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (m == MAP_FAILED) return false;
    madvise(m, n, MADV_HUGEPAGE);
    p = static_cast<uint8_t *>(m);
    bytes = n;
    return true;
  }
  void release() {
    if (p) SlabCache::get().give(p, bytes);
    p = nullptr;
    bytes = 0;
  }
  ~Slab() { release(); }
  Slab() = default;
  Slab(const Slab &) = delete;
  Slab &operator=(const Slab &) = delete;
};

// The file into `room` bytes at `buf` when it fits (the slot sized from stat()), otherwise into `raw`.
// *data / *n describe where the bytes ended up.
bool read_file(const std::string &path, uint8_t *buf, size_t room, std::vector<uint8_t> &raw, const uint8_t **data,
               size_t *n, std::string &err) {
  FILE *f = fopen(path.c_str(), "rb");
  if (!f) { err = "Input " + path + " not found"; return false; }
  if (fseeko(f, 0, SEEK_END) != 0) { fclose(f); err = "Cannot seek in " + path; return false; }
  const off_t sz = ftello(f);
  if (sz < 0 || fseeko(f, 0, SEEK_SET) != 0) { fclose(f); err = "Cannot size " + path; return false; }
  uint8_t *dst = buf;
  if ((size_t)sz > room || !buf) {  // no slot, or the file grew since it was sized
    raw.resize((size_t)sz);
    dst = raw.data();
  }
  const size_t got = sz == 0 ? 0 : fread(dst, 1, (size_t)sz, f);
  fclose(f);
  if (got != (size_t)sz) { err = "Short read on " + path; return false; }
  *data = dst;
  *n = (size_t)sz;
  return true;
}

// multi-member gzip inflate: the fast decoder of inflate_fast.h first (every member checked against its CRC-32 and
// length), zlib over the same bytes whenever that reports anything unexpected
bool gunzip_zlib(const uint8_t *raw_p, size_t raw_n, std::vector<uint8_t> &out, std::string &err);
bool gunzip(const uint8_t *raw_p, size_t raw_n, std::vector<uint8_t> &out, std::string &err) {
  static const bool zlib_only = [] {
    const char *v = getenv("PA_GUNZIP");
    return v && v[0] == 'z';
  }();
  if (!zlib_only && pa_inflate::gunzip_all(raw_p, raw_n, out)) return true;
  return gunzip_zlib(raw_p, raw_n, out, err);
}
bool gunzip_zlib(const uint8_t *raw_p, size_t raw_n, std::vector<uint8_t> &out, std::string &err) {
  z_stream zs;
  memset(&zs, 0, sizeof(zs));
  if (inflateInit2(&zs, 15 + 16) != Z_OK) { err = "inflateInit2 failed"; return false; }
  // the gzip trailer holds the size of the last member mod 2^32: a good first guess
  size_t guess = raw_n * 4 + 65536;
  if (raw_n >= 18) {
    const uint8_t *t = raw_p + raw_n - 4;
    const size_t isize = (size_t)t[0] | ((size_t)t[1] << 8) | ((size_t)t[2] << 16) | ((size_t)t[3] << 24);
    if (isize > raw_n / 2 && isize < raw_n * 1100) guess = isize + 64;
  }
  if (out.size() < guess) out.resize(guess);
  // zlib counts in 32-bit `uInt`s: both sides are fed in pieces of at most 1 GiB
  const uint8_t *in = raw_p;
  size_t in_left = raw_n;
  zs.next_in = const_cast<Bytef *>(in);
  zs.avail_in = 0;
  size_t have = 0;
  for (;;) {
    if (zs.avail_in == 0 && in_left) {
      const size_t piece = std::min<size_t>(in_left, 1u << 30);
      zs.next_in = const_cast<Bytef *>(in);
      zs.avail_in = (uInt)piece;
      in += piece;
      in_left -= piece;
    }
    if (have == out.size()) out.resize(out.size() * 2);
    zs.next_out = out.data() + have;
    zs.avail_out = (uInt)std::min<size_t>(out.size() - have, 1u << 30);
    const int rc = inflate(&zs, Z_NO_FLUSH);
    have = (size_t)(zs.next_out - out.data());
    if (rc == Z_STREAM_END) {
      if (zs.avail_in == 0 && in_left == 0) break;
      // Python's gzip module, which the reference reads with (pyani_plus/utils.py:178-196), accepts zero
      // padding after the last member: nothing but NUL bytes left means end of file
      bool only_zeros = true;
      for (uInt i = 0; i < zs.avail_in && only_zeros; ++i) only_zeros = zs.next_in[i] == 0;
      for (size_t i = 0; i < in_left && only_zeros; ++i) only_zeros = in[i] == 0;
      if (only_zeros) break;
      if (inflateReset(&zs) != Z_OK) { err = "inflateReset failed"; inflateEnd(&zs); return false; }  // next member
      continue;
    }
    if (rc != Z_OK) { err = "corrupt gzip stream"; inflateEnd(&zs); return false; }
    if (zs.avail_in == 0 && in_left == 0 && zs.avail_out != 0) { err = "truncated gzip stream"; inflateEnd(&zs); return false; }
  }
  inflateEnd(&zs);
  out.resize(have);
  return true;
}

bool ends_with(const std::string &s, const char *suf) {
  const size_t n = strlen(suf);
  return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

std::string basename_of(const std::string &p) {
  const size_t k = p.find_last_of('/');
  return k == std::string::npos ? p : p.substr(k + 1);
}

// Step 1 of a file: its bytes.  `buf`/`room`: the worker's slot for them; raw: scratch for a file that does not fit
// the slot; text: where inflated text goes (kept until the file is finished).  Returns false with status/message set.
bool acquire(const std::string &path, FileResult &r, uint8_t *buf, size_t room, std::vector<uint8_t> &raw,
             std::vector<uint8_t> &text, const uint8_t **data_out, size_t *n_out) {
  const uint8_t *data = nullptr;
  size_t n_data = 0;
  if (!read_file(path, buf, room, raw, &data, &n_data, r.message)) { r.status = PA_E_INVALID; return false; }
  r.gz = n_data >= 2 && data[0] == 0x1f && data[1] == 0x8b;
  const std::string name = basename_of(path);
  if (r.gz && !ends_with(path, ".gz")) {
    r.status = PA_E_INVALID;
    r.message = "No .gz ending, but " + name + " is gzip compressed";
    return false;
  }
  if (!r.gz && ends_with(path, ".gz")) {
    r.status = PA_E_INVALID;
    r.message = "Has .gz ending, but " + name + " is NOT gzip compressed";
    return false;
  }
  if (r.gz) {
    if (!gunzip(data, n_data, text, r.message)) { r.status = PA_E_INVALID; r.message = name + ": " + r.message; return false; }
    data = text.data();
    n_data = text.size();
  } else if (data == raw.data()) {
    text.swap(raw);  // the scratch is reused by the next file of the group: keep these bytes
    data = text.data();
  }
  r.n_text = n_data;
  *data_out = data;
  *n_out = n_data;
  return true;
}

// Step 2 (after the checksum): first title, 2-bit arena and record table in one pass over the text.
// `out_packed`/`out_mask`/`out_cap`: the file's room in the batch slab (out_cap bases, 0 = none: the file keeps
// vectors of its own).
void finish(const std::string &path, FileResult &r, const uint8_t *data, size_t n_data, uint32_t *out_packed,
            uint32_t *out_mask, uint64_t out_cap) {
  const std::string name = basename_of(path);
  // first title = description (db_orm.py:836-838): first line starting with '>'
  {
    size_t i = 0;
    while (i < n_data) {
      const void *nl = memchr(data + i, '\n', n_data - i);
      const size_t e = nl ? (size_t)(static_cast<const uint8_t *>(nl) - data) : n_data;
      if (data[i] == '>') {
        size_t b = i + 1, t = e;
        while (t > b && (data[t - 1] == ' ' || data[t - 1] == '\t' || data[t - 1] == '\r' || data[t - 1] == '\n' ||
                         data[t - 1] == '\v' || data[t - 1] == '\f'))
          --t;
        r.description.assign(reinterpret_cast<const char *>(data) + b, t - b);
        break;
      }
      i = e + 1;
    }
  }
  const uint64_t cap = pa_pack_bound(n_data);
  uint32_t *packed = out_packed, *mask = out_mask;
  if (cap > out_cap) {
    r.own_packed.resize(cap / 16);
    r.own_mask.resize(cap / 32);
    packed = r.own_packed.data();
    mask = r.own_mask.data();
  }
  const int st = pa_pack_fasta_records(data, n_data, packed, mask, cap, &r.n_bases, &r.n_residues, &r.n_records, &r.n_invalid,
                                       &r.rec_start, &r.rec_len, &r.amb_pos, &r.amb_byte);
  if (st != PA_OK) { r.status = st; r.message = name + ": packing failed"; return; }
  r.packed = packed;
  r.mask = mask;
  if (r.n_records == 0) {
    r.status = PA_E_INVALID;
    r.message = "File " + name + " is not recognised as a FASTA record";
  }
}

}  // namespace

struct pa_fasta_batch {
  std::vector<FileResult> files;
  Slab packed, mask;
};

static int fasta_batch_load(const char *const *paths, uint32_t n, int threads, pa_fasta_batch *b, pa_fasta_batch **out) {
  b->files.resize(n);
  std::vector<std::string> p(n);
  for (uint32_t i = 0; i < n; ++i) p[i] = paths[i] ? paths[i] : "";
  // Room per file, from its size: a plain file of s bytes holds at most s bases, so its packed form has a slot in
  // the batch slab; .gz files (the text size is unknown until inflated) keep vectors of their own.
  std::vector<uint64_t> slot_off(n + 1, 0), slot_cap(n, 0);
  uint64_t largest = 0;
  for (uint32_t i = 0; i < n; ++i) {
    struct stat sb;
    if (!ends_with(p[i], ".gz") && stat(p[i].c_str(), &sb) == 0 && S_ISREG(sb.st_mode)) {
      slot_cap[i] = pa_pack_bound((uint64_t)sb.st_size);
      largest = std::max<uint64_t>(largest, (uint64_t)sb.st_size);
    }
    slot_off[i + 1] = slot_off[i] + slot_cap[i];
  }
  uint32_t nt = threads > 0 ? (uint32_t)threads : pa_cpu_budget();
  if (nt > n) nt = n ? n : 1u;
  // Files are taken in groups so that their checksums can be computed side by side (md5_mb.h: sixteen messages per
  // AVX-512 register), largest files first: the lanes of a group then have similar lengths and the long files do
  // not end up alone at the end of the batch.  A worker holds the bytes of its whole group: one read slot per lane,
  // the group size reduced so that the slots stay within ~2 GiB.
  const size_t read_room = (size_t)((largest + 4095) & ~4095ull);
  uint32_t lanes = md5mb::have_avx512() ? 16u : 1u;
  while (lanes > 1 && (uint64_t)read_room * lanes * nt > (2ull << 30)) lanes /= 2;
  if ((uint64_t)n < (uint64_t)lanes * nt) lanes = std::max<uint32_t>(1u, (n + nt - 1) / nt);  // few files: spread them over the workers
  std::vector<uint32_t> order(n);
  for (uint32_t i = 0; i < n; ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return slot_cap[x] > slot_cap[y]; });
  Slab reads;
  if (!b->packed.alloc(slot_off[n] / 4) || !b->mask.alloc(slot_off[n] / 8) || !reads.alloc(read_room * lanes * nt)) {
    pa_set_error("pa_fasta_batch_load: cannot map %llu bytes of host memory",
                 (unsigned long long)(slot_off[n] / 4 + slot_off[n] / 8 + read_room * lanes * nt));
    return PA_E_NOMEM;
  }
  std::atomic<uint32_t> next{0};
  HostPool::get().run(nt, [&](uint32_t worker, uint32_t) {
    std::vector<uint8_t> raw;
    std::vector<std::vector<uint8_t>> text(lanes);
    std::vector<const uint8_t *> data(lanes);
    std::vector<size_t> n_data(lanes);
    std::vector<uint32_t> file(lanes);
    const std::unique_ptr<char[][33]> hex(new char[lanes][33]);
    for (;;) {
      const uint32_t g0 = next.fetch_add(lanes);
      if (g0 >= n) break;
      const uint32_t g1 = std::min(n, g0 + lanes);
      uint32_t have = 0;
      for (uint32_t q = g0; q < g1; ++q) {
        const uint32_t i = order[q];
        uint8_t *buf = read_room ? reads.p + ((size_t)worker * lanes + have) * read_room : nullptr;
        try {
          if (acquire(p[i], b->files[i], buf, read_room, raw, text[have], &data[have], &n_data[have])) file[have++] = i;
        } catch (const std::exception &e) {
          b->files[i].status = PA_E_NOMEM;
          b->files[i].message = std::string("exception while loading ") + p[i] + ": " + e.what();
        }
      }
      md5mb::md5_many(data.data(), n_data.data(), have, hex.get());
      for (uint32_t l = 0; l < have; ++l) {
        const uint32_t i = file[l];
        memcpy(b->files[i].md5, hex[l], 33);
        try {
          finish(p[i], b->files[i], data[l], n_data[l], reinterpret_cast<uint32_t *>(b->packed.p + slot_off[i] / 4),
                 reinterpret_cast<uint32_t *>(b->mask.p + slot_off[i] / 8), slot_cap[i]);
        } catch (const std::exception &e) {
          b->files[i].status = PA_E_NOMEM;
          b->files[i].message = std::string("exception while loading ") + p[i] + ": " + e.what();
        }
      }
    }
  });
  *out = b;
  return PA_OK;
}

extern "C" {

int pa_fasta_batch_load(const char *const *paths, uint32_t n, int threads, pa_fasta_batch **out) {
  if (!out || (n && !paths)) { pa_set_error("pa_fasta_batch_load: null argument"); return PA_E_INVALID; }
  *out = nullptr;
  pa_fasta_batch *b = new (std::nothrow) pa_fasta_batch();
  if (!b) { pa_set_error("out of host memory"); return PA_E_NOMEM; }
  // an allocation that fails on the calling thread or on a pool thread comes back as PA_E_NOMEM, never as std::terminate
  const int st = pa_host_guard("pa_fasta_batch_load", pa_set_error, [&] { return fasta_batch_load(paths, n, threads, b, out); });
  if (st != PA_OK) { *out = nullptr; delete b; }
  return st;
}

int pa_fasta_batch_info(const pa_fasta_batch *b, uint32_t i, char md5hex33[33], uint64_t *n_residues,
                        uint64_t *n_records, uint64_t *n_invalid, uint64_t *n_bases, uint64_t *n_text,
                        const char **description, const char **message, int *was_gzip) {
  if (!b || i >= b->files.size()) { pa_set_error("pa_fasta_batch_info: index out of range"); return PA_E_INVALID; }
  const FileResult &r = b->files[i];
  if (md5hex33) memcpy(md5hex33, r.md5, 33);
  if (n_residues) *n_residues = r.n_residues;
  if (n_records) *n_records = r.n_records;
  if (n_invalid) *n_invalid = r.n_invalid;
  if (n_bases) *n_bases = r.n_bases;
  if (n_text) *n_text = r.n_text;
  if (description) *description = r.description.c_str();
  if (message) *message = r.message.c_str();
  if (was_gzip) *was_gzip = r.gz ? 1 : 0;
  return r.status;
}

int pa_fasta_batch_records(const pa_fasta_batch *b, uint32_t i, const uint64_t **rec_start, const uint64_t **rec_len,
                           uint64_t *n_records) {
  if (!b || i >= b->files.size() || !rec_start || !rec_len || !n_records) {
    pa_set_error("pa_fasta_batch_records: bad argument");
    return PA_E_INVALID;
  }
  const FileResult &r = b->files[i];
  *rec_start = r.rec_start.data();
  *rec_len = r.rec_len.data();
  *n_records = r.rec_start.size();
  return r.status;
}

uint64_t pa_fasta_batch_arena_bases(const pa_fasta_batch *b) {
  uint64_t total = 0;
  if (b)
    for (const FileResult &r : b->files)
      if (r.status == PA_OK) total += r.n_bases;
  return total;
}

int pa_fasta_batch_copy_arena(const pa_fasta_batch *b, uint32_t *h_packed, uint32_t *h_mask, uint64_t *h_genome_start) {
  if (!b || !h_genome_start) { pa_set_error("pa_fasta_batch_copy_arena: null argument"); return PA_E_INVALID; }
  uint64_t pos = 0;
  for (size_t i = 0; i < b->files.size(); ++i) {
    const FileResult &r = b->files[i];
    h_genome_start[i] = pos;
    if (r.status != PA_OK) continue;  // failed files occupy no space
    pos += r.n_bases;
  }
  h_genome_start[b->files.size()] = pos;
  // the genomes land in disjoint ranges: copy them on the host pool (2 GB at N = 1000 is 0.25 s on one core)
  const size_t n = b->files.size();
  std::atomic<size_t> next{0};
  return pa_host_guard("pa_fasta_batch_copy_arena", pa_set_error, [&] {
    HostPool::get().run(pa_host_threads(pos, 32u << 20, 0), [&](uint32_t, uint32_t) {
      for (;;) {
        const size_t i = next.fetch_add(1);
        if (i >= n) break;
        const FileResult &r = b->files[i];
        if (r.status != PA_OK || !r.n_bases) continue;
        memcpy(h_packed + h_genome_start[i] / 16, r.packed, r.n_bases / 4);
        memcpy(h_mask + h_genome_start[i] / 32, r.mask, r.n_bases / 8);
      }
    });
    return (int)PA_OK;
  });
}

// The residues of the batch's genomes that are neither ACGT nor N, in arena coordinates (the genome starts of
// pa_fasta_batch_copy_arena), ascending.  Returns their number (only the first `cap` are written), negative on failure.
int64_t pa_fasta_batch_ambiguous(const pa_fasta_batch *b, uint64_t *h_pos, uint8_t *h_byte, uint64_t cap) {
  if (!b || (cap && (!h_pos || !h_byte))) { pa_set_error("pa_fasta_batch_ambiguous: null argument"); return -1; }
  uint64_t start = 0, n = 0;
  for (const FileResult &r : b->files) {
    if (r.status != PA_OK) continue;  // failed files occupy no space
    for (size_t i = 0; i < r.amb_pos.size(); ++i, ++n)
      if (n < cap) { h_pos[n] = start + r.amb_pos[i]; h_byte[n] = r.amb_byte[i]; }
    start += r.n_bases;
  }
  return (int64_t)n;
}

void pa_fasta_batch_free(pa_fasta_batch *b) { delete b; }

int pa_gunzip(const uint8_t *h_gz, uint64_t n_gz, uint8_t *h_out, uint64_t cap, uint64_t *n_out, int decoder) {
  if ((!h_gz && n_gz) || (!h_out && cap) || !n_out || decoder < 0 || decoder > 2) {
    pa_set_error("pa_gunzip: bad argument");
    return PA_E_INVALID;
  }
  *n_out = 0;
  std::vector<uint8_t> out;
  std::string err;
  bool ok;
  try {
    if (decoder == 1) {
      ok = pa_inflate::gunzip_all(h_gz, n_gz, out);
      if (!ok) err = "not accepted by the fast decoder";
    } else if (decoder == 2) {
      ok = gunzip_zlib(h_gz, n_gz, out, err);
    } else {
      ok = gunzip(h_gz, n_gz, out, err);
    }
  } catch (const std::exception &e) {
    pa_set_error("pa_gunzip: %s", e.what());
    return PA_E_NOMEM;
  }
  if (!ok) { pa_set_error("pa_gunzip: %s", err.c_str()); return PA_E_INVALID; }
  *n_out = out.size();
  if (out.size() > cap) { pa_set_error("pa_gunzip: %llu bytes needed, room for %llu", (unsigned long long)out.size(), (unsigned long long)cap); return PA_E_CAPACITY; }
  if (!out.empty()) memcpy(h_out, out.data(), out.size());
  return PA_OK;
}

}  // extern "C"
