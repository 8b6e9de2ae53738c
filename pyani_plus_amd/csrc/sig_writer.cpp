// sig_writer.cpp -- bulk writer of sourmash-format signature files.
//
// The reference gets one `<md5>.sig` per genome from `sourmash scripts singlesketch`
// (pyani_plus/methods/sourmash.py:67-83) and its tests compare every JSON key of those files
// (tests/snakemake/test_sourmash_workflow.py:43-67).  The host side formats everything around the hash
// list with json.dumps (so string escaping is Python's); what is done here, on a pool of threads, is the
// part that costs time at N = 1000: ~5 000 decimal numbers per file and the md5 over them.
#include <unistd.h>

#include <charconv>
#include <cstdio>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pyani_hip.h"
#include "host_pool.h"
#include "md5.h"

void pa_set_error(const char *fmt, ...);

namespace {

int write_one(const char *path, const char *head, const char *mid, const char *tail, uint32_t ksize,
              const uint64_t *mins, uint64_t n) {
  std::string body;
  body.reserve(n * 21 + 64);
  Md5 md5;
  char num[24];
  {
    auto r = std::to_chars(num, num + sizeof num, ksize);
    md5.update(reinterpret_cast<const uint8_t *>(num), (size_t)(r.ptr - num));
  }
  for (uint64_t i = 0; i < n; ++i) {
    auto r = std::to_chars(num, num + sizeof num, mins[i]);
    md5.update(reinterpret_cast<const uint8_t *>(num), (size_t)(r.ptr - num));
    if (i) body.push_back(',');
    body.append(num, (size_t)(r.ptr - num));
  }
  char hex[33];
  md5.hex(hex);
  // the temporary name carries the process id: two workers of a multi-GPU run that hold the same genome (a duplicated
  // input file, reported by the parent afterwards) must not write through one temporary file
  const std::string tmp = std::string(path) + "." + std::to_string((long long)getpid()) + ".tmp";
  FILE *f = fopen(tmp.c_str(), "wb");
  if (!f) return PA_E_INVALID;
  bool ok = fputs(head, f) >= 0 && fwrite(body.data(), 1, body.size(), f) == body.size() && fputs(mid, f) >= 0 &&
            fputs(hex, f) >= 0 && fputs(tail, f) >= 0;
  ok = (fclose(f) == 0) && ok;
  if (!ok || rename(tmp.c_str(), path) != 0) { remove(tmp.c_str()); return PA_E_INVALID; }
  return PA_OK;
}

}  // namespace

extern "C" int pa_write_sigs(uint32_t n_files, const char *const *paths, const char *const *heads, const char *const *mids,
                             const char *const *tails, uint32_t ksize, const uint64_t *h_mins, const uint64_t *h_off,
                             uint32_t n_threads) {
  if (n_files == 0) return PA_OK;
  if (!paths || !heads || !mids || !tails || !h_off || (!h_mins && h_off[n_files] > 0)) {
    pa_set_error("pa_write_sigs: null argument");
    return PA_E_INVALID;
  }
  if (n_threads == 0) n_threads = pa_cpu_budget();
  if (n_threads == 0) n_threads = 1;
  if (n_threads > n_files) n_threads = n_files;
  return pa_host_guard("pa_write_sigs", pa_set_error, [&] {
    std::vector<int> status(n_threads, PA_OK);
    std::vector<uint32_t> failed(n_threads, 0);
    HostPool::get().run(n_threads, [&](uint32_t t, uint32_t) {  // a throwing worker (out of memory) surfaces in run()
      for (uint32_t i = t; i < n_files; i += n_threads) {
        const int st = write_one(paths[i], heads[i], mids[i], tails[i], ksize, h_mins + h_off[i], h_off[i + 1] - h_off[i]);
        if (st != PA_OK && status[t] == PA_OK) { status[t] = st; failed[t] = i; }
      }
    });
    for (uint32_t t = 0; t < n_threads; ++t)
      if (status[t] != PA_OK) {
        pa_set_error("pa_write_sigs: could not write %s", paths[failed[t]]);
        return status[t];
      }
    return (int)PA_OK;
  });
}
