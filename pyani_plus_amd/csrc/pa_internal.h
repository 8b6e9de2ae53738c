// pa_internal.h -- shared internals of libpyani_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/pyani_hip.h"

// ---- error plumbing -------------------------------------------------------
void pa_set_error(const char *fmt, ...);

#define PA_HIP(call)                                                                          \
  do {                                                                                        \
    hipError_t _e = (call);                                                                   \
    if (_e != hipSuccess) {                                                                   \
      pa_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
      return PA_E_HIP;                                                                        \
    }                                                                                         \
  } while (0)

#define PA_TRY(call)          \
  do {                        \
    int _s = (call);          \
    if (_s != PA_OK) return _s; \
  } while (0)

#define PA_REQUIRE(cond, ...)   \
  do {                          \
    if (!(cond)) {              \
      pa_set_error(__VA_ARGS__); \
      return PA_E_INVALID;      \
    }                           \
  } while (0)

// ---- switches for tools and tests ------------------------------------------
// Environment variables that force a rare path, cut a kernel short or select an ablation variant exist in
// libpyani_hip_tools.so only (built with -DPA_TOOLS from the same sources; tools/ and the tests of the rare paths load
// it).  In the product library the look-up is a null constant: the names are not even in the binary, and a stray variable
// in a worker's environment cannot change a result.
#ifdef PA_TOOLS
#include <cstdlib>
#define PA_TOOL_ENV(name) getenv(name)
#else
#define PA_TOOL_ENV(name) static_cast<const char *>(nullptr)
#endif

// ---- growable device buffer owned by the context ---------------------------
// A group of buffers may share a budget: bytes held, the most ever held, and an optional cap (0: none) past which a
// buffer refuses to grow -- with a message that names the sizes -- instead of asking the driver (fragani.hip).
struct DevBudget {
  uint64_t held = 0, peak = 0, cap = 0;
  char what[192] = {0};  // the call the buffers are growing for, for the message
};
struct DevBuf {
  void *p = nullptr;
  uint64_t bytes = 0;
  DevBudget *budget = nullptr;
  int reserve(uint64_t want) {
    if (want <= bytes) return PA_OK;
    // grow geometrically so steady-state calls never allocate
    uint64_t sz = want + want / 4 + 256;
    if (budget && budget->cap && budget->held - bytes + sz > budget->cap) {
      pa_set_error("%s: the workspace would grow to %llu bytes (%llu held; this buffer from %llu to %llu), above the cap of %llu bytes",
                   budget->what[0] ? budget->what : "device workspace", (unsigned long long)(budget->held - bytes + sz),
                   (unsigned long long)budget->held, (unsigned long long)bytes, (unsigned long long)sz, (unsigned long long)budget->cap);
      return PA_E_NOMEM;
    }
    if (p) (void)hipFree(p);
    if (budget) budget->held -= bytes;
    p = nullptr;
    bytes = 0;
    hipError_t e = hipMalloc(&p, sz);
    if (e != hipSuccess) {
      if (budget)
        pa_set_error("%s: hipMalloc(%llu) failed with %llu bytes of workspace held: %s", budget->what[0] ? budget->what : "device workspace",
                     (unsigned long long)sz, (unsigned long long)budget->held, hipGetErrorString(e));
      else
        pa_set_error("hipMalloc(%llu) failed: %s", (unsigned long long)sz, hipGetErrorString(e));
      p = nullptr;
      return PA_E_NOMEM;
    }
    bytes = sz;
    if (budget) {
      budget->held += sz;
      if (budget->held > budget->peak) budget->peak = budget->held;
    }
    return PA_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    if (budget) budget->held -= bytes;
    p = nullptr;
    bytes = 0;
  }
  template <typename T>
  T *as() const { return reinterpret_cast<T *>(p); }
};

struct ProfPhase {
  double total_ms = 0.0;
  uint64_t launches = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

struct pa_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  hipDeviceProp_t prop;
  // workspaces (sketch phase)
  DevBuf cand_keys[2], cand_vals[2];  // double-buffered radix sort storage
  DevBuf genome_blk;                  // genome start block index (u32[n+1])
  DevBuf counters;                    // small device scalars
  DevBuf hist;                        // radix histograms / scan scratch
  DevBuf flags, scan_tmp;             // compaction
  DevBuf region_off, region_cursor;   // per-genome candidate regions (LDS-sort path)
  DevBuf dirty;                       // dirty-block bitmap for callers that pass none
  // workspaces (pair phase)
  DevBuf dict_keys[2], dict_vals[2];
  DevBuf ids, post_genome, bitrows;
  // dictionary built ahead of the pair phase by pa_pair_dict_prepare (multi-GPU overlap)
  bool dict_prepared = false;
  uint64_t dict_prepared_postings = 0;
  uint32_t dict_prepared_cap = 0;
  // scalars of the hash dictionary, touched by no other phase: [0] id counter, [1] id of the key ~0 (u32 each);
  // u64 [1..2] fingerprint of the postings a prepared dictionary was built from, [3..4] of the tile that consumes it
  DevBuf dict_scalars;
  hipStream_t copy_stream = nullptr;  // uploads of pa_sketch_streamed, created on first use
  void *frag_work = nullptr;  // fragment-ANI workspace (fragani.hip), created on first use
  // pinned host scalars
  uint64_t *h_pinned = nullptr;
  // profiling
  bool prof_on = false;
  ProfPhase prof[PA_PROF_NPHASES];
  std::vector<hipEvent_t> event_pool;
};

// RAII-ish phase timer: records events around a group of launches when enabled.
struct ProfScope {
  pa_ctx *c;
  int phase;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  ProfScope(pa_ctx *ctx, int ph);
  ~ProfScope();
};

// ---- launch geometry helpers ----------------------------------------------
static inline uint32_t ceil_div_u64(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

// ---- device primitives implemented in the .hip files -----------------------
// radix_sort.hip
// Stable LSD radix sort of (u64 key, u32 val) pairs on bits [bit_lo, bit_hi) of the key
// (or of the value when by_val).  keys/vals are double buffers: *which says which one holds
// the input on entry and the result on return.
int pa_radix_sort_pairs(pa_ctx *c, uint64_t *keys[2], uint32_t *vals[2], uint64_t n, int bit_lo,
                        int bit_hi, bool by_val, int *which);
// Exclusive prefix sum of u32 -> u32 (n up to 2^32-1 elements, total must fit u32... u64 total out).
int pa_exclusive_scan_u32(pa_ctx *c, const uint32_t *d_in, uint32_t *d_out, uint64_t n,
                          uint64_t *d_total_u64 /*nullable device ptr*/);

// fragani.hip
void pa_fragani_release(pa_ctx *c);

// kmer_hash.hip
int pa_launch_kmer_hash(pa_ctx *c, const uint32_t *d_packed, const uint32_t *d_mask, const uint64_t *d_dirty, uint64_t n_blocks64,
                        const uint32_t *d_genome_blk, uint32_t n_genomes, uint32_t k, uint64_t max_hash,
                        uint64_t *d_cand_hash, uint32_t *d_cand_genome, uint64_t cap, uint64_t *d_count,
                        const uint64_t *d_region_off = nullptr, uint32_t *d_cursor = nullptr,
                        uint32_t *d_overflow = nullptr, uint64_t blk0 = 0, hipStream_t stream = nullptr);
// The launch covers arena blocks [blk0, n_blocks64) on `stream` (default: the context's); blk0 is a multiple of 64.
// d_dirty: one bit per block, set when the block needs its mask words (pa_build_dirty).
// With d_region_off != nullptr the survivors of genome g go, unordered, to d_cand_hash[region_off[g] + i),
// i < cursor[g] (zeroed by the caller); *d_overflow is set if a region was too small.

// Dirty bitmap of an arena (sketch_stream.hip): bit b of word w <-> block 64*w + b holds an invalid position, or
// the 32 positions before it do, or it is block 0.  ceil(n_blocks64 / 64) words.
int pa_build_dirty(pa_ctx *c, const uint32_t *d_mask, uint64_t n_blocks64, uint64_t *d_dirty, hipStream_t stream = nullptr);
// The bitmap the caller handed in, or one built into the context's own buffer when it handed in none.
int pa_dirty_or_build(pa_ctx *c, const uint32_t *d_mask, uint64_t n_blocks64, const uint64_t *d_dirty, const uint64_t **out);

// sketch_lds.hip: per-genome regions -> sorted unique CSR sketches, one workgroup and one LDS sort per genome
constexpr uint32_t kLdsSortMax = 16384;  // longest region the LDS sort takes
int pa_sketch_from_regions(pa_ctx *c, uint64_t *d_regions, const uint64_t *d_region_off, const uint32_t *d_cursor,
                           const uint32_t *d_overflow, uint32_t n_genomes, uint32_t longest_region, uint64_t max_hash,
                           uint64_t *d_hashes, uint64_t cap_hashes, uint64_t *d_off, uint64_t *h_total, bool *h_overflow);

// sketch_build.hip
int pa_build_sketch_csr(pa_ctx *c, const uint64_t *d_sorted_hash, const uint32_t *d_sorted_genome,
                        uint64_t n_cand, uint32_t n_genomes, uint64_t *d_hashes, uint64_t cap_hashes,
                        uint64_t *d_off, uint64_t *h_total);

// pairs_bitrow.hip / pairs_merge.hip
int pa_pairs_bitrow(pa_ctx *c, const uint64_t *d_hashes, const uint64_t *d_off, uint32_t n, uint64_t total,
                    uint32_t q0, uint32_t q1, uint32_t s0, uint32_t s1, uint32_t *d_counts);
int pa_dense_ids_sorted(pa_ctx *c, const uint64_t *d_hashes, const uint64_t *d_off, uint32_t n, uint64_t P,
                        uint64_t *n_distinct);
int pa_pairs_bitrow_hash(pa_ctx *c, const uint64_t *d_hashes, const uint64_t *d_off, const uint64_t *h_off /*nullable*/,
                         uint32_t n, uint64_t total, uint32_t q0, uint32_t q1, uint32_t s0, uint32_t s1,
                         uint32_t *d_counts);
int pa_pair_dict_prepare_impl(pa_ctx *c, const uint64_t *d_subject_hashes, uint64_t n_postings);
int pa_pairs_merge(pa_ctx *c, const uint64_t *d_hashes, const uint64_t *d_off, uint32_t n, uint32_t q0,
                   uint32_t q1, uint32_t s0, uint32_t s1, uint32_t *d_counts);

// ani.hip
int pa_launch_ani(pa_ctx *c, const uint32_t *d_counts, const uint64_t *d_off, uint32_t q0, uint32_t q1,
                  uint32_t s0, uint32_t s1, uint32_t k, double *d_identity, double *d_cov_query);
