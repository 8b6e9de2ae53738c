// capi.hip -- extern "C" entry points of libpyani_hip.so (see include/pyani_hip.h).
#include <algorithm>
#include <cstdarg>
#include <cstring>

#include "pa_internal.h"

// ---- errors -----------------------------------------------------------------
static thread_local char g_err[1024] = "";

void pa_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---- profiling ----------------------------------------------------------------
static hipEvent_t take_event(pa_ctx *c) {
  if (!c->event_pool.empty()) {
    hipEvent_t e = c->event_pool.back();
    c->event_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}

ProfScope::ProfScope(pa_ctx *ctx, int ph) : c(ctx), phase(ph) {
  if (!c->prof_on) return;
  e0 = take_event(c);
  e1 = take_event(c);
  if (e0) (void)hipEventRecord(e0, c->stream);
}

ProfScope::~ProfScope() {
  if (!c->prof_on || !e0 || !e1) return;
  (void)hipEventRecord(e1, c->stream);
  c->prof[phase].pending.emplace_back(e0, e1);
}

extern "C" {

int pa_abi_version(void) { return PA_ABI_VERSION; }
const char *pa_last_error(void) { return g_err; }

int pa_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int pa_ctx_create(int device, pa_ctx **out) {
  PA_REQUIRE(out != nullptr, "pa_ctx_create: out is null");
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    pa_set_error("no HIP device available (%s); libpyani_hip has no CPU fallback",
                 e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    return PA_E_NODEVICE;
  }
  PA_REQUIRE(device >= 0 && device < n, "pa_ctx_create: device %d out of range [0,%d)", device, n);
  PA_HIP(hipSetDevice(device));
  pa_ctx *c = new (std::nothrow) pa_ctx();
  if (!c) { pa_set_error("out of host memory"); return PA_E_NOMEM; }
  c->device = device;
  if (hipGetDeviceProperties(&c->prop, device) != hipSuccess) { delete c; pa_set_error("hipGetDeviceProperties failed"); return PA_E_HIP; }
  if (strncmp(c->prop.gcnArchName, "gfx950", 6) != 0) {
    pa_set_error("device %d is %s; this library is built for gfx950 (MI355X) only", device, c->prop.gcnArchName);
    delete c;
    return PA_E_NODEVICE;
  }
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; pa_set_error("hipStreamCreate failed"); return PA_E_HIP; }
  c->own_stream = true;
  if (hipHostMalloc(reinterpret_cast<void **>(&c->h_pinned), 64, hipHostMallocDefault) != hipSuccess) {
    (void)hipStreamDestroy(c->stream); delete c; pa_set_error("hipHostMalloc failed"); return PA_E_NOMEM;
  }
  if (c->counters.reserve(64) != PA_OK) { (void)hipHostFree(c->h_pinned); (void)hipStreamDestroy(c->stream); delete c; return PA_E_NOMEM; }
  *out = c;
  return PA_OK;
}

void pa_ctx_destroy(pa_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  DevBuf *bufs[] = {&c->cand_keys[0], &c->cand_keys[1], &c->cand_vals[0], &c->cand_vals[1], &c->genome_blk,
                    &c->counters, &c->hist, &c->flags, &c->scan_tmp, &c->region_off, &c->region_cursor, &c->dirty, &c->dict_keys[0], &c->dict_keys[1],
                    &c->dict_vals[0], &c->dict_vals[1], &c->ids, &c->post_genome, &c->bitrows, &c->dict_scalars};
  for (DevBuf *b : bufs) b->release();
  pa_fragani_release(c);
  for (auto &ph : c->prof)
    for (auto &pr : ph.pending) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
  for (hipEvent_t e : c->event_pool) (void)hipEventDestroy(e);
  if (c->h_pinned) (void)hipHostFree(c->h_pinned);
  if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); c->copy_stream = nullptr; }
  if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

int pa_ctx_set_stream(pa_ctx *c, void *hip_stream) {
  PA_REQUIRE(c != nullptr, "null context");
  PA_HIP(hipStreamSynchronize(c->stream));
  // the copy stream of pa_sketch_streamed does not depend on the compute stream: it stays as it is
  if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
  c->stream = reinterpret_cast<hipStream_t>(hip_stream);  // nullptr = the default stream
  c->own_stream = false;
  return PA_OK;
}

int pa_ctx_own_stream(pa_ctx *c) {
  PA_REQUIRE(c != nullptr, "null context");
  if (c->own_stream) return PA_OK;
  PA_HIP(hipStreamSynchronize(c->stream));
  PA_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  c->own_stream = true;
  return PA_OK;
}

int pa_ctx_sync(pa_ctx *c) {
  PA_REQUIRE(c != nullptr, "null context");
  PA_HIP(hipStreamSynchronize(c->stream));
  return PA_OK;
}

int pa_ctx_device_info(pa_ctx *c, char *name256, int *compute_units, uint64_t *global_mem) {
  PA_REQUIRE(c != nullptr, "null context");
  if (name256) { snprintf(name256, 256, "%s (%s)", c->prop.name, c->prop.gcnArchName); }
  if (compute_units) *compute_units = c->prop.multiProcessorCount;
  if (global_mem) *global_mem = (uint64_t)c->prop.totalGlobalMem;
  return PA_OK;
}

int pa_dev_alloc(pa_ctx *c, uint64_t bytes, void **d_out) {
  PA_REQUIRE(c && d_out, "pa_dev_alloc: null argument");
  PA_HIP(hipSetDevice(c->device));
  hipError_t e = hipMalloc(d_out, bytes ? bytes : 16);
  if (e != hipSuccess) { pa_set_error("hipMalloc(%llu) failed: %s", (unsigned long long)bytes, hipGetErrorString(e)); return PA_E_NOMEM; }
  return PA_OK;
}
int pa_dev_free(pa_ctx *c, void *d_ptr) {
  PA_REQUIRE(c != nullptr, "null context");
  if (d_ptr) { PA_HIP(hipStreamSynchronize(c->stream)); PA_HIP(hipFree(d_ptr)); }
  return PA_OK;
}
int pa_memcpy_h2d(pa_ctx *c, void *d_dst, const void *h_src, uint64_t bytes) {
  PA_REQUIRE(c != nullptr, "null context");
  if (bytes == 0) return PA_OK;
  PA_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, c->stream));
  PA_HIP(hipStreamSynchronize(c->stream));
  return PA_OK;
}
int pa_memcpy_d2h(pa_ctx *c, void *h_dst, const void *d_src, uint64_t bytes) {
  PA_REQUIRE(c != nullptr, "null context");
  if (bytes == 0) return PA_OK;
  PA_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
  PA_HIP(hipStreamSynchronize(c->stream));
  return PA_OK;
}
int pa_memset_d(pa_ctx *c, void *d_dst, int value, uint64_t bytes) {
  PA_REQUIRE(c != nullptr, "null context");
  if (bytes == 0) return PA_OK;
  PA_HIP(hipMemsetAsync(d_dst, value, bytes, c->stream));
  return PA_OK;
}

// ---- sketch -------------------------------------------------------------------
int pa_sketch(pa_ctx *c, const uint32_t *d_packed, const uint32_t *d_mask, const uint64_t *d_dirty_in, uint64_t arena_bases,
              const uint64_t *h_genome_start, uint32_t n_genomes, uint32_t k, uint64_t max_hash, uint64_t *d_hashes,
              uint64_t cap_hashes, uint64_t *d_off, uint64_t *h_total) {
  PA_REQUIRE(c && d_off && h_total && h_genome_start, "pa_sketch: null argument");
  PA_REQUIRE((arena_bases % PA_ALIGN_BASES) == 0, "pa_sketch: arena_bases %llu is not a multiple of %u",
             (unsigned long long)arena_bases, PA_ALIGN_BASES);
  PA_REQUIRE(arena_bases == 0 || (d_packed && d_mask), "pa_sketch: null arena");
  PA_REQUIRE(k >= 1 && k <= PA_MAX_K, "pa_sketch: k=%u outside [1,%u]", k, PA_MAX_K);
  PA_REQUIRE(h_genome_start[n_genomes] == arena_bases, "pa_sketch: genome_start[n] must equal arena_bases");
  PA_HIP(hipSetDevice(c->device));
  *h_total = 0;
  const uint64_t n_blocks = arena_bases / PA_ALIGN_BASES;
  std::vector<uint32_t> blk(n_genomes + 1);
  for (uint32_t g = 0; g <= n_genomes; ++g) {
    const uint64_t s = h_genome_start[g];
    PA_REQUIRE((s % PA_ALIGN_BASES) == 0 && (g == 0 || s >= h_genome_start[g - 1]) && s <= arena_bases,
               "pa_sketch: genome_start[%u]=%llu must be an ascending multiple of %u inside the arena", g,
               (unsigned long long)s, PA_ALIGN_BASES);
    blk[g] = (uint32_t)(s / PA_ALIGN_BASES);
  }
  PA_REQUIRE(n_blocks < (1ULL << 32), "pa_sketch: arena too large");
  PA_TRY(c->genome_blk.reserve((uint64_t)(n_genomes + 1) * sizeof(uint32_t)));
  PA_HIP(hipMemcpyAsync(c->genome_blk.p, blk.data(), (uint64_t)(n_genomes + 1) * sizeof(uint32_t),
                        hipMemcpyHostToDevice, c->stream));
  // expected survivors: one window in 2^64/(max_hash+1)
  const double frac = (max_hash == UINT64_MAX) ? 1.0 : ((double)max_hash + 1.0) / 18446744073709551616.0;
  // Per-genome candidate regions (expectation + 25 % + 128 slots).  When the longest fits an LDS sort the
  // sketches are finished by sketch_lds.hip; otherwise, or if a region overflows, by the global sort below.
  static const bool force_global = [] {
    const char *v = PA_TOOL_ENV("PA_SKETCH_SORT");
    return v && v[0] == 'g';
  }();
  std::vector<uint64_t> region_off(n_genomes + 1, 0);
  uint64_t longest_region = 0;
  for (uint32_t g = 0; g < n_genomes; ++g) {
    const uint64_t room = (uint64_t)((double)(h_genome_start[g + 1] - h_genome_start[g]) * frac * 1.25) + 128;
    longest_region = std::max(longest_region, room);
    region_off[g + 1] = region_off[g] + room;
  }
  const bool use_regions = !force_global && n_genomes > 0 && longest_region <= kLdsSortMax;
  if (use_regions) {
    PA_TRY(c->region_off.reserve((uint64_t)(n_genomes + 1) * sizeof(uint64_t)));
    PA_TRY(c->region_cursor.reserve((uint64_t)n_genomes * sizeof(uint32_t)));
    PA_HIP(hipMemcpyAsync(c->region_off.p, region_off.data(), (uint64_t)(n_genomes + 1) * sizeof(uint64_t),
                          hipMemcpyHostToDevice, c->stream));
  }
  const uint64_t *d_dirty = nullptr;
  PA_TRY(pa_dirty_or_build(c, d_mask, n_blocks, d_dirty_in, &d_dirty));
  PA_HIP(hipStreamSynchronize(c->stream));  // blk and region_off are stack-owned vectors

  if (use_regions) {
    uint32_t *d_overflow = c->counters.as<uint32_t>() + 12;
    PA_TRY(c->cand_keys[0].reserve(region_off[n_genomes] * sizeof(uint64_t)));
    PA_HIP(hipMemsetAsync(c->region_cursor.p, 0, (uint64_t)n_genomes * sizeof(uint32_t), c->stream));
    PA_HIP(hipMemsetAsync(d_overflow, 0, sizeof(uint32_t), c->stream));
    {
      ProfScope prof(c, PA_PROF_KMER_HASH);
      PA_TRY(pa_launch_kmer_hash(c, d_packed, d_mask, d_dirty, n_blocks, c->genome_blk.as<uint32_t>(), n_genomes, k, max_hash,
                                 c->cand_keys[0].as<uint64_t>(), nullptr, 0, nullptr, c->region_off.as<uint64_t>(),
                                 c->region_cursor.as<uint32_t>(), d_overflow));
    }
    bool overflow = false;
    {
      ProfScope prof(c, PA_PROF_SKETCH_SORT);
      const int st = pa_sketch_from_regions(c, c->cand_keys[0].as<uint64_t>(), c->region_off.as<uint64_t>(),
                                            c->region_cursor.as<uint32_t>(), d_overflow, n_genomes,
                                            (uint32_t)longest_region, max_hash, d_hashes, cap_hashes, d_off, h_total, &overflow);
      if (st != PA_OK) return st;
    }
    if (!overflow) return PA_OK;
    *h_total = 0;  // a region was too small (repeats, low-complexity sequence): take the general path
  }
  uint64_t cap = (uint64_t)((double)arena_bases * frac * 1.25) + 65536;
  if (cap > arena_bases) cap = arena_bases;
  uint64_t *d_count = c->counters.as<uint64_t>();
  uint64_t n_cand = 0;
  for (int attempt = 0; attempt < 2; ++attempt) {
    for (int b = 0; b < 2; ++b) {
      PA_TRY(c->cand_keys[b].reserve(cap * sizeof(uint64_t)));
      PA_TRY(c->cand_vals[b].reserve(cap * sizeof(uint32_t)));
    }
    PA_HIP(hipMemsetAsync(d_count, 0, sizeof(uint64_t), c->stream));
    {
      ProfScope prof(c, PA_PROF_KMER_HASH);
      PA_TRY(pa_launch_kmer_hash(c, d_packed, d_mask, d_dirty, n_blocks, c->genome_blk.as<uint32_t>(), n_genomes, k, max_hash,
                                 c->cand_keys[0].as<uint64_t>(), c->cand_vals[0].as<uint32_t>(), cap, d_count));
    }
    PA_HIP(hipMemcpyAsync(c->h_pinned, d_count, sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    PA_HIP(hipStreamSynchronize(c->stream));
    n_cand = c->h_pinned[0];
    if (n_cand <= cap) break;
    PA_REQUIRE(attempt == 0, "pa_sketch: candidate count changed between runs (%llu > %llu)",
               (unsigned long long)n_cand, (unsigned long long)cap);
    cap = n_cand;  // low-complexity input: rerun with the exact size
  }
  {
    ProfScope prof(c, PA_PROF_SKETCH_SORT);
    uint64_t *keys[2] = {c->cand_keys[0].as<uint64_t>(), c->cand_keys[1].as<uint64_t>()};
    uint32_t *vals[2] = {c->cand_vals[0].as<uint32_t>(), c->cand_vals[1].as<uint32_t>()};
    int which = 0;
    int hash_bits = max_hash ? 64 - __builtin_clzll(max_hash) : 0;
    hash_bits = (hash_bits + 7) & ~7;
    int genome_bits = n_genomes > 1 ? 32 - __builtin_clz(n_genomes - 1) : 0;
    genome_bits = (genome_bits + 7) & ~7;
    // LSD: least significant component (hash) first, then genome -> genome-major, hash-minor
    PA_TRY(pa_radix_sort_pairs(c, keys, vals, n_cand, 0, hash_bits, false, &which));
    PA_TRY(pa_radix_sort_pairs(c, keys, vals, n_cand, 0, genome_bits, true, &which));
    int st = pa_build_sketch_csr(c, keys[which], vals[which], n_cand, n_genomes, d_hashes, cap_hashes, d_off, h_total);
    if (st != PA_OK) return st;
  }
  return PA_OK;
}

// ---- pairs ----------------------------------------------------------------------
int pa_pair_counts_ex(pa_ctx *c, const uint64_t *d_hashes, const uint64_t *d_off, const uint64_t *h_off, uint32_t n,
                      uint32_t q0, uint32_t q1, uint32_t s0, uint32_t s1, uint32_t *d_counts, int algo) {
  PA_REQUIRE(c && d_off, "pa_pair_counts: null argument");
  PA_REQUIRE(q0 <= q1 && q1 <= n && s0 <= s1 && s1 <= n, "pa_pair_counts: ranges [%u,%u) x [%u,%u) outside [0,%u)", q0,
             q1, s0, s1, n);
  PA_REQUIRE(algo == PA_PAIRS_AUTO || algo == PA_PAIRS_BITROW || algo == PA_PAIRS_MERGE || algo == PA_PAIRS_BITROW_HASH,
             "pa_pair_counts: unknown algo %d", algo);
  PA_HIP(hipSetDevice(c->device));
  if (q0 == q1 || s0 == s1) { c->dict_prepared = false; return PA_OK; }  // empty tile: nothing to write
  PA_REQUIRE(d_counts != nullptr, "pa_pair_counts: null counts buffer");
  uint64_t total = 0;
  if (h_off) {
    total = h_off[n];
  } else {
    PA_HIP(hipMemcpyAsync(c->h_pinned, d_off + n, sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    PA_HIP(hipStreamSynchronize(c->stream));
    total = c->h_pinned[0];
  }
  PA_REQUIRE(total == 0 || d_hashes, "pa_pair_counts: null hashes");
  if (algo != PA_PAIRS_AUTO && algo != PA_PAIRS_BITROW_HASH) c->dict_prepared = false;
  switch (algo) {
    case PA_PAIRS_AUTO:
    case PA_PAIRS_BITROW_HASH:
      return pa_pairs_bitrow_hash(c, d_hashes, d_off, h_off, n, total, q0, q1, s0, s1, d_counts);
    case PA_PAIRS_BITROW:
      return pa_pairs_bitrow(c, d_hashes, d_off, n, total, q0, q1, s0, s1, d_counts);
    case PA_PAIRS_MERGE:
      return pa_pairs_merge(c, d_hashes, d_off, n, q0, q1, s0, s1, d_counts);
    default:
      pa_set_error("pa_pair_counts: unknown algo %d", algo);
      return PA_E_INVALID;
  }
}

int pa_pair_counts(pa_ctx *c, const uint64_t *d_hashes, const uint64_t *d_off, uint32_t n, uint32_t q0, uint32_t q1,
                   uint32_t s0, uint32_t s1, uint32_t *d_counts, int algo) {
  return pa_pair_counts_ex(c, d_hashes, d_off, nullptr, n, q0, q1, s0, s1, d_counts, algo);
}

int pa_pair_dict_prepare(pa_ctx *c, const uint64_t *d_subject_hashes, uint64_t n_postings) {
  PA_REQUIRE(c && (n_postings == 0 || d_subject_hashes), "pa_pair_dict_prepare: null argument");
  PA_HIP(hipSetDevice(c->device));
  return pa_pair_dict_prepare_impl(c, d_subject_hashes, n_postings);
}

int pa_ani(pa_ctx *c, const uint32_t *d_counts, const uint64_t *d_off, uint32_t q0, uint32_t q1, uint32_t s0,
           uint32_t s1, uint32_t k, double *d_identity, double *d_cov_query) {
  PA_REQUIRE(c && d_counts && d_off && d_identity && d_cov_query, "pa_ani: null argument");
  PA_REQUIRE(k >= 1 && q0 <= q1 && s0 <= s1, "pa_ani: bad k or ranges");
  PA_HIP(hipSetDevice(c->device));
  return pa_launch_ani(c, d_counts, d_off, q0, q1, s0, s1, k, d_identity, d_cov_query);
}

// ---- profiling API ----------------------------------------------------------------
int pa_prof_enable(pa_ctx *c, int on) {
  PA_REQUIRE(c != nullptr, "null context");
  c->prof_on = on != 0;
  return PA_OK;
}

static int prof_drain(pa_ctx *c) {
  PA_HIP(hipStreamSynchronize(c->stream));
  for (auto &ph : c->prof) {
    for (auto &pr : ph.pending) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
        ph.total_ms += ms;
        ph.launches += 1;
      }
      c->event_pool.push_back(pr.first);
      c->event_pool.push_back(pr.second);
    }
    ph.pending.clear();
  }
  return PA_OK;
}

int pa_prof_reset(pa_ctx *c) {
  PA_REQUIRE(c != nullptr, "null context");
  PA_TRY(prof_drain(c));
  for (auto &ph : c->prof) { ph.total_ms = 0.0; ph.launches = 0; }
  return PA_OK;
}

int pa_prof_get(pa_ctx *c, int phase, double *total_ms, uint64_t *launches) {
  PA_REQUIRE(c != nullptr && phase >= 0 && phase < PA_PROF_NPHASES, "pa_prof_get: bad phase %d", phase);
  PA_TRY(prof_drain(c));
  if (total_ms) *total_ms = c->prof[phase].total_ms;
  if (launches) *launches = c->prof[phase].launches;
  return PA_OK;
}

}  // extern "C"
