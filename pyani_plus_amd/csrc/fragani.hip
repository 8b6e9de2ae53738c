// fragani.hip -- fastANI-style fragment-mapping ANI on gfx950 (BASELINE configs[3]).
//
// Replaces one `fastANI --ql queries -r subject --fragLen F -k K --minFraction M` process per
// subject column (pyani_plus/private_cli.py:1044-1063) by an all-vs-all device pipeline.  The
// algorithm is the restatement pinned in oracle/fragani_oracle.c (published fastANI / Mashmap
// method; it reproduces the reference's 25 fixture rows and test pins exactly); every number this
// file produces (minimizers, per-fragment shared counts, kept fragments, the float sums of the
// identities) equals the oracle's.
//
//   1. minimizer_kernel   both-strand 32-bit murmur of every K-mer (first multiply by LDS table,
//                         as in kmer_hash.hip), winnowing minimum over w positions from an LDS tile,
//                         ordered compaction -> minimizers (hash, window id, contig) per contig
//   2. radix sort by hash -> dense hash ids, postings, "same hash earlier in this contig" links
//   3. query_sketch_kernel a fragment's sketch is a SLICE of its genome's minimizers (window ids
//                         inside the fragment + the one still active at the first window at which the
//                         fragment, sketched alone as fastANI does it, selects any): sort, de-duplicate
//                         in LDS, no re-hashing
//   4. seed hits           every posting of every sketch hash -> (rank of the hash in the sketch, ref contig, window id),
//                         bucketed by reference genome (one fragment per workgroup, the listed pairs' hits
//                         staged in LDS and written as whole slices); one wave per (fragment, reference
//                         genome) segment then orders its hits in registers, applies the L1 run test and
//                         slides the fragment over every candidate range: the winnowed-MinHash Jaccard
//                         of every window that could be the optimum (bit tables over query rank x
//                         reference position in LDS, no per-window sort; a tight bound from the hits' ranks
//                         and one hash comparison per stretch entry decides which windows that is); segments
//                         of at most eight hits take the hit-by-hit form (map_sparse_kernel); of equally
//                         good candidates the last
//   5. one best fragment per reference bin by atomicMax on (J, shared, s); per pair the kept
//      fragments and the float sum of their float identities in bin order (fastANI's arithmetic).
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <type_traits>
#include <vector>

#include "murmur_dev.h"
#include <optional>

#include "pa_internal.h"
#include "wave_dev.h"

namespace {

using namespace pa_dev;

constexpr int kThreads = 256;
constexpr uint32_t kSkip = 0xffffffffu;
constexpr int kQMax = 512;       // largest fragment sketch handled
constexpr int kHitCapSmall = 256;  // segments up to this many hits run with the smaller LDS footprint
constexpr int kHitCap = 512;     // seed hits of one (fragment, reference genome) segment staged in LDS
constexpr double kPercIdentity = 80.0, kConfLevel = 0.9, kPvalCutoff = 1e-3, kRefSize = 5e6;

// ============================================================== host statistics (Mashmap)
double md2j(double d, int k) { return 1.0 / (2.0 * std::exp(k * d) - 1.0); }
double j2md(double j, int k) {
  if (j == 0) return 1.0;
  if (j == 1) return 0.0;
  return (-1.0 / k) * std::log(2.0 * j / (1.0 + j));
}
double binom_cdf(int x, int n, double p) {
  if (x < 0) return 0.0;
  if (x >= n) return 1.0;
  double sum = 0.0;
  const double lp = std::log(p), lq = std::log1p(-p);
  for (int i = 0; i <= x; ++i)
    sum += std::exp(std::lgamma(n + 1.0) - std::lgamma(i + 1.0) - std::lgamma(n - i + 1.0) + i * lp + (n - i) * lq);
  return sum > 1.0 ? 1.0 : sum;
}
int binom_quantile_upper(int n, double p, double q) {
  if (p <= 0.0) return 0;
  if (p >= 1.0) return n;
  for (int x = 0; x <= n; ++x)
    if (1.0 - binom_cdf(x, n, p) <= q) return x;
  return n;
}
double md_lower_bound(double d, int s, int k) {
  const int x = binom_quantile_upper(s, md2j(d, k), (1.0 - kConfLevel) / 2.0);
  return j2md((double)x / s, k);
}
bool upper_bound_passes(int shared, int s, int k) {
  const double d = j2md((double)shared / s, k);
  return 100.0 * (1.0 - md_lower_bound(d, s, k)) >= kPercIdentity;
}
int min_shared_for(int s, int k) {
  for (int x = 0; x <= s; ++x)
    if (upper_bound_passes(x, s, k)) return x;
  return s + 1;
}
int relaxed_min_hits(int s, int k) {
  int best = (int)std::ceil(1.0 * s * md2j(1.0 - kPercIdentity / 100.0, k));
  for (int i = best; i >= 0; --i) {
    if (upper_bound_passes(i, s, k)) best = i; else break;
  }
  return best;
}
// Winnowing window: the smallest sketch size of Mashmap's list 1, 2, 5, 10, 20, 30, ... whose random-match p-value
// over a 5 Mb reference is <= 1e-3, then w = 2*fragLen/sketch (24 for k=16, fragLen=3000: the window fastANI logs).
int window_size_for(int k, int frag_len) {
  auto passes = [&](int s) {
    const double px = 1.0 / (1.0 + std::pow(4.0, k) / frag_len);
    const double r = px * px / (px + px - px * px);
    const int x = relaxed_min_hits(s, k);
    const double comp = x == 0 ? 1.0 : 1.0 - binom_cdf(x - 1, s, r);
    return kRefSize * comp <= kPvalCutoff;
  };
  int s = 1;
  bool found = false;
  for (int cand : {1, 2, 5}) { s = cand; if ((found = passes(s))) break; }
  for (int cand = 10; cand < frag_len && !found; cand += 10) { s = cand; found = passes(s); }
  int w = (int)(2.0 * frag_len / s);
  if (w < 1) w = 1;
  if (w > frag_len) w = frag_len;
  return w;
}

// ============================================================== device helpers
__device__ __forceinline__ uint32_t lower_bound_u32(const uint32_t *__restrict__ v, uint32_t lo, uint32_t hi, uint32_t x) {
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (v[mid] < x) lo = mid + 1; else hi = mid;
  }
  return lo;
}
// first minimizer of contig c with window id >= x, through the per-contig bucket index
// (bucket b of a contig = window ids [b*256, b*256+256); bucket_first holds global minimizer indices and
// one closing entry per contig): two dependent loads plus a ~20-entry search instead of a 19-step
// binary search whose every probe misses the caches
constexpr uint32_t kBucketShift = 8;
__device__ __forceinline__ uint32_t wpos_lower_bound(const uint32_t *__restrict__ mini_wpos,
                                                     const uint32_t *__restrict__ bucket_first, uint32_t bucket_base,
                                                     uint32_t n_buckets, uint32_t x) {
  uint32_t b = x >> kBucketShift;
  if (b >= n_buckets) b = n_buckets;  // beyond the contig: the closing entry
  const uint32_t lo = bucket_first[bucket_base + b];
  const uint32_t hi = b < n_buckets ? bucket_first[bucket_base + b + 1] : lo;
  return lower_bound_u32(mini_wpos, lo, hi, x);
}

__device__ __forceinline__ uint32_t contig_of(const uint64_t *__restrict__ start, uint32_t n, uint64_t pos) {
  uint32_t lo = 0, hi = n;  // largest c with start[c] <= pos (start[0] == 0)
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (start[mid] <= pos) lo = mid; else hi = mid;
  }
  return lo;
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) { return pa_dev::wave_sum_dpp(v); }
__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t v, uint32_t lane) {
  (void)lane;
  return pa_dev::wave_incl_scan_dpp(v) - v;
}

// The kernels, in pipeline order (one translation unit: the files are included here, inside the anonymous namespace)
#include "fragani_index.inc"   // 1. minimizers, 2. dictionary of minimizer hashes, postings, frequency cut
#include "fragani_seed.inc"    // 3. fragment sketches, 4. seed hits bucketed by reference genome, segment lists
#include "fragani_map.inc"     // 4. (continued) map_segments_kernel: L1, the exact slide, bounds
#include "fragani_sparse.inc"  // 4b. map_sparse_kernel: segments of a handful of seed hits

// ============================================================== 5. per-pair reduction
// One wave per (query of the batch, reference genome): kept fragments and the sum of their identities.  fastANI holds the
// identities as floats and adds them up in a float, in (contig, bin) order; a float sum depends on its order, so the wave
// adds in exactly that order: 64 bins per load, then one addition per kept bin through a scalar loop over the wave.
__global__ __launch_bounds__(64) void reduce_pairs_kernel(const unsigned long long *__restrict__ table,
                                                          uint64_t table_stride,
                                                          const uint32_t *__restrict__ genome_bin_off,
                                                          uint32_t n_genomes, const float *__restrict__ ident_tab,
                                                          uint32_t *__restrict__ matched, double *__restrict__ ident_sum) {
  const uint32_t lane = threadIdx.x;
  const uint32_t q = blockIdx.x / n_genomes, r = blockIdx.x % n_genomes;
  const uint32_t b0 = genome_bin_off[r], b1 = genome_bin_off[r + 1];
  uint32_t cnt = 0;
  float sum = 0.0f;
  for (uint32_t base = b0; base < b1; base += 64) {
    const uint32_t bidx = base + lane;
    const unsigned long long v = bidx < b1 ? table[(uint64_t)q * table_stride + bidx] : 0ull;
    float id = 0.0f;
    if (v) {
      const uint32_t shared = (uint32_t)(v >> 16) & 0xffffu, s = (uint32_t)v & 0xffffu;
      id = ident_tab[(uint64_t)s * (kQMax + 1) + shared];
    }
    uint64_t kept = __ballot(v != 0ull);
    cnt += (uint32_t)__popcll(kept);
    while (kept) {
      const int l = __builtin_ctzll(kept);
      kept &= kept - 1;
      sum = __fadd_rn(sum, __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, id), l)));
    }
  }
  if (lane == 0) {
    matched[(uint64_t)q * n_genomes + r] = cnt;
    ident_sum[(uint64_t)q * n_genomes + r] = (double)sum;
  }
}

// ============================================================== host driver
template <typename T>
int upload(pa_ctx *c, DevBuf &buf, const std::vector<T> &v) {
  PA_TRY(buf.reserve(v.size() * sizeof(T) + 16));
  if (!v.empty()) PA_HIP(hipMemcpyAsync(buf.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, c->stream));
  return PA_OK;
}

struct FragWork {
  DevBuf contig_start, contig_len, contig_genome, block_counts, block_offsets, mini_hash, mini_wpos, mini_contig,
      contig_mini_off, keys[2], vals[2], flags, mini_id, post_start, prev_same, sorted_idx, frag_contig, frag_no,
      frag_genome_local, q_hash, q_pos, q_id, q_s, hit_count, hit_off, hkeys[2], hvals[2], seg_start, tab_min_hits,
      tab_min_shared, ident_tab, contig_bin_off, genome_bin_off, table, matched, ident_sum, scalars, run_g, seg_list, seg2_a0, seg2_nh, post_cw, seg_a0, seg_nh, genome_first_contig,
      contig_bucket_off, bucket_first, post_g, seg_rec, hash_cut, q_cut, post_cw2, post_g2, run_hist, long_runs, frag_d, uniq_hash, lookup_at, seg_f, seg2_f, amb_pos, amb_byte, seg_over, q_tab;
  // the arena's residues that are neither ACGT nor N (pa_fragani_set_ambiguous), and the arena they belong to
  const void *amb_for = nullptr;
  uint32_t amb_n = 0;
  std::vector<uint64_t> amb_host_pos;  // what the device arrays hold: a call that hands over the same list again changes nothing
  std::vector<uint8_t> amb_host_byte;
  AmbiguousList ambiguous(const void *d_packed) const {
    const bool mine = amb_n && amb_for == d_packed;
    return AmbiguousList{mine ? amb_pos.as<uint64_t>() : nullptr, mine ? amb_byte.as<uint8_t>() : nullptr, mine ? amb_n : 0u};
  }
  // the reference index (stages 1 and 2) of the last pa_fragani(_ex) call, for PA_FRAGANI_REUSE_INDEX
  bool index_valid = false;
  const void *index_packed = nullptr;
  uint64_t index_arena_bases = 0;
  uint32_t index_contigs = 0, index_genomes = 0, index_k = 0, index_frag_len = 0, index_m = 0, index_ids = 0;
  uint32_t index_ref0 = 0, index_ref1 = 0;  // the reference genomes whose minimizers the dictionary holds
  uint32_t index_lookup_bits = 10;          // log2 of the slots of the look-up table by hash value (such a dictionary only)
  uint32_t tables_k = 0;                    // the k the tables by sketch size (tab_min_hits, tab_min_shared, ident_tab) on the device were made for; 0: none
  uint64_t index_key_room = 0;              // keys of one half of the sort's key buffer (W.keys[0] holds both halves)
  int index_which = 0;
  // every buffer of the workspace shares one budget: what pa_fragani_workspace reports and pa_fragani_set_workspace_cap bounds
  DevBudget budget;
  template <class Fn>
  void each_buffer(Fn &&fn) {
    DevBuf *all[] = {&contig_start, &contig_len, &contig_genome, &block_counts, &block_offsets, &mini_hash, &mini_wpos,
                     &mini_contig, &contig_mini_off, &keys[0], &keys[1], &vals[0], &vals[1], &flags, &mini_id,
                     &post_start, &prev_same, &sorted_idx, &frag_contig, &frag_no, &frag_genome_local, &q_hash, &q_pos,
                     &q_id, &q_s, &hit_count, &hit_off, &hkeys[0], &hkeys[1], &hvals[0], &hvals[1], &seg_start,
                     &tab_min_hits, &tab_min_shared, &ident_tab, &contig_bin_off, &genome_bin_off, &table, &matched,
                     &ident_sum, &scalars, &run_g, &seg_list, &seg2_a0, &seg2_nh, &post_cw, &seg_a0, &seg_nh, &genome_first_contig, &contig_bucket_off, &bucket_first, &post_g, &seg_rec, &hash_cut, &q_cut, &post_cw2, &post_g2, &run_hist, &long_runs, &frag_d, &uniq_hash, &lookup_at, &seg_f, &seg2_f, &amb_pos, &amb_byte, &seg_over, &q_tab};
    for (DevBuf *b : all) fn(*b);
  }
  FragWork() { each_buffer([this](DevBuf &b) { b.budget = &budget; }); }
  FragWork(const FragWork &) = delete;
  FragWork &operator=(const FragWork &) = delete;
  ~FragWork() { each_buffer([](DevBuf &b) { b.release(); }); }
};

// The workspace lives in the context like the sketch and pair workspaces do: buffers only grow, and a
// second call does not pay for returning tens of GB to the driver and asking for them again.
FragWork &frag_work(pa_ctx *c) {
  if (!c->frag_work) c->frag_work = new FragWork();
  return *static_cast<FragWork *>(c->frag_work);
}

// Mashmap's frequency cut (see posting_run_hist_kernel): thresholds per reference genome on the host, from the histograms
// of the run lengths; the runs at or above them leave the posting lists.  `heads`: 1 at the first posting of every hash,
// `ids_before`: the hashes before a posting's own (the scan of `heads`), `sorted_idx`: the postings' minimizers, `idx_spare`: m words; `scratch`: 2 m words,
// free at this point.
int cut_frequent_postings(pa_ctx *c, FragWork &W, const uint32_t *d_heads, const uint32_t *d_ids_before, uint32_t *d_sorted_idx,
                          uint32_t *d_idx_spare, uint32_t *scratch, uint32_t m,
                          uint32_t n_ids, const uint32_t *h_contig_genome, uint32_t n_contigs, uint32_t n_genomes) {
  uint32_t *scratch_a = scratch, *scratch_b = scratch + m;
  PA_TRY(W.hash_cut.reserve(((uint64_t)n_ids / 32 + 2) * 4));
  PA_HIP(hipMemsetAsync(W.hash_cut.p, 0, ((uint64_t)n_ids / 32 + 2) * 4, c->stream));
  if (const char *v = PA_TOOL_ENV("PA_FRAGANI_NO_FREQ_CUT")) { if (atoi(v)) return PA_OK; }  // tools: the seeds as rounds 1-4 looked them up
  constexpr uint32_t kOverCap = 1u << 20;
  std::vector<uint32_t> threshold(n_genomes, 0xffffffffu);
  bool any = false;
  {
    const uint64_t hist_words = (uint64_t)n_genomes * kFreqBins + n_genomes + 1;  // histograms, duplicate counts, overflow cursor
    PA_TRY(W.run_hist.reserve(hist_words * 4));
    PA_TRY(W.long_runs.reserve((uint64_t)kOverCap * 8));
    uint32_t *d_hist = W.run_hist.as<uint32_t>(), *d_dups = d_hist + (uint64_t)n_genomes * kFreqBins, *d_over_n = d_dups + n_genomes;
    PA_HIP(hipMemsetAsync(d_hist, 0, hist_words * 4, c->stream));
    // the runs numbered: flags in scratch_a, runs before each posting in scratch_b, the runs' first postings in idx_spare
    // (m + 1 words: every buffer here was reserved with room to spare)
    const uint32_t gm0 = ceil_div_u64(m, kThreads);
    hipLaunchKernelGGL(posting_run_flags_kernel, dim3(gm0), dim3(kThreads), 0, c->stream, d_heads, W.post_g.as<uint16_t>(), m, scratch_a);
    PA_TRY(pa_exclusive_scan_u32(c, scratch_a, scratch_b, m, nullptr));
    hipLaunchKernelGGL(posting_run_starts_kernel, dim3(gm0), dim3(kThreads), 0, c->stream, scratch_a, scratch_b, m, d_idx_spare);
    hipLaunchKernelGGL(posting_run_hist_kernel, dim3(gm0), dim3(kThreads), 0, c->stream, scratch_a, scratch_b, d_idx_spare,
                       W.post_g.as<uint16_t>(), m, d_hist, d_dups, W.long_runs.as<uint2>(), kOverCap, d_over_n);
    std::vector<uint32_t> h((size_t)hist_words);
    PA_HIP(hipMemcpyAsync(h.data(), d_hist, hist_words * 4, hipMemcpyDeviceToHost, c->stream));
    PA_HIP(hipStreamSynchronize(c->stream));
    const uint32_t n_over = h[hist_words - 1];
    PA_REQUIRE(n_over <= kOverCap, "pa_fragani: %u (minimizer, genome) pairs with %u or more occurrences", n_over, kFreqBins - 1);
    std::vector<uint2> h_over(n_over);
    if (n_over) PA_HIP(hipMemcpy(h_over.data(), W.long_runs.p, (size_t)n_over * 8, hipMemcpyDeviceToHost));
    // minimizers per genome: its contigs' shares of the minimizer array
    std::vector<uint32_t> cmo(n_contigs + 1);
    PA_HIP(hipMemcpy(cmo.data(), W.contig_mini_off.p, (size_t)(n_contigs + 1) * 4, hipMemcpyDeviceToHost));
    std::vector<uint64_t> n_min(n_genomes, 0);
    for (uint32_t ci = 0; ci < n_contigs; ++ci) n_min[h_contig_genome[ci]] += cmo[ci + 1] - cmo[ci];
    std::vector<std::vector<uint32_t>> long_runs(n_genomes);
    for (const uint2 &e : h_over) long_runs[e.x].push_back(e.y);
    for (uint32_t g = 0; g < n_genomes; ++g) {
      const uint32_t *bars = h.data() + (uint64_t)g * kFreqBins;
      const uint64_t uniq = n_min[g] - h[(uint64_t)n_genomes * kFreqBins + g];
      const int64_t to_ignore = (int64_t)((float)uniq * 0.001f / 100);
      // the bars from the most frequent minimizers down: (count, distinct minimizers with that count)
      std::vector<std::pair<uint32_t, uint64_t>> top;
      std::sort(long_runs[g].begin(), long_runs[g].end(), std::greater<uint32_t>());
      for (uint32_t v : long_runs[g]) { if (!top.empty() && top.back().first == v) ++top.back().second; else top.push_back({v, 1}); }
      uint64_t repeated = long_runs[g].size();
      for (uint32_t cnt = kFreqBins - 2; cnt >= 2; --cnt) if (bars[cnt]) { top.push_back({cnt, bars[cnt]}); repeated += bars[cnt]; }
      if (uniq > repeated) top.push_back({1u, uniq - repeated});
      int64_t sum = 0;
      for (const auto &bar : top) {
        sum += (int64_t)bar.second;
        if (sum < to_ignore) threshold[g] = bar.first;
        else { if (sum == to_ignore) threshold[g] = bar.first; break; }
      }
      any = any || threshold[g] != 0xffffffffu;
    }
  }
  if (!any) return PA_OK;
  // take the runs out: flags, their prefix sums, the postings moved into the workspace's second pair of arrays (which then
  // change places with the first), the lists' bounds rewritten
  PA_TRY(upload(c, W.run_hist, threshold));  // (the histograms are on the host by now)
  PA_TRY(W.post_cw2.reserve((uint64_t)m * 8));
  PA_TRY(W.post_g2.reserve((uint64_t)m * 2 + 16));
  const uint32_t gm = ceil_div_u64(m, kThreads);
  hipLaunchKernelGGL(posting_cut_flags_kernel, dim3(gm), dim3(kThreads), 0, c->stream, d_heads, d_ids_before, scratch_b, d_idx_spare,
                     W.post_g.as<uint16_t>(), m, W.run_hist.as<uint32_t>(), scratch_a, W.hash_cut.as<uint32_t>());
  hipLaunchKernelGGL(mark_cut_minimizers_kernel, dim3(gm), dim3(kThreads), 0, c->stream, d_heads, d_ids_before, d_sorted_idx, m,
                     W.hash_cut.as<uint32_t>(), W.mini_id.as<uint32_t>());
  PA_TRY(pa_exclusive_scan_u32(c, scratch_a, scratch_b, m, W.scalars.as<uint64_t>()));
  PA_HIP(hipMemcpyAsync(c->h_pinned, W.scalars.p, 8, hipMemcpyDeviceToHost, c->stream));
  PA_HIP(hipStreamSynchronize(c->stream));
  const uint32_t kept = (uint32_t)c->h_pinned[0];
  if (kept != m) {
    hipLaunchKernelGGL(posting_compact_kernel, dim3(gm), dim3(kThreads), 0, c->stream, scratch_a, scratch_b, W.post_cw.as<uint64_t>(),
                       W.post_g.as<uint16_t>(), d_sorted_idx, m, W.post_cw2.as<uint64_t>(), W.post_g2.as<uint16_t>(), d_idx_spare);
    hipLaunchKernelGGL(posting_starts_kernel, dim3(ceil_div_u64((uint64_t)n_ids + 1, kThreads)), dim3(kThreads), 0, c->stream,
                       W.post_start.as<uint32_t>(), n_ids, scratch_b, m, kept);
    std::swap(W.post_cw, W.post_cw2);
    std::swap(W.post_g, W.post_g2);
    // (the postings' minimizer indices -- what the path for more than 8 192 genomes reads -- go back where they are looked for)
    PA_HIP(hipMemcpyAsync(d_sorted_idx, d_idx_spare, (uint64_t)kept * 4, hipMemcpyDeviceToDevice, c->stream));
  }
  return PA_OK;
}

template <int K>
int run_minimizers(pa_ctx *c, FragWork &W, const uint32_t *d_packed, const uint32_t *d_mask, uint64_t arena_bases,
                   uint32_t n_contigs, int w, uint32_t *m_out, const std::function<void()> *meanwhile) {
  const uint32_t blocks = ceil_div_u64(arena_bases, kOwn);
  PA_TRY(W.block_counts.reserve((uint64_t)blocks * 8));  // the look-back words of minimizer_kernel
  PA_TRY(W.scalars.reserve(kMiniScalarBytes));
  PA_TRY(W.block_offsets.reserve((uint64_t)blocks * 4 + 16));  // the contig of every tile's first position
  hipLaunchKernelGGL(tile_contig_kernel, dim3(ceil_div_u64(blocks, kThreads)), dim3(kThreads), 0, c->stream, W.contig_start.as<uint64_t>(), n_contigs,
                     blocks, W.block_offsets.as<uint32_t>());
  // expected density of winnowed minimizers is 2 / (w + 1); the arrays are sized a quarter above that and the run is
  // repeated with the exact size should a low-complexity data set need more
  uint64_t cap = (uint64_t)((double)arena_bases * 2.5 / (double)(w + 1)) + (1u << 20);
  if (const char *v = PA_TOOL_ENV("PA_FRAGANI_MINIMIZER_ROOM")) cap = std::max<uint64_t>(1, strtoull(v, nullptr, 10));  // tests: force the repeat
  // tiles are drawn from one counter per XCD; from a single counter when a wait of the chained scan ran out under that
  // scheme (see the kernel) -- or when a tool asks for it
  uint32_t ticket_mode = 1;
  if (const char *v = PA_TOOL_ENV("PA_FRAGANI_ONE_TICKET")) { if (atoi(v)) ticket_mode = 0; }
  for (int attempt = 0; attempt < 2; ++attempt) {
    PA_REQUIRE(cap < (1ULL << 31), "fragment ANI: room for %llu minimizers exceeds the 31-bit index space", (unsigned long long)cap);
    PA_TRY(W.mini_hash.reserve(cap * 4 + 16));
    PA_TRY(W.mini_wpos.reserve(cap * 4 + 16));
    PA_TRY(W.mini_contig.reserve(cap * 4 + 16));
    PA_HIP(hipMemsetAsync(W.block_counts.p, 0, (uint64_t)blocks * 8, c->stream));
    PA_HIP(hipMemsetAsync(W.scalars.p, 0, kMiniScalarBytes, c->stream));
    hipLaunchKernelGGL((minimizer_kernel<K>), dim3(blocks), dim3(kMiniThreads), 0, c->stream,
                       d_packed, d_mask, arena_bases, W.contig_start.as<uint64_t>(), W.contig_len.as<uint32_t>(), n_contigs, w,
                       W.block_counts.as<unsigned long long>(), W.scalars.as<uint32_t>(), (uint32_t)cap,
                       W.mini_hash.as<uint32_t>(), W.mini_wpos.as<uint32_t>(), W.mini_contig.as<uint32_t>(), blocks, W.ambiguous(d_packed), ticket_mode,
                       W.block_offsets.as<uint32_t>());
    PA_HIP(hipGetLastError());
    PA_HIP(hipMemcpyAsync(c->h_pinned, W.scalars.p, 16, hipMemcpyDeviceToHost, c->stream));
    if (meanwhile && *meanwhile) {  // the caller's host-side bookkeeping, while the kernel runs (once)
      (*meanwhile)();
      meanwhile = nullptr;
    }
    PA_HIP(hipStreamSynchronize(c->stream));
    const uint32_t *h = reinterpret_cast<const uint32_t *>(c->h_pinned);
    bool ran_out = h[2] != 0;
    if (const char *v = PA_TOOL_ENV("PA_FRAGANI_TICKET_TIMEOUT")) { if (atoi(v) && ticket_mode != 0) ran_out = true; }  // tests: as if a wait had run out
    if (ran_out && ticket_mode != 0) {  // a wait ran out with the tiles drawn per XCD: once more, from one counter
      ticket_mode = 0;
      --attempt;
      continue;
    }
    PA_REQUIRE(h[2] == 0, "fragment ANI: the minimizer scan gave up waiting for a tile (%u tiles)", blocks);
    const uint64_t m = h[1];
    if (m <= cap) {
      *m_out = (uint32_t)m;
      return PA_OK;
    }
    cap = m;
  }
  pa_set_error("fragment ANI: minimizer count changed between two runs over the same arena");
  return PA_E_HIP;
}

int dispatch_minimizers(pa_ctx *c, FragWork &W, const uint32_t *d_packed, const uint32_t *d_mask, uint64_t arena_bases,
                        uint32_t n_contigs, uint32_t k, int w, uint32_t *m_out, const std::function<void()> *meanwhile = nullptr) {
  switch (k) {
    case 8: return run_minimizers<8>(c, W, d_packed, d_mask, arena_bases, n_contigs, w, m_out, meanwhile);
    case 9: return run_minimizers<9>(c, W, d_packed, d_mask, arena_bases, n_contigs, w, m_out, meanwhile);
    case 10: return run_minimizers<10>(c, W, d_packed, d_mask, arena_bases, n_contigs, w, m_out, meanwhile);
    case 11: return run_minimizers<11>(c, W, d_packed, d_mask, arena_bases, n_contigs, w, m_out, meanwhile);
    case 12: return run_minimizers<12>(c, W, d_packed, d_mask, arena_bases, n_contigs, w, m_out, meanwhile);
    case 13: return run_minimizers<13>(c, W, d_packed, d_mask, arena_bases, n_contigs, w, m_out, meanwhile);
    case 14: return run_minimizers<14>(c, W, d_packed, d_mask, arena_bases, n_contigs, w, m_out, meanwhile);
    case 15: return run_minimizers<15>(c, W, d_packed, d_mask, arena_bases, n_contigs, w, m_out, meanwhile);
    case 16: return run_minimizers<16>(c, W, d_packed, d_mask, arena_bases, n_contigs, w, m_out, meanwhile);
    default:
      pa_set_error("fragment ANI: k=%u outside [8,16] (fastANI itself stops at 16)", k);
      return PA_E_INVALID;
  }
}

int stage_contigs(pa_ctx *c, FragWork &W, const uint64_t *h_contig_start, const uint32_t *h_contig_len,
                  const uint32_t *h_contig_genome, uint32_t n_contigs, uint32_t n_genomes, uint64_t arena_bases) {
  PA_REQUIRE(n_contigs >= 1 && n_contigs < (1u << 20), "fragment ANI: %u contigs (supported: 1 .. 2^20-1)", n_contigs);
  std::vector<uint64_t> cs(h_contig_start, h_contig_start + n_contigs);
  std::vector<uint32_t> cl(h_contig_len, h_contig_len + n_contigs), cg(h_contig_genome, h_contig_genome + n_contigs);
  for (uint32_t i = 0; i < n_contigs; ++i) {
    PA_REQUIRE(cs[i] + cl[i] <= arena_bases && (i == 0 || cs[i] >= cs[i - 1] + cl[i - 1]) && cg[i] < n_genomes &&
                   (i == 0 || cg[i] >= cg[i - 1]) && cl[i] < (1u << 24),
               "fragment ANI: contig %u is out of order, outside the arena, or longer than 2^24", i);
  }
  cs[0] = 0;  // positions before the first contig (none in practice) resolve to contig 0
  PA_TRY(upload(c, W.contig_start, cs));
  PA_TRY(upload(c, W.contig_len, cl));
  PA_TRY(upload(c, W.contig_genome, cg));
  PA_HIP(hipStreamSynchronize(c->stream));
  return PA_OK;
}

}  // namespace

void pa_fragani_release(pa_ctx *c) {
  delete static_cast<FragWork *>(c->frag_work);
  c->frag_work = nullptr;
}

extern "C" {

int pa_fragani_window(uint32_t k, uint32_t frag_len) { return window_size_for((int)k, (int)frag_len); }

// The fragment-ANI workspace of a context: bytes held now, the most it ever held (buffers only grow: the same until
// pa_fragani_release), and the cap (0: none).  A cap bounds what later calls may ask the device for: a call that would
// pass it ends with PA_E_NOMEM and a message naming the call and the sizes, the workspace as it was.
int pa_fragani_workspace(pa_ctx *c, uint64_t *held_bytes, uint64_t *peak_bytes, uint64_t *cap_bytes) {
  PA_REQUIRE(c, "pa_fragani_workspace: null context");
  const FragWork *W = static_cast<const FragWork *>(c->frag_work);
  if (held_bytes) *held_bytes = W ? W->budget.held : 0;
  if (peak_bytes) *peak_bytes = W ? W->budget.peak : 0;
  if (cap_bytes) *cap_bytes = W ? W->budget.cap : 0;
  return PA_OK;
}
int pa_fragani_set_workspace_cap(pa_ctx *c, uint64_t cap_bytes) {
  PA_REQUIRE(c, "pa_fragani_set_workspace_cap: null context");
  frag_work(c).budget.cap = cap_bytes;
  return PA_OK;
}

int pa_fragani_set_ambiguous(pa_ctx *c, const uint32_t *d_packed, const uint64_t *h_pos, const uint8_t *h_byte, uint64_t n) {
  PA_REQUIRE(c && (n == 0 || (d_packed && h_pos && h_byte)), "pa_fragani_set_ambiguous: null argument");
  PA_REQUIRE(n < (1ULL << 31), "pa_fragani_set_ambiguous: %llu residues (limit 2^31)", (unsigned long long)n);
  for (uint64_t i = 1; i < n; ++i)
    PA_REQUIRE(h_pos[i] > h_pos[i - 1], "pa_fragani_set_ambiguous: positions must ascend (entry %llu)", (unsigned long long)i);
  PA_HIP(hipSetDevice(c->device));
  FragWork &W = frag_work(c);
  if (n == 0 && W.amb_n == 0) return PA_OK;
  if (n && W.amb_n == n && W.amb_for == (const void *)d_packed && memcmp(W.amb_host_pos.data(), h_pos, n * 8) == 0 &&
      memcmp(W.amb_host_byte.data(), h_byte, n) == 0)
    return PA_OK;  // the list the context holds already (a reusable index stays reusable)
  W.index_valid = false;  // an index built with another list (or none) hashed those residues differently
  W.amb_for = nullptr;
  W.amb_n = 0;
  W.amb_host_pos.clear();
  W.amb_host_byte.clear();
  if (n == 0) return PA_OK;
  W.amb_host_pos.assign(h_pos, h_pos + n);
  W.amb_host_byte.assign(h_byte, h_byte + n);
  PA_TRY(W.amb_pos.reserve(n * 8));
  PA_TRY(W.amb_byte.reserve(n + 16));
  PA_HIP(hipMemcpyAsync(W.amb_pos.p, h_pos, n * 8, hipMemcpyHostToDevice, c->stream));
  PA_HIP(hipMemcpyAsync(W.amb_byte.p, h_byte, n, hipMemcpyHostToDevice, c->stream));
  PA_HIP(hipStreamSynchronize(c->stream));
  W.amb_for = d_packed;
  W.amb_n = (uint32_t)n;
  return PA_OK;
}

int pa_fragani_tables(uint32_t k, uint32_t s_max, uint32_t *h_min_hits, uint32_t *h_min_shared) {
  if (!h_min_hits || !h_min_shared) { pa_set_error("pa_fragani_tables: null argument"); return PA_E_INVALID; }
  h_min_hits[0] = h_min_shared[0] = 0;
  for (uint32_t s = 1; s <= s_max; ++s) {
    const int mh = relaxed_min_hits((int)s, (int)k);
    h_min_hits[s] = (uint32_t)(mh < 1 ? 1 : mh);
    h_min_shared[s] = (uint32_t)min_shared_for((int)s, (int)k);
  }
  return PA_OK;
}

// fastANI holds the Jaccard estimate, the Mash distance and the identity as floats (Mashmap's j2md takes and returns a
// float, and nucIdentity = 100 * (1 - mash_dist) is float arithmetic): the value here is that float, widened.
double pa_fragani_identity(uint32_t shared, uint32_t s, uint32_t k) {
  if (!s) return 0.0;
  const float j = (float)(1.0 * shared / s);
  float d;
  if (j == 0) d = 1.0f;
  else if (j == 1) d = 0.0f;
  else d = (float)((-1.0 / (int)k) * std::log(2.0 * j / (1 + j)));
  return (double)(100 * (1 - d));
}

int pa_fragani_sketch(pa_ctx *c, const uint32_t *d_packed, const uint32_t *d_mask, uint64_t arena_bases,
                      const uint64_t *h_contig_start, const uint32_t *h_contig_len, const uint32_t *h_contig_genome,
                      uint32_t n_contigs, uint32_t n_genomes, uint32_t k, uint32_t window, uint32_t *h_hash,
                      uint32_t *h_wpos, uint32_t *h_contig, uint64_t cap, uint64_t *n_out) {
  PA_REQUIRE(c && d_packed && d_mask && n_out, "pa_fragani_sketch: null argument");
  PA_REQUIRE(window >= 1 && window <= 64, "pa_fragani_sketch: window %u outside [1,64]", window);
  PA_HIP(hipSetDevice(c->device));
  FragWork &W = frag_work(c);
  W.index_valid = false;  // the minimizer arrays are about to be overwritten
  PA_TRY(stage_contigs(c, W, h_contig_start, h_contig_len, h_contig_genome, n_contigs, n_genomes, arena_bases));
  uint32_t m = 0;
  PA_TRY(dispatch_minimizers(c, W, d_packed, d_mask, arena_bases, n_contigs, k, (int)window, &m));
  *n_out = m;
  if (m > cap) { pa_set_error("pa_fragani_sketch: %u minimizers, room for %llu", m, (unsigned long long)cap); return PA_E_CAPACITY; }
  if (m) {
    PA_HIP(hipMemcpyAsync(h_hash, W.mini_hash.p, (uint64_t)m * 4, hipMemcpyDeviceToHost, c->stream));
    PA_HIP(hipMemcpyAsync(h_wpos, W.mini_wpos.p, (uint64_t)m * 4, hipMemcpyDeviceToHost, c->stream));
    PA_HIP(hipMemcpyAsync(h_contig, W.mini_contig.p, (uint64_t)m * 4, hipMemcpyDeviceToHost, c->stream));
  }
  PA_HIP(hipStreamSynchronize(c->stream));
  return PA_OK;
}

int pa_fragani(pa_ctx *c, const uint32_t *d_packed, const uint32_t *d_mask, uint64_t arena_bases,
               const uint64_t *h_contig_start, const uint32_t *h_contig_len, const uint32_t *h_contig_genome,
               uint32_t n_contigs, uint32_t n_genomes, uint32_t k, uint32_t frag_len, uint32_t ref0, uint32_t ref1,
               uint32_t *h_total_frags, uint32_t *h_matched, double *h_ident_sum) {
  return pa_fragani_ex(c, d_packed, d_mask, arena_bases, h_contig_start, h_contig_len, h_contig_genome, n_contigs,
                       n_genomes, k, frag_len, 0, n_genomes, ref0, ref1, 0, h_total_frags, h_matched, h_ident_sum);
}

static int fragani_ex_impl(pa_ctx *c, const uint32_t *d_packed, const uint32_t *d_mask, uint64_t arena_bases,
                           const uint64_t *h_contig_start, const uint32_t *h_contig_len, const uint32_t *h_contig_genome,
                           uint32_t n_contigs, uint32_t n_genomes, uint32_t k, uint32_t frag_len, uint32_t qry0, uint32_t qry1,
                           uint32_t ref0, uint32_t ref1, uint32_t flags, uint32_t *h_total_frags, uint32_t *h_matched,
                           double *h_ident_sum);
int pa_fragani_ex(pa_ctx *c, const uint32_t *d_packed, const uint32_t *d_mask, uint64_t arena_bases,
                  const uint64_t *h_contig_start, const uint32_t *h_contig_len, const uint32_t *h_contig_genome,
                  uint32_t n_contigs, uint32_t n_genomes, uint32_t k, uint32_t frag_len, uint32_t qry0, uint32_t qry1,
                  uint32_t ref0, uint32_t ref1, uint32_t flags, uint32_t *h_total_frags, uint32_t *h_matched,
                  double *h_ident_sum) {
  const int status = fragani_ex_impl(c, d_packed, d_mask, arena_bases, h_contig_start, h_contig_len, h_contig_genome, n_contigs, n_genomes, k,
                                     frag_len, qry0, qry1, ref0, ref1, flags, h_total_frags, h_matched, h_ident_sum);
  if (status == PA_E_NOMEM && c && c->frag_work) {
    // A call that ends for want of memory -- the cap, or the device -- gives back what the workspace held (the buffers that
    // had grown for it included), so that a smaller call starts from nothing instead of from a half-grown workspace; the
    // index of an earlier call goes with it.  The list of residues that are neither ACGT nor N stays (a few bytes).
    FragWork &W = frag_work(c);
    (void)hipStreamSynchronize(c->stream);
    W.each_buffer([&](DevBuf &b) { if (&b != &W.amb_pos && &b != &W.amb_byte) b.release(); });
    W.index_valid = false;
    W.tables_k = 0;
  }
  return status;
}

static int fragani_ex_impl(pa_ctx *c, const uint32_t *d_packed, const uint32_t *d_mask, uint64_t arena_bases,
                           const uint64_t *h_contig_start, const uint32_t *h_contig_len, const uint32_t *h_contig_genome,
                           uint32_t n_contigs, uint32_t n_genomes, uint32_t k, uint32_t frag_len, uint32_t qry0, uint32_t qry1,
                           uint32_t ref0, uint32_t ref1, uint32_t flags, uint32_t *h_total_frags, uint32_t *h_matched,
                           double *h_ident_sum) {
  PA_REQUIRE(c && d_packed && d_mask && h_total_frags && h_matched && h_ident_sum, "pa_fragani: null argument");
  PA_REQUIRE(ref0 <= ref1 && ref1 <= n_genomes, "pa_fragani: reference range [%u,%u) outside [0,%u)", ref0, ref1, n_genomes);
  PA_REQUIRE(qry0 <= qry1 && qry1 <= n_genomes, "pa_fragani: query range [%u,%u) outside [0,%u)", qry0, qry1, n_genomes);
  PA_REQUIRE(n_genomes <= 0xffffu, "pa_fragani: %u genomes (limit 65 535: a posting names its genome in 16 bits)", n_genomes);
  PA_REQUIRE((flags & ~(uint32_t)(PA_FRAGANI_REUSE_INDEX | PA_FRAGANI_COLUMNS_ONLY)) == 0, "pa_fragani: unknown flags 0x%x", flags);
  // the row length of the two result matrices on the host, and the first column they hold
  const uint32_t out_cols = (flags & PA_FRAGANI_COLUMNS_ONLY) ? ref1 - ref0 : n_genomes;
  const uint32_t out_col0 = (flags & PA_FRAGANI_COLUMNS_ONLY) ? ref0 : 0u;
  PA_REQUIRE(frag_len >= 100 && frag_len <= 0xffffu, "pa_fragani: fragLen %u outside [100, 65535]", frag_len);
  PA_HIP(hipSetDevice(c->device));
  const int w = window_size_for((int)k, (int)frag_len);
  PA_REQUIRE(w >= 1 && w <= 64, "pa_fragani: winnowing window %d outside [1,64] for k=%u fragLen=%u", w, k, frag_len);
  PA_REQUIRE((int)frag_len > w + (int)k, "pa_fragani: fragLen %u too short for window %d", frag_len, w);
  const uint32_t count_windows = frag_len - (uint32_t)(w - 1) - (k - 1);
  FragWork &W = frag_work(c);
  snprintf(W.budget.what, sizeof(W.budget.what), "pa_fragani: %u genomes (%llu residues of arena), queries [%u,%u), reference range [%u,%u), fragLen %u",
           n_genomes, (unsigned long long)arena_bases, qry0, qry1, ref0, ref1, frag_len);
  PA_TRY(stage_contigs(c, W, h_contig_start, h_contig_len, h_contig_genome, n_contigs, n_genomes, arena_bases));

  // ---- 1. minimizers of every contig
  std::optional<ProfScope> prof;  // phases timed for bench.py: index build, seeding, mapping
  const bool reuse = (flags & PA_FRAGANI_REUSE_INDEX) != 0;
  {
    // no contig holds a fragment: every pair is 0 of 0, whatever index an earlier call left (or did not leave) behind
    uint64_t any_frags = 0;
    for (uint32_t ci = 0; ci < n_contigs; ++ci) any_frags += h_contig_len[ci] / frag_len;
    if (reuse && any_frags == 0) {
      for (uint32_t g = 0; g < n_genomes; ++g) h_total_frags[g] = 0;
      for (uint64_t i = (uint64_t)qry0 * out_cols; i < (uint64_t)qry1 * out_cols; ++i) { h_matched[i] = 0; h_ident_sum[i] = 0.0; }
      return PA_OK;
    }
  }
  if (reuse) {
    PA_REQUIRE(W.index_valid && W.index_packed == (const void *)d_packed && W.index_arena_bases == arena_bases &&
                   W.index_contigs == n_contigs && W.index_genomes == n_genomes && W.index_k == k && W.index_frag_len == frag_len &&
                   W.index_ref0 <= ref0 && ref1 <= W.index_ref1,
               "pa_fragani: PA_FRAGANI_REUSE_INDEX without a preceding call on the same arena, contigs, k and fragLen whose "
               "reference range holds this one");
  }
  // ---- fragments and reference bins (host bookkeeping: some 10 ms for the 1.7 million fragments and the two result
  // matrices of 1 000 genomes -- done while minimizer_kernel runs, not before or after it with the device idle)
  std::vector<uint32_t> frag_contig, frag_no, genome_frag_off(n_genomes + 1, 0), contig_bin_off(n_contigs + 1, 0),
      genome_bin_off(n_genomes + 1, 0);
  const std::function<void()> bookkeeping = [&]() {
    for (uint32_t g = 0; g < n_genomes; ++g) h_total_frags[g] = 0;
    uint64_t all_frags = 0;
    for (uint32_t ci = 0; ci < n_contigs; ++ci) all_frags += h_contig_len[ci] / frag_len;
    frag_contig.resize(all_frags);
    frag_no.resize(all_frags);
    uint64_t at = 0;
    for (uint32_t ci = 0; ci < n_contigs; ++ci) {
      const uint32_t nf = h_contig_len[ci] / frag_len;
      for (uint32_t f = 0; f < nf; ++f) { frag_contig[at + f] = ci; frag_no[at + f] = f; }
      at += nf;
      h_total_frags[h_contig_genome[ci]] += nf;
      contig_bin_off[ci + 1] = contig_bin_off[ci] + h_contig_len[ci] / (frag_len - 20u) + 2;
    }
    for (uint32_t g = 0; g < n_genomes; ++g) genome_frag_off[g + 1] = genome_frag_off[g] + h_total_frags[g];
    {
      uint32_t ci = 0;
      for (uint32_t g = 0; g < n_genomes; ++g) {
        genome_bin_off[g] = contig_bin_off[ci];
        while (ci < n_contigs && h_contig_genome[ci] == g) ++ci;
      }
      genome_bin_off[n_genomes] = contig_bin_off[n_contigs];
    }
    memset(h_matched + (uint64_t)qry0 * out_cols, 0, (uint64_t)(qry1 - qry0) * out_cols * sizeof(uint32_t));
    for (uint64_t i = (uint64_t)qry0 * out_cols; i < (uint64_t)qry1 * out_cols; ++i) h_ident_sum[i] = 0.0;
  };
  W.index_valid = false;  // until this call has passed stage 2 (or taken it over)
  prof.emplace(c, PA_PROF_FRAG_INDEX);
  uint32_t m = reuse ? W.index_m : 0;
  if (!reuse) {
  PA_TRY(dispatch_minimizers(c, W, d_packed, d_mask, arena_bases, n_contigs, k, w, &m, &bookkeeping));
  PA_TRY(W.contig_mini_off.reserve((uint64_t)(n_contigs + 2) * 4));
  hipLaunchKernelGGL(contig_offsets_kernel, dim3(ceil_div_u64(n_contigs + 1, kThreads)), dim3(kThreads), 0, c->stream,
                     W.mini_contig.as<uint32_t>(), m, n_contigs, W.contig_mini_off.as<uint32_t>());
  }

  // ---- per-contig bucket index over window ids
  if (!reuse) {
    std::vector<uint32_t> cbo(n_contigs + 1, 0);
    for (uint32_t ci = 0; ci < n_contigs; ++ci) cbo[ci + 1] = cbo[ci] + (h_contig_len[ci] >> kBucketShift) + 2;
    PA_REQUIRE((uint64_t)cbo[n_contigs] < (1ULL << 31), "pa_fragani: bucket index too large");
    PA_TRY(upload(c, W.contig_bucket_off, cbo));
    PA_HIP(hipStreamSynchronize(c->stream));
    PA_TRY(W.bucket_first.reserve((uint64_t)cbo[n_contigs] * 4 + 16));
    hipLaunchKernelGGL(bucket_index_kernel, dim3(ceil_div_u64(cbo[n_contigs], kThreads)), dim3(kThreads), 0, c->stream,
                       W.mini_wpos.as<uint32_t>(), W.contig_mini_off.as<uint32_t>(), W.contig_bucket_off.as<uint32_t>(),
                       n_contigs, cbo[n_contigs], W.bucket_first.as<uint32_t>());
  }

  if (reuse) bookkeeping();  // (otherwise done while the minimizer kernel ran)
  const uint32_t bin_base = genome_bin_off[ref0];
  const uint64_t range_bins = std::max<uint32_t>(genome_bin_off[ref1] - bin_base, 1u);  // bins of the reference genomes asked for
  const uint32_t n_frags = (uint32_t)frag_contig.size();
  const uint32_t dict_ref0 = ref0, dict_ref1 = ref1;  // the reference genomes a dictionary built by this call holds
  auto remember_index = [&](int which_buf) {
    W.index_packed = d_packed; W.index_arena_bases = arena_bases; W.index_contigs = n_contigs; W.index_genomes = n_genomes;
    W.index_k = k; W.index_frag_len = frag_len; W.index_m = m; W.index_which = which_buf;
    if (!reuse) { W.index_ref0 = dict_ref0; W.index_ref1 = dict_ref1; }
    W.index_valid = true;
  };
  if (m == 0 || n_frags == 0) {  // nothing to map (and nothing a later call could not take over)
    prof.reset();
    PA_HIP(hipStreamSynchronize(c->stream));
    if (m == 0) remember_index(0);
    return PA_OK;
  }

  // ---- 2. dictionary of minimizer hashes: ids, postings, same-hash links
  // The dictionary holds the minimizers of the REFERENCE genomes asked for (a worker asked for one subject column sorts
  // and lists one genome's minimizers, and its seed-hit arrays are that small too); the query genomes' minimizers find
  // their hashes in it by value.  With every genome a reference the minimizers know their hash ids themselves.
  const bool restricted = reuse ? !(W.index_ref0 == 0 && W.index_ref1 == n_genomes) : !(ref0 == 0 && ref1 == n_genomes);
  uint32_t m_lo = 0, m_hi = m;
  if (!reuse && restricted) {
    uint32_t c_lo = 0, c_hi = n_contigs;  // the contigs of the reference range (contigs are listed genome by genome)
    while (c_lo < n_contigs && h_contig_genome[c_lo] < ref0) ++c_lo;
    c_hi = c_lo;
    while (c_hi < n_contigs && h_contig_genome[c_hi] < ref1) ++c_hi;
    PA_HIP(hipMemcpy(&m_lo, W.contig_mini_off.as<uint32_t>() + c_lo, 4, hipMemcpyDeviceToHost));
    PA_HIP(hipMemcpy(&m_hi, W.contig_mini_off.as<uint32_t>() + c_hi, 4, hipMemcpyDeviceToHost));
  }
  const uint32_t md = m_hi - m_lo;  // minimizers in the dictionary
  if (!reuse) {  // (an index that is taken over stays where it is: asking for room again could move -- and lose -- it)
    const uint64_t md_room = std::max<uint32_t>(md, 1u);
    // (the two key buffers of the sort are the halves of ONE allocation: once the index stands they are free, and the seed
    // hits of the batches, 8 bytes each, go there -- memory this process has touched already instead of fresh pages)
    PA_TRY(W.keys[0].reserve(2 * md_room * 8));
    for (int b = 0; b < 2; ++b) PA_TRY(W.vals[b].reserve(md_room * 4));
    PA_TRY(W.flags.reserve(md_room * 8 + 64));
    W.index_key_room = md_room;
    PA_TRY(W.mini_id.reserve((uint64_t)m * 4));
    PA_TRY(W.prev_same.reserve((uint64_t)m * 4));
    PA_TRY(W.post_cw.reserve(md_room * 8));
    PA_TRY(W.post_g.reserve(md_room * 2 + 16));
  }
  uint64_t *keys[2] = {W.keys[0].as<uint64_t>(), W.keys[0].as<uint64_t>() + W.index_key_room};
  uint32_t *vals[2] = {W.vals[0].as<uint32_t>(), W.vals[1].as<uint32_t>()};
  int which = reuse ? W.index_which : 0;
  if (!reuse) {
    PA_HIP(hipMemsetAsync(W.prev_same.p, 0xff, (uint64_t)m * 4, c->stream));  // -1: no earlier occurrence
    uint32_t n_ids = 0;
    if (md) {
      const uint32_t gm = ceil_div_u64(md, kThreads);
      hipLaunchKernelGGL(mini_keys_kernel, dim3(gm), dim3(kThreads), 0, c->stream, W.mini_hash.as<uint32_t>(), W.mini_wpos.as<uint32_t>(),
                         m_lo, md, keys[0], vals[0]);
      PA_TRY(pa_radix_sort_pairs(c, keys, vals, md, 0, 32, false, &which));
      uint32_t *d_flags = W.flags.as<uint32_t>(), *d_pos = d_flags + md;
      hipLaunchKernelGGL(key_heads_kernel, dim3(gm), dim3(kThreads), 0, c->stream, keys[which], md, d_flags);
      PA_TRY(pa_exclusive_scan_u32(c, d_flags, d_pos, md, W.scalars.as<uint64_t>()));
      PA_HIP(hipMemcpyAsync(c->h_pinned, W.scalars.p, 8, hipMemcpyDeviceToHost, c->stream));
      PA_HIP(hipStreamSynchronize(c->stream));
      n_ids = (uint32_t)c->h_pinned[0];
      PA_REQUIRE(n_ids < (1u << 31), "pa_fragani: %u distinct minimizer hashes (limit 2^31)", n_ids);
      PA_TRY(W.post_start.reserve((uint64_t)(n_ids + 2) * 4));
      PA_TRY(W.uniq_hash.reserve((uint64_t)(n_ids + 2) * 4));
      // (the contig of every block of minimizers: the look-back words of minimizer_kernel are free again)
      const uint32_t n_blocks = (uint32_t)(((uint64_t)m + (1u << kContigBlockShift) - 1u) >> kContigBlockShift);
      PA_TRY(W.block_counts.reserve((uint64_t)n_blocks * 4 + 16));
      hipLaunchKernelGGL(block_contig_kernel, dim3(ceil_div_u64(n_blocks, kThreads)), dim3(kThreads), 0, c->stream, W.mini_contig.as<uint32_t>(), m,
                         W.block_counts.as<uint32_t>());
      hipLaunchKernelGGL(postings_kernel, dim3(gm), dim3(kThreads), 0, c->stream, keys[which], vals[which], d_flags, d_pos, md,
                         n_ids, W.mini_contig.as<uint32_t>(), W.mini_id.as<uint32_t>(), W.post_start.as<uint32_t>(),
                         W.prev_same.as<int32_t>(), W.mini_wpos.as<uint32_t>(), W.contig_genome.as<uint32_t>(),
                         W.post_cw.as<uint64_t>(), W.post_g.as<uint16_t>(), W.contig_mini_off.as<uint32_t>(), n_contigs,
                         W.uniq_hash.as<uint32_t>(), W.block_counts.as<uint32_t>());
      PA_TRY(cut_frequent_postings(c, W, d_flags, d_pos, vals[which], vals[1 - which], reinterpret_cast<uint32_t *>(keys[1 - which]), md, n_ids,
                                   h_contig_genome, n_contigs, n_genomes));
    } else {  // the reference genomes hold no minimizer: an empty dictionary
      PA_TRY(W.post_start.reserve(16));
      PA_TRY(W.uniq_hash.reserve(16));
      PA_TRY(W.hash_cut.reserve(16));
      PA_HIP(hipMemsetAsync(W.post_start.p, 0, 8, c->stream));
      PA_HIP(hipMemsetAsync(W.hash_cut.p, 0, 8, c->stream));
    }
    W.index_ids = n_ids;
    if (restricted) {
      uint32_t bits = 10;
      while ((1ull << bits) < 2ull * n_ids + 1) ++bits;
      W.index_lookup_bits = bits;
      PA_TRY(W.lookup_at.reserve((16ull << bits) + 16));
      PA_HIP(hipMemsetAsync(W.lookup_at.p, 0xff, 16ull << bits, c->stream));
      if (n_ids)
        hipLaunchKernelGGL(lookup_insert_kernel, dim3(ceil_div_u64(n_ids, kThreads)), dim3(kThreads), 0, c->stream, W.uniq_hash.as<uint32_t>(),
                           W.post_start.as<uint32_t>(), W.hash_cut.as<uint32_t>(), n_ids, bits, W.lookup_at.as<uint4>());
    }
  }
  const uint32_t *d_sorted_idx = vals[which];
  // the index (minimizers, bucket index, dictionary, postings) is complete: a later call may take it over
  remember_index(which);

  // ---- tables indexed by sketch size
  {
    // Mashmap's statistics per sketch size and the identity of every (shared, s): functions of k alone, ~10 ms of binomial
    // tails and logarithms on the host -- made once per context and k (the device copies stay where they are), not in
    // every call with the device waiting
    if (W.tables_k != k) {
      std::vector<uint32_t> mh(kQMax + 1), ms(kQMax + 1);
      PA_TRY(pa_fragani_tables(k, kQMax, mh.data(), ms.data()));
      std::vector<float> ident((uint64_t)(kQMax + 1) * (kQMax + 1), 0.0f);
      for (uint32_t s = 1; s <= (uint32_t)kQMax; ++s)
        for (uint32_t x = 0; x <= s; ++x) ident[(uint64_t)s * (kQMax + 1) + x] = (float)pa_fragani_identity(x, s, k);
      W.tables_k = 0;
      PA_TRY(upload(c, W.tab_min_hits, mh));
      PA_TRY(upload(c, W.tab_min_shared, ms));
      PA_TRY(upload(c, W.ident_tab, ident));
      PA_HIP(hipStreamSynchronize(c->stream));  // (the vectors go out of scope)
      W.tables_k = k;
    }
    // The table of best fragments per reference bin holds the bins of the reference range only (a worker asked for one
    // subject column keeps 1 700 bins per query genome instead of 1.7 million): bin numbers relative to the range's first;
    // genomes outside it get an empty run of bins, so the reduction gives them nothing, as an all-zero table did.
    std::vector<uint32_t> cbo_rel(contig_bin_off.size()), gbo_rel(genome_bin_off.size());
    for (size_t i = 0; i < contig_bin_off.size(); ++i) cbo_rel[i] = contig_bin_off[i] - std::min(contig_bin_off[i], bin_base);
    for (size_t g = 0; g < genome_bin_off.size(); ++g)
      gbo_rel[g] = std::min<uint32_t>(genome_bin_off[g] - std::min(genome_bin_off[g], bin_base), (uint32_t)range_bins);
    PA_TRY(upload(c, W.contig_bin_off, cbo_rel));
    PA_TRY(upload(c, W.genome_bin_off, gbo_rel));
    PA_HIP(hipStreamSynchronize(c->stream));
  }

  // ---- batches of query genomes
  // Query genomes go through in batches of up to 2^17 fragments (13 batches for 1 000 genomes of 5 Mb: 1.46 s against
  // 1.48 s with 2^16).  A batch whose seed hits do not fit 31-bit indices is halved and started again.
#ifndef PA_FRAGANI_BATCH_FRAGS
#define PA_FRAGANI_BATCH_FRAGS (1u << 17)
#endif
  uint32_t batch_frags = PA_FRAGANI_BATCH_FRAGS;
  uint64_t hit_limit = 1ULL << 31;
  if (const char *v = PA_TOOL_ENV("PA_FRAGANI_BATCH_HITS")) hit_limit = std::max<uint64_t>(1, strtoull(v, nullptr, 10));  // tests: force the halving
  const uint64_t kMaxTableBytes = 1ULL << 31;
  PA_TRY(W.scalars.reserve(64));
  uint32_t *d_overflow = W.scalars.as<uint32_t>() + 8;
  uint32_t *d_max_hits = W.scalars.as<uint32_t>() + 12;  // most seed hits of one fragment in the batch
  // [0] segments of <= kHitCapSmall hits, [1] of > kHitCap hits, [2] the longest of those, [3] of > kHitCapSmall hits;
  // [6] cursor of big_segments_kernel
  uint32_t *d_seg_counters = W.scalars.as<uint32_t>() + 4;
  // The bucketed pipeline needs an LDS counter per reference genome for each of a workgroup's waves and
  // 16-bit fields for query window ids and for contigs within a genome; otherwise the sorted pipeline runs.
  std::vector<uint32_t> gfc(n_genomes + 1, n_contigs);
  uint32_t most_contigs = 0;
  {
    for (uint32_t ci = n_contigs; ci-- > 0;) gfc[h_contig_genome[ci]] = ci;
    for (uint32_t g = n_genomes; g-- > 0;) if (gfc[g] == n_contigs) gfc[g] = gfc[g + 1];  // genome without contigs
    for (uint32_t g = 0; g < n_genomes; ++g) most_contigs = std::max(most_contigs, gfc[g + 1] - gfc[g]);
    bool genome_major = true;
    for (uint32_t ci = 1; ci < n_contigs; ++ci) genome_major = genome_major && h_contig_genome[ci] >= h_contig_genome[ci - 1];
    PA_REQUIRE(genome_major, "pa_fragani: contigs must be listed genome by genome");
  }
  PA_TRY(upload(c, W.genome_first_contig, gfc));
  PA_HIP(hipStreamSynchronize(c->stream));
  const bool trace = PA_TOOL_ENV("PA_FRAGANI_TRACE") != nullptr;  // per-batch sizes on stderr
  const bool force_sorted = [] {  // PA_FRAGANI_HITS=sorted: the path of more than 8 192 genomes, for any number (tests)
    const char *v = PA_TOOL_ENV("PA_FRAGANI_HITS");
    return v && v[0] == 's';
  }();
  const bool fields_fit = count_windows <= 0xffffu && most_contigs <= 0xffffu;
  PA_REQUIRE(fields_fit, "pa_fragani: fragment length %u or %u contigs in one genome exceed the 16-bit fields of the "
                         "mapping kernel", frag_len, most_contigs);
  const bool use_buckets = !force_sorted && (uint64_t)kBucketWaves * n_genomes * 4u <= 128u * 1024u;
  PA_HIP(hipMemsetAsync(d_overflow, 0, 8, c->stream));
  prof.reset();
  for (uint32_t g0 = qry0; g0 < qry1;) {
    uint32_t g1 = g0 + 1;
    while (g1 < qry1 && genome_frag_off[g1 + 1] - genome_frag_off[g0] <= batch_frags &&
           (uint64_t)(g1 + 1 - g0) * range_bins * 8 <= kMaxTableBytes)
      ++g1;
    const uint32_t f0 = genome_frag_off[g0], nf = genome_frag_off[g1] - f0, nq = g1 - g0;
    PA_REQUIRE(nf < (1u << 20), "pa_fragani: genome %u alone has %u fragments (limit 2^20)", g0, nf);
    if (nf == 0) { g0 = g1; continue; }
    std::vector<uint32_t> fc(frag_contig.begin() + f0, frag_contig.begin() + f0 + nf),
        fn(frag_no.begin() + f0, frag_no.begin() + f0 + nf), fg(nf);
    for (uint32_t i = 0; i < nf; ++i) fg[i] = h_contig_genome[fc[i]] - g0;
    PA_TRY(upload(c, W.frag_contig, fc));
    PA_TRY(upload(c, W.frag_no, fn));
    PA_TRY(upload(c, W.frag_genome_local, fg));
    PA_HIP(hipStreamSynchronize(c->stream));
    // The batch's working arrays go into memory the index build has left behind wherever they fit (a fresh process pays
    // ~65 ms per GB for the FIRST use of device memory -- more than the kernels of a whole 1 000-genome run for the 6 GB
    // these are): the fragments' sketches into the sort's spare value buffer, the seed hits into its key buffers, the
    // table of best fragments into the flag / scan scratch.  (All three are dead once the index stands, also for a call
    // that takes the index over.)
    uint32_t *q_hash_p, *q_pos_p, *q_id_p;
    {
      const uint64_t per = (uint64_t)nf * kQMax;
      DevBuf &spare = W.vals[1 - which];
      if (spare.bytes >= 3 * per * 4) {
        q_hash_p = spare.as<uint32_t>(); q_pos_p = q_hash_p + per; q_id_p = q_pos_p + per;
      } else {
        PA_TRY(W.q_hash.reserve(per * 4));
        PA_TRY(W.q_pos.reserve(per * 4));
        PA_TRY(W.q_id.reserve(per * 4));
        q_hash_p = W.q_hash.as<uint32_t>(); q_pos_p = W.q_pos.as<uint32_t>(); q_id_p = W.q_id.as<uint32_t>();
      }
    }
    PA_TRY(W.q_s.reserve((uint64_t)nf * 4));
    PA_TRY(W.q_tab.reserve((uint64_t)nf * kQtBuckets * 2));  // the sketches' bucket tables
    PA_TRY(W.q_cut.reserve((uint64_t)nf * 4));
    PA_TRY(W.hit_count.reserve((uint64_t)nf * 4));
    PA_TRY(W.hit_off.reserve((uint64_t)nf * 4));
    const uint32_t gw = ceil_div_u64(nf, kThreads / 64);
    PA_HIP(hipMemsetAsync(d_max_hits, 0, 8, c->stream));  // [0] most hits, [1] longest sketch of a fragment
    prof.emplace(c, PA_PROF_FRAG_SEED);
    PA_TRY(W.frag_d.reserve((uint64_t)nf * 4));
    hipLaunchKernelGGL(windows_without_selection_kernel, dim3(ceil_div_u64(nf, kThreads)), dim3(kThreads), 0, c->stream, d_packed, d_mask,
                       arena_bases, W.contig_start.as<uint64_t>(), k, (uint32_t)w, W.frag_contig.as<uint32_t>(), W.frag_no.as<uint32_t>(),
                       nf, frag_len, count_windows, W.frag_d.as<uint32_t>(), W.ambiguous(d_packed));
    hipLaunchKernelGGL(query_sketch_kernel, dim3(gw), dim3(kThreads), 0, c->stream, W.frag_d.as<uint32_t>(), W.frag_contig.as<uint32_t>(),
                       W.frag_no.as<uint32_t>(), nf, frag_len, count_windows, W.contig_mini_off.as<uint32_t>(),
                       W.contig_bucket_off.as<uint32_t>(), W.bucket_first.as<uint32_t>(), W.mini_hash.as<uint32_t>(), W.mini_wpos.as<uint32_t>(), W.mini_id.as<uint32_t>(),
                       W.post_start.as<uint32_t>(), q_hash_p, q_pos_p,
                       q_id_p, W.q_s.as<uint32_t>(), W.hit_count.as<uint32_t>(), d_overflow, d_max_hits,
                       W.q_cut.as<uint32_t>(), restricted ? W.lookup_at.as<uint4>() : nullptr, W.index_lookup_bits, W.q_tab.as<uint32_t>());
    PA_TRY(pa_exclusive_scan_u32(c, W.hit_count.as<uint32_t>(), W.hit_off.as<uint32_t>(), nf, W.scalars.as<uint64_t>()));
    PA_HIP(hipMemcpyAsync(c->h_pinned, W.scalars.p, 8, hipMemcpyDeviceToHost, c->stream));
    PA_HIP(hipMemcpyAsync(c->h_pinned + 1, d_max_hits, 8, hipMemcpyDeviceToHost, c->stream));
    PA_HIP(hipStreamSynchronize(c->stream));
    const uint64_t n_hits = c->h_pinned[0];
    const uint32_t max_hits = (uint32_t)c->h_pinned[1];
    const uint32_t s_cap = std::min<uint32_t>(kQMax, (((uint32_t)(c->h_pinned[1] >> 32) + 63u) / 64u) * 64u);
    if (n_hits >= hit_limit && nq > 1) {  // closely related or repetitive genomes: fewer query genomes per batch
      batch_frags = std::max<uint32_t>(1u, std::min(batch_frags, nf) / 2u);
      prof.reset();
      continue;  // the same g0 again
    }
    if (trace) fprintf(stderr, "pa_fragani: batch of genomes %u..%u: %u fragments, %llu seed hits\n", g0, g1, nf, (unsigned long long)n_hits);
    PA_REQUIRE(n_hits < (1ULL << 31), "pa_fragani: %llu seed hits for the fragments of genome %u alone (limit 2^31); highly "
               "repetitive input", (unsigned long long)n_hits, g0);
    unsigned long long *table_p;
    if (use_buckets && W.flags.bytes >= (uint64_t)nq * range_bins * 8) {
      table_p = W.flags.as<unsigned long long>();  // (the sorted path keeps its head flags there)
    } else {
      PA_TRY(W.table.reserve((uint64_t)nq * range_bins * 8));
      table_p = W.table.as<unsigned long long>();
    }
    PA_HIP(hipMemsetAsync(table_p, 0, (uint64_t)nq * range_bins * 8, c->stream));
    if (n_hits) {
      const bool hits_in_sort_keys = W.keys[0].bytes >= n_hits * 8;
      if (!hits_in_sort_keys) PA_TRY(W.hkeys[0].reserve(n_hits * 8));
      PA_TRY(W.hvals[0].reserve(n_hits * 4));  // (written by the paths that order the hits as a whole only)
      uint64_t *hk[2] = {hits_in_sort_keys ? W.keys[0].as<uint64_t>() : W.hkeys[0].as<uint64_t>(), nullptr};
      uint32_t *hv[2] = {W.hvals[0].as<uint32_t>(), nullptr};
      auto second_buffers = [&]() -> int {  // only the radix sort needs the ping-pong copies
        PA_TRY(W.hkeys[1].reserve(n_hits * 8));
        PA_TRY(W.hvals[1].reserve(n_hits * 4));
        hk[1] = W.hkeys[1].as<uint64_t>();
        hv[1] = W.hvals[1].as<uint32_t>();
        return PA_OK;
      };
      PA_TRY(W.run_g.reserve(256));  // the mapping kernel's event counters (-DPA_MAP_STATS)
      int bits = 44;
      for (uint32_t x = nf; x > 1; x >>= 1) ++bits;
      bits = (bits + 1 + 7) & ~7;
      if (bits > 64) bits = 64;
      int hw = 0;
      uint32_t n_keep = 0, n_large = 0, large_at = 0;  // bucketed path: long segments sit at the end of the lists
      bool presorted = true;
      if (use_buckets) {
        // hits bucketed by (fragment, reference genome); segments listed by the same kernel
        const uint64_t seg_cap64 = std::min<uint64_t>(n_hits, (uint64_t)nf * n_genomes);
        const uint32_t seg_cap = (uint32_t)std::min<uint64_t>(seg_cap64, 0xfffffff0ull);
        PA_TRY(W.seg_a0.reserve((uint64_t)seg_cap * 4 + 16));
        PA_TRY(W.seg_nh.reserve((uint64_t)seg_cap * 4 + 16));
        PA_TRY(W.seg_f.reserve((uint64_t)seg_cap * 4 + 16));
        unsigned long long *d_cursor64 = reinterpret_cast<unsigned long long *>(W.scalars.as<uint32_t>() + 14);  // [lo] short, [hi] long segments
        const uint32_t lds_bytes = (uint32_t)kBucketWaves * n_genomes * 4u;
        PA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(bucket_hits_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        uint32_t n_big = 0, max_big = 0;
        // one fragment per workgroup with the listed pairs' hits staged in LDS (whole-sector writes) when the two per-genome
        // arrays leave room for a staging area; the batch that is about to be ordered as a whole (write_all) keeps the
        // wave-per-fragment form, which writes every slot
        // (four workgroups of eight waves per CU: 40 KB of LDS each)
        const uint32_t lds_room = 40u * 1024u;
        const uint32_t stage_cap = n_genomes * 8u + 8192u <= lds_room ? ((lds_room - n_genomes * 8u) / 8u) & ~63u : 0u;
        const uint32_t stage_lds = stage_cap * 8u + n_genomes * 8u;
        const bool stage_hits = stage_cap >= 1024u;
        if (stage_hits)
          PA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(bucket_hits_staged_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)stage_lds));
        auto bucket_pass = [&](bool write_all) -> int {
          PA_HIP(hipMemsetAsync(d_seg_counters, 0, 16, c->stream));
          PA_HIP(hipMemsetAsync(d_cursor64, 0, 8, c->stream));
          if (stage_hits && !write_all)
            hipLaunchKernelGGL(bucket_hits_staged_kernel, dim3(nf), dim3(kStageWaves * 64), stage_lds, c->stream, nf, q_pos_p, q_id_p,
                               W.q_s.as<uint32_t>(), W.hit_off.as<uint32_t>(), W.post_g.as<uint16_t>(),
                               W.post_cw.as<uint64_t>(), n_genomes, W.tab_min_hits.as<uint32_t>(), hk[0], W.seg_a0.as<uint32_t>(),
                               W.seg_nh.as<uint32_t>(), W.seg_f.as<uint32_t>(), seg_cap, d_seg_counters, d_cursor64, ref0, ref1, stage_cap);
          else
          hipLaunchKernelGGL(bucket_hits_kernel, dim3(ceil_div_u64(nf, kBucketWaves)), dim3(kBucketWaves * 64), lds_bytes,
                             c->stream, nf, q_pos_p, q_id_p, W.q_s.as<uint32_t>(),
                             W.hit_off.as<uint32_t>(), W.post_g.as<uint16_t>(), W.post_cw.as<uint64_t>(), n_genomes,
                             W.tab_min_hits.as<uint32_t>(), hk[0], hv[0],
                             W.seg_a0.as<uint32_t>(), W.seg_nh.as<uint32_t>(), W.seg_f.as<uint32_t>(), seg_cap, d_seg_counters,
                             d_cursor64, ref0, ref1, write_all);
          PA_HIP(hipMemcpyAsync(c->h_pinned, d_seg_counters, 16, hipMemcpyDeviceToHost, c->stream));
          PA_HIP(hipMemcpyAsync(c->h_pinned + 2, d_cursor64, 8, hipMemcpyDeviceToHost, c->stream));
          PA_HIP(hipStreamSynchronize(c->stream));
          const uint32_t *hc32 = reinterpret_cast<const uint32_t *>(c->h_pinned);
          n_keep = (uint32_t)c->h_pinned[2];
          n_large = (uint32_t)(c->h_pinned[2] >> 32);
          large_at = seg_cap - n_large;
          n_big = hc32[1];
          max_big = hc32[2];
          PA_REQUIRE((uint64_t)n_keep + n_large <= seg_cap, "pa_fragani: %u + %u segments exceed the list capacity %u",
                     n_keep, n_large, seg_cap);
          return PA_OK;
        };
        PA_TRY(bucket_pass(false));
        presorted = false;
        uint32_t frag_sort_max = kFragSortMax;  // tests: PA_FRAGANI_SORT_MAX=600 sends a 60-copy repeat family down this path
        if (const char *v = PA_TOOL_ENV("PA_FRAGANI_SORT_MAX")) frag_sort_max = (uint32_t)std::max(1, atoi(v));
        if (n_big && max_big > frag_sort_max) {
          // a repeat family with more hits than one LDS sort takes: order the whole batch by key; the
          // (fragment, genome) slices keep their places because contigs are numbered genome by genome -- provided every
          // slot holds its own key, so the bucketing runs again and this time also writes the hits of the pairs that are
          // not listed (they are skipped otherwise: unwritten slots would be sorted into other fragments' ranges)
          PA_TRY(bucket_pass(true));
          PA_TRY(second_buffers());
          PA_TRY(pa_radix_sort_pairs(c, hk, hv, n_hits, 0, bits, false, &hw));
          presorted = true;
        } else if (n_big) {
          PA_TRY(W.seg_list.reserve((uint64_t)n_big * 8 + 16));
          uint32_t *big_a0 = W.seg_list.as<uint32_t>(), *big_nh = big_a0 + n_big;
          PA_HIP(hipMemsetAsync(d_seg_counters + 6, 0, 4, c->stream));
          hipLaunchKernelGGL(big_segments_kernel, dim3(ceil_div_u64(n_large, kThreads)), dim3(kThreads), 0, c->stream,
                             W.seg_a0.as<uint32_t>() + large_at, W.seg_nh.as<uint32_t>() + large_at, n_large, big_a0,
                             big_nh, d_seg_counters + 6);
          uint32_t np2_max = 2;
          while (np2_max < max_big) np2_max <<= 1;
          const uint32_t sort_lds = np2_max * 12u;
          PA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(frag_sort_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)sort_lds));
          hipLaunchKernelGGL(frag_sort_kernel, dim3(n_big), dim3(kFragSortThreads), sort_lds, c->stream, hk[0], hv[0],
                             big_a0, big_nh, np2_max, 64u - kHitRankShift + 11u);  // (contig, window id) to the top, the rank below
        }
      } else {
        // general path: all hits sorted by key, segments from head flags
        hipLaunchKernelGGL(fill_hits_kernel, dim3(gw), dim3(kThreads), 0, c->stream, nf, q_pos_p,
                           q_id_p, W.q_s.as<uint32_t>(), W.hit_off.as<uint32_t>(),
                           W.post_start.as<uint32_t>(), d_sorted_idx, W.mini_wpos.as<uint32_t>(),
                           W.mini_contig.as<uint32_t>(), hk[0], hv[0]);
        if (max_hits <= kFragSortMax) {  // every fragment's hits fit one LDS sort
          uint32_t np2_max = 2;
          while (np2_max < max_hits) np2_max <<= 1;
          const uint32_t lds_bytes = np2_max * 12u;
          PA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(frag_sort_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
          hipLaunchKernelGGL(frag_sort_kernel, dim3(nf), dim3(kFragSortThreads), lds_bytes, c->stream, hk[0], hv[0],
                             W.hit_off.as<uint32_t>(), W.hit_count.as<uint32_t>(), np2_max, 0u);
        } else {
          PA_TRY(second_buffers());
          PA_TRY(pa_radix_sort_pairs(c, hk, hv, n_hits, 0, bits, false, &hw));
        }
        PA_TRY(W.flags.reserve(n_hits * 8 + 64));
        uint32_t *hf = W.flags.as<uint32_t>(), *hp = hf + n_hits;
        const uint32_t gh = ceil_div_u64(n_hits, kThreads);
        hipLaunchKernelGGL(segment_heads_kernel, dim3(gh), dim3(kThreads), 0, c->stream, hk[hw], (uint32_t)n_hits,
                           W.contig_genome.as<uint32_t>(), hf);
        PA_TRY(pa_exclusive_scan_u32(c, hf, hp, n_hits, W.scalars.as<uint64_t>()));
        PA_HIP(hipMemcpyAsync(c->h_pinned, W.scalars.p, 8, hipMemcpyDeviceToHost, c->stream));
        PA_HIP(hipStreamSynchronize(c->stream));
        const uint32_t n_segs = (uint32_t)c->h_pinned[0];
        PA_TRY(W.seg_start.reserve((uint64_t)(n_segs + 2) * 4));
        hipLaunchKernelGGL(segment_starts_kernel, dim3(gh), dim3(kThreads), 0, c->stream, hf, hp, (uint32_t)n_hits,
                           W.seg_start.as<uint32_t>());
        // most segments are chance hits of unrelated genomes (fewer hits than any L1 run needs): drop them
        // here, one thread each, instead of spending a workgroup launch on each in the mapping kernel
        const uint32_t gs = ceil_div_u64(n_segs, kThreads);
        hipLaunchKernelGGL(segment_keep_kernel, dim3(gs), dim3(kThreads), 0, c->stream, hk[hw],
                           W.seg_start.as<uint32_t>(), n_segs, W.q_s.as<uint32_t>(), W.tab_min_hits.as<uint32_t>(),
                           W.contig_genome.as<uint32_t>(), ref0, ref1, hf);
        PA_TRY(pa_exclusive_scan_u32(c, hf, hp, n_segs, W.scalars.as<uint64_t>()));
        PA_HIP(hipMemcpyAsync(c->h_pinned, W.scalars.p, 8, hipMemcpyDeviceToHost, c->stream));
        PA_HIP(hipStreamSynchronize(c->stream));
        n_keep = (uint32_t)c->h_pinned[0];
        PA_TRY(W.seg_list.reserve((uint64_t)(n_keep + 1) * 4));
        PA_TRY(W.seg_a0.reserve((uint64_t)n_keep * 4 + 16));
        PA_TRY(W.seg_nh.reserve((uint64_t)n_keep * 4 + 16));
        hipLaunchKernelGGL(segment_list_kernel, dim3(gs), dim3(kThreads), 0, c->stream, hf, hp, n_segs,
                           W.seg_list.as<uint32_t>());
        if (n_keep)
          hipLaunchKernelGGL(segments_from_list_kernel, dim3(ceil_div_u64(n_keep, kThreads)), dim3(kThreads), 0, c->stream,
                             W.seg_start.as<uint32_t>(), W.seg_list.as<uint32_t>(), n_keep, W.seg_a0.as<uint32_t>(),
                             W.seg_nh.as<uint32_t>());
      }
      prof.reset();
      prof.emplace(c, PA_PROF_FRAG_MAP);
      // stretch capacity: the expected minimizers of one window (density 2 / (w + 1)) plus a third, in steps of 64
      const uint32_t per_window = (uint32_t)(2.0 * count_windows / (w + 1.0));
      const uint32_t ref_cap = std::min<uint32_t>(kRefCapMax, std::max<uint32_t>(256u, (per_window * 4u / 3u + 63u) / 64u * 64u));
#ifdef PA_MAP_STATS
      PA_HIP(hipMemsetAsync(W.run_g.p, 0, 256, c->stream));
#endif
#ifdef PA_TOOLS
      const char *cut_env = PA_TOOL_ENV("PA_MAP_CUT");  // tools: the mapping kernel cut short after a phase (timing by difference)
      const uint32_t map_cut = cut_env ? (uint32_t)atoi(cut_env) : 0xffffffffu;
#endif
      auto launch_map = [&](const uint32_t *list_a0, const uint32_t *list_nh, const uint32_t *list_f, uint32_t count, uint32_t hit_cap,
                            auto all_staged) -> int {
        if (count == 0) return PA_OK;
        constexpr bool kAll = decltype(all_staged)::value;
        PA_TRY(W.seg_rec.reserve((uint64_t)count * 48));
        hipLaunchKernelGGL(segment_records_kernel, dim3(ceil_div_u64(count, kThreads)), dim3(kThreads), 0, c->stream, hk[hw],
                           list_a0, list_nh, list_f, count, W.q_s.as<uint32_t>(), W.q_cut.as<uint32_t>(), W.tab_min_hits.as<uint32_t>(),
                           W.tab_min_shared.as<uint32_t>(), W.contig_genome.as<uint32_t>(), W.genome_first_contig.as<uint32_t>(),
                           W.contig_mini_off.as<uint32_t>(), W.contig_bucket_off.as<uint32_t>(), W.frag_genome_local.as<uint32_t>(), W.seg_rec.as<uint4>());
#ifndef PA_MAP_LDS_PAD
#define PA_MAP_LDS_PAD 0u  // (an experiment's switch: LDS asked for and not used, to see what a wave less per SIMD costs)
#endif
#define PA_MAP_CASE(CAP)                                                                                                  \
  case CAP:                                                                                                               \
    hipLaunchKernelGGL((map_segments_kernel<CAP, kAll>), dim3(count), dim3(64), eval_lds_bytes(s_cap, hit_cap, CAP) + PA_MAP_LDS_PAD, c->stream,  \
                       hk[hw], hv[hw], W.seg_rec.as<uint4>(), count, presorted, q_hash_p, W.q_tab.as<uint32_t>(),          \
                       W.frag_genome_local.as<uint32_t>(), frag_len, count_windows,                                        \
                       W.tab_min_shared.as<uint32_t>(), W.contig_mini_off.as<uint32_t>(),                                  \
                       W.contig_bucket_off.as<uint32_t>(), W.bucket_first.as<uint32_t>(), W.mini_hash.as<uint32_t>(),      \
                       W.mini_wpos.as<uint32_t>(), W.prev_same.as<int32_t>(), W.contig_bin_off.as<uint32_t>(), range_bins, \
                       table_p, W.run_g.as<uint32_t>(), s_cap, hit_cap PA_MAP_CUT_ARG);                                   \
    break;
        switch (ref_cap) {
          PA_MAP_CASE(256) PA_MAP_CASE(320) PA_MAP_CASE(384) PA_MAP_CASE(448)
          default: PA_MAP_CASE(512)
        }
#undef PA_MAP_CASE
        return PA_OK;
      };
      if (use_buckets) {
        if (n_keep) {
          PA_TRY(W.seg2_a0.reserve((uint64_t)n_keep * 4 + 16));
          PA_TRY(W.seg2_nh.reserve((uint64_t)n_keep * 4 + 16));
          PA_TRY(W.seg2_f.reserve((uint64_t)n_keep * 4 + 16));
          // segments of at most kSparseHits hits go to map_sparse_kernel (listed from the back of the same arrays), unless
          // the hits carry their ranks in the sort's payload (a batch ordered as a whole)
          uint32_t sparse_max = presorted ? 0u : kSparseHits;
          if (const char *v = PA_TOOL_ENV("PA_FRAGANI_SPARSE")) sparse_max = (atoi(v) && !presorted) ? kSparseHits : 0u;  // tests: 0 = every segment through the general kernel (the switch can only take the sparse kernel away: its key layout is the bucketed one)
          unsigned long long *d_pre_cursor = reinterpret_cast<unsigned long long *>(d_seg_counters + 6);  // [6] general, [7] sparse (8-byte aligned)
          PA_HIP(hipMemsetAsync(d_pre_cursor, 0, 8, c->stream));
          hipLaunchKernelGGL(prefilter_segments_kernel, dim3(ceil_div_u64(n_keep, kThreads)), dim3(kThreads), 0, c->stream,
                             hk[hw], W.seg_a0.as<uint32_t>(), W.seg_nh.as<uint32_t>(), W.seg_f.as<uint32_t>(), n_keep,
                             W.q_s.as<uint32_t>(), W.tab_min_hits.as<uint32_t>(), W.q_cut.as<uint32_t>(), frag_len, W.seg2_a0.as<uint32_t>(),
                             W.seg2_nh.as<uint32_t>(), W.seg2_f.as<uint32_t>(), d_pre_cursor, sparse_max);
          PA_HIP(hipMemcpyAsync(c->h_pinned, d_pre_cursor, 8, hipMemcpyDeviceToHost, c->stream));
          PA_HIP(hipStreamSynchronize(c->stream));
          const uint32_t n_small = (uint32_t)c->h_pinned[0], n_sparse = (uint32_t)(c->h_pinned[0] >> 32);
          if (trace)
            fprintf(stderr, "pa_fragani: genomes %u..%u: %u fragments, %llu seed hits, %u + %u listed segments, %u left "
                            "after the tiny-segment filter\n", g0, g1, nf, (unsigned long long)n_hits, n_keep, n_large, n_small);
          PA_TRY(launch_map(W.seg2_a0.as<uint32_t>(), W.seg2_nh.as<uint32_t>(), W.seg2_f.as<uint32_t>(), n_small, (uint32_t)kHitCapSmall,
                            std::true_type{}));
          if (n_sparse) {
            const uint32_t at = n_keep - n_sparse;  // the sparse list sits at the back of the same arrays
            PA_TRY(W.seg_over.reserve((uint64_t)n_sparse * 12 + 16));
            uint32_t *over_a0 = W.seg_over.as<uint32_t>(), *over_nh = over_a0 + n_sparse, *over_f = over_nh + n_sparse;
            uint32_t *d_over_n = W.scalars.as<uint32_t>() + 3;
            PA_HIP(hipMemsetAsync(d_over_n, 0, 4, c->stream));
            hipLaunchKernelGGL(map_sparse_kernel, dim3(n_sparse), dim3(64), 0, c->stream, hk[hw], W.seg2_a0.as<uint32_t>() + at,
                               W.seg2_nh.as<uint32_t>() + at, W.seg2_f.as<uint32_t>() + at, n_sparse, W.q_s.as<uint32_t>(), q_hash_p,
                               W.frag_genome_local.as<uint32_t>(), frag_len, count_windows, W.tab_min_hits.as<uint32_t>(),
                               W.tab_min_shared.as<uint32_t>(), W.contig_mini_off.as<uint32_t>(), W.contig_bucket_off.as<uint32_t>(),
                               W.bucket_first.as<uint32_t>(), W.mini_hash.as<uint32_t>(), W.mini_wpos.as<uint32_t>(),
                               W.prev_same.as<int32_t>(), W.contig_bin_off.as<uint32_t>(), range_bins, table_p, over_a0, over_nh, over_f,
                               d_over_n
#ifdef PA_MAP_STATS
                               , W.run_g.as<uint32_t>()
#endif
            );
            PA_HIP(hipMemcpyAsync(c->h_pinned, d_over_n, 4, hipMemcpyDeviceToHost, c->stream));
            PA_HIP(hipStreamSynchronize(c->stream));
            const uint32_t n_over = *reinterpret_cast<const uint32_t *>(c->h_pinned);
            if (trace) fprintf(stderr, "pa_fragani: %u segments of at most %u hits in the sparse kernel, %u of them handed on\n", n_sparse, kSparseHits, n_over);
            // what does not fit the simple form (a hash twice in a stretch, over-long windows or ranges) goes through the general kernel
            PA_TRY(launch_map(over_a0, over_nh, over_f, n_over, (uint32_t)kHitCapSmall, std::true_type{}));
          }
        }
        PA_TRY(launch_map(W.seg_a0.as<uint32_t>() + large_at, W.seg_nh.as<uint32_t>() + large_at, W.seg_f.as<uint32_t>() + large_at, n_large,
                          (uint32_t)kHitCap, std::false_type{}));
      } else {
        PA_TRY(launch_map(W.seg_a0.as<uint32_t>(), W.seg_nh.as<uint32_t>(), nullptr, n_keep, (uint32_t)kHitCap, std::false_type{}));
      }
#ifdef PA_MAP_STATS
      if (trace) {
        uint32_t st[64];
        PA_HIP(hipMemcpy(st, W.run_g.p, 256, hipMemcpyDeviceToHost));
        fprintf(stderr, "pa_fragani: map stats: %u segments at L1 with %u hits, %u candidates, %u groups, %u begins past the bound, "
                        "%u rounds, %u stretch entries, %u windows evaluated, %u fine passes, %u cooperative, %u begins in rounds, "
                        "%u begins finished, %u rounds without items, %u second passes, %u windows with an exact value, %u of them at or above the bar, "
                        "%u begins dropped by the tight bound, %u rounds ended by it; rounds by seed hits of the segment (<= 7, 8-15, 16-31, 32-63, "
                        "64-127, 128-255, more): %u %u %u %u %u %u %u, segments: %u %u %u %u %u %u %u; work model: %u minimizers in the candidates' ranges, "
                        "%u states tying their candidate's optimum; past the tight bound: %u rounds, %u begins, %u windows, %u of them at or above the bar; %u begins dropped by the bound asked at a round's end; %u segments that are one run of hits (no L1 scan) with %u hits; hits within 4096 window ids on one contig (ordered by counting): %u segments, all but one hit: %u, all but two: %u\n",
                st[0], st[1], st[2], st[3], st[4], st[5], st[6], st[7], st[8], st[9], st[10], st[11], st[12], st[13], st[14], st[15], st[16], st[17],
                st[18], st[19], st[20], st[21], st[22], st[23], st[24], st[25], st[26], st[27], st[28], st[29], st[30], st[31], st[32], st[33], st[40], st[39], st[37], st[38], st[41], st[42], st[46], st[43], st[44], st[45]);
        fprintf(stderr, "pa_fragani: sparse stats: %u segments with a candidate, %u candidates, %u groups of begins evaluated, %u begins, %u states, %u begins tying "
                        "their candidate's best when folded; %u one-run segments with strays; %u begins with enough hits left out because they lie between two tying begins and hold no more hits than those share, %u groups between the ties passed over unseen, %u groups that hold a candidate's first or last tying begin, %u candidates whose first hit lies past the batch of 512 window ids\n",
                st[48], st[49], st[50], st[51], st[52], st[53], st[47], st[58], st[59], st[54], st[55]);
      }
#endif
    }
    PA_TRY(W.matched.reserve((uint64_t)nq * n_genomes * 4));
    PA_TRY(W.ident_sum.reserve((uint64_t)nq * n_genomes * 8));
    hipLaunchKernelGGL(reduce_pairs_kernel, dim3(nq * n_genomes), dim3(64), 0, c->stream,
                       table_p, range_bins, W.genome_bin_off.as<uint32_t>(), n_genomes,
                       W.ident_tab.as<float>(), W.matched.as<uint32_t>(), W.ident_sum.as<double>());
    PA_HIP(hipGetLastError());
    prof.reset();
    if (out_cols == n_genomes) {
      PA_HIP(hipMemcpyAsync(h_matched + (uint64_t)g0 * n_genomes, W.matched.p, (uint64_t)nq * n_genomes * 4,
                            hipMemcpyDeviceToHost, c->stream));
      PA_HIP(hipMemcpyAsync(h_ident_sum + (uint64_t)g0 * n_genomes, W.ident_sum.p, (uint64_t)nq * n_genomes * 8,
                            hipMemcpyDeviceToHost, c->stream));
    } else if (out_cols) {  // the columns of the reference range only
      PA_HIP(hipMemcpy2DAsync(h_matched + (uint64_t)g0 * out_cols, (size_t)out_cols * 4, W.matched.as<uint32_t>() + out_col0,
                              (size_t)n_genomes * 4, (size_t)out_cols * 4, nq, hipMemcpyDeviceToHost, c->stream));
      PA_HIP(hipMemcpy2DAsync(h_ident_sum + (uint64_t)g0 * out_cols, (size_t)out_cols * 8, W.ident_sum.as<double>() + out_col0,
                              (size_t)n_genomes * 8, (size_t)out_cols * 8, nq, hipMemcpyDeviceToHost, c->stream));
    }
    PA_HIP(hipStreamSynchronize(c->stream));
    g0 = g1;
  }
  uint32_t h_over[2] = {0, 0};
  PA_HIP(hipMemcpy(h_over, d_overflow, 8, hipMemcpyDeviceToHost));
  if (h_over[0]) {
    pa_set_error("pa_fragani: %u fragment sketches exceeded %d minimizers (fragLen too long for this window); "
                 "those sketches were truncated", h_over[0], kQMax);
    return PA_E_CAPACITY;
  }
  return PA_OK;
}

}  // extern "C"
