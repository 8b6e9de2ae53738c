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

// ============================================================== 1. minimizers
constexpr int kPPT = 8;                       // positions per thread
constexpr int kTile = kThreads * kPPT;        // 2048 positions per workgroup, of which
constexpr int kHalo = 128;                    // the first 128 are look-back (needs w <= 64)
constexpr int kOwn = kTile - kHalo;

// One pass: every workgroup hashes and winnows its tile once, then learns where its minimizers go from the
// workgroups before it -- a chained scan with look-back (each publishes first the count of its own tile, then, once it
// knows it, the count of everything up to and including itself; a workgroup adds up published tile counts backwards
// until it meets such a running total).  Tiles are handed out by a ticket counter, so a workgroup only ever waits for
// workgroups that started before it.  `look` holds one 64-bit word per tile: state (0 nothing yet, 1 tile count,
// 2 running total) in the top two bits, the count below; `scalars`: [0] ticket, [1] total, [2] a wait ran out.
constexpr uint64_t kLookTile = 1ULL << 62, kLookTotal = 2ULL << 62;
constexpr uint32_t kLookSpinLimit = 1u << 24;
// (A ticket per RUN of consecutive tiles was tried to take load off the ticket counter -- one address, ~11 ns per
// returning atomic, 2.6 million tiles in the 1 000 x 5 Mb run: a floor of 29 ms under this 58 ms kernel -- and is wrong
// for a chained scan: the first tile of a run waits for the last tile of the run before it, which its workgroup
// reaches last, so the workgroups execute one after the other: 34 s.)

// A k-mer that holds residues other than A, C, G, T: fastANI hashes the characters as they are (upper-cased; the reverse
// complement leaves what it does not know in place), so such a k-mer is a k-mer like any other -- only a k-mer equal to its
// own reverse complement (a run of N, for one) is passed over, as every k-mer whose two strands hash alike is.  The packed
// arena keeps two bits per residue and one "not ACGT" bit: such a residue is 'N' (by far the commonest) unless the arena's
// list of other letters (IUPAC codes, ...) holds its position -- one search per such residue, only here.
// Rare, and off the hot path: the bytes are put together one by one, the two multiplies of MurmurHash3 done in full.
// `codes`: residue j in bits 2j, 2j+1; `bad`: bit j set = residue j is not ACGT; `pos`: arena position of residue 0.
// Returns kSkip when the strands hash alike.
// The residues that are neither ACGT nor N, as the packers list them (pa_fragani_set_ambiguous): ascending arena
// positions and upper-cased bytes.  The letter at arena position `pos`, of a residue whose "not ACGT" bit is set.
struct AmbiguousList {
  const uint64_t *pos;
  const uint8_t *byte;
  uint32_t n;
};
__device__ __forceinline__ uint32_t ambiguous_letter(const AmbiguousList &amb, uint64_t pos) {
  uint32_t lo = 0, hi = amb.n;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (amb.pos[mid] < pos) lo = mid + 1; else hi = mid;
  }
  return (lo < amb.n && amb.pos[lo] == pos) ? (uint32_t)amb.byte[lo] : (uint32_t)'N';
}

template <int K>
__device__ __forceinline__ uint32_t hash_kmer_with_unknowns(uint32_t codes, uint32_t bad, uint64_t pos, const AmbiguousList &amb) {
  constexpr int kWords = (K + 7) / 8;
  // the letters of the residues that are not ACGT, four to a word (only when the arena has a list: else every one is N)
  uint32_t letters[(K + 3) / 4];
#pragma unroll
  for (int q = 0; q < (K + 3) / 4; ++q) letters[q] = 0x4e4e4e4eu;  // "NNNN"
  if (amb.n) {
    for (uint32_t rest = bad; rest; rest &= rest - 1u) {
      const int j = __builtin_ctz(rest);
      const uint32_t ch = ambiguous_letter(amb, pos + (uint64_t)j);
#pragma unroll
      for (int q = 0; q < (K + 3) / 4; ++q)
        if ((j >> 2) == q) letters[q] = (letters[q] & ~(0xffu << (8 * (j & 3)))) | (ch << (8 * (j & 3)));
    }
  }
  uint32_t hs[2];
#pragma unroll
  for (int strand = 0; strand < 2; ++strand) {
    uint64_t P[4] = {0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < kWords; ++q) {
      uint64_t word = 0;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int j = 8 * q + t;  // byte j of the strand's text
        if (j >= K) break;
        const int src = strand ? K - 1 - j : j;
        const uint32_t code = ((codes >> (2 * src)) & 3u) ^ (strand ? 3u : 0u);
        const uint32_t ch = ((bad >> src) & 1u) ? (letters[src >> 2] >> (8 * (src & 3))) & 0xffu : (0x54474341u >> (8 * code)) & 0xffu;  // "ACGT"
        word |= (uint64_t)ch << (8 * t);
      }
      P[q] = word * ((q & 1) ? kC2 : kC1);
    }
    hs[strand] = (uint32_t)murmur3_from_products<K>(P);
  }
  return hs[0] == hs[1] ? kSkip : (hs[0] < hs[1] ? hs[0] : hs[1]);
}

template <int K>
__global__ __launch_bounds__(kThreads) void minimizer_kernel(
    const uint32_t *__restrict__ packed, const uint32_t *__restrict__ mask, uint64_t arena_bases,
    const uint64_t *__restrict__ contig_start, const uint32_t *__restrict__ contig_len, uint32_t n_contigs, int w,
    unsigned long long *__restrict__ look, uint32_t *__restrict__ scalars, uint32_t cap, uint32_t *__restrict__ out_hash,
    uint32_t *__restrict__ out_wpos, uint32_t *__restrict__ out_contig, uint32_t n_tiles, AmbiguousList amb) {
  static_assert(K >= 8 && K <= 16, "both k-mer registers are 32-bit");
  constexpr int kWords = (K + 7) / 8;
  // the first-multiply tables; once the hashes are there, the same memory holds the winnowing's suffix-minimum positions
  constexpr int kTabBytes = kWords * 256 * 12 > kTile * 2 ? kWords * 256 * 12 : kTile * 2;
  __shared__ __attribute__((aligned(16))) unsigned char s_tab_raw[kTabBytes];
  uint64_t (*s_lo)[256] = reinterpret_cast<uint64_t (*)[256]>(s_tab_raw);
  uint32_t (*s_hi)[256] = reinterpret_cast<uint32_t (*)[256]>(s_tab_raw + kWords * 256 * 8);
  uint16_t *s_sufp = reinterpret_cast<uint16_t *>(s_tab_raw);  // [kTile], after the hashing
  __shared__ uint32_t s_h[kTile];
  __shared__ int32_t s_mp[kTile];
  uint32_t *s_sufh = reinterpret_cast<uint32_t *>(s_mp);  // [kTile] suffix minima until the window minima are all known
  __shared__ uint32_t s_scan[kThreads / 64];
  __shared__ uint32_t s_tile, s_before;
  const uint32_t tid = threadIdx.x;
  if (tid == 0) s_tile = atomicAdd(&scalars[0], 1u);
#pragma unroll
  for (int j = 0; j < kWords; ++j) {
    const uint64_t cj = (j & 1) ? kC2 : kC1;
    s_lo[j][tid] = (uint64_t)ascii_group(tid, K - 8 * j) * cj;
    s_hi[j][tid] = (uint32_t)((uint64_t)ascii_group(tid, K - 8 * j - 4) * cj);
  }
  __syncthreads();
  const uint32_t tile = s_tile;
  const int64_t tile0 = (int64_t)tile * kOwn - kHalo;
  const int64_t p0 = tile0 + (int64_t)tid * kPPT;
  __syncthreads();

  // ---- both-strand hashes of the kPPT k-mers starting at p0 .. p0+kPPT-1
  uint32_t own[kPPT];  // the thread's hashes stay in registers for the winnowing
  {
    uint64_t bases = 0, bad = ~0ULL;  // bit j of `bad`: position p0+j is not a usable base
    if (p0 >= 0 && (uint64_t)p0 < arena_bases) {
      const uint64_t wi = (uint64_t)p0 >> 4, mi = (uint64_t)p0 >> 5;
      const uint64_t nw = arena_bases >> 4, nm = arena_bases >> 5;
      const uint64_t w0 = packed[wi], w1 = wi + 1 < nw ? packed[wi + 1] : 0u;
      bases = ((w1 << 32) | w0) >> (2 * ((uint32_t)p0 & 15u));
      const uint64_t m0 = mask[mi], m1 = mi + 1 < nm ? mask[mi + 1] : 0xffffffffu;
      bad = ((m1 << 32) | m0) >> ((uint32_t)p0 & 31u);
      if ((uint32_t)p0 & 31u) bad |= ~0ULL << (64 - ((uint32_t)p0 & 31u));  // beyond the two mask words: unusable
    }
    constexpr uint32_t kMask = (K == 16) ? 0xffffffffu : (uint32_t)((1ull << (2 * K)) - 1ull);
    constexpr uint64_t kBadMask = (1ULL << K) - 1;
    uint32_t fm = 0, fl = 0;
#pragma unroll
    for (int j = 0; j < K - 1; ++j) {
      const uint32_t b = (uint32_t)(bases >> (2 * j)) & 3u;
      fm = ((fm << 2) | b) & kMask;
      fl = (fl >> 2) | (b << (2 * (K - 1)));
    }
#pragma unroll
    for (int j = 0; j < kPPT; ++j) {
      const uint32_t b = (uint32_t)(bases >> (2 * (j + K - 1))) & 3u;
      fm = ((fm << 2) | b) & kMask;
      fl = (fl >> 2) | (b << (2 * (K - 1)));
      uint32_t h = kSkip;
      if (((bad >> j) & kBadMask) == 0) {
        const uint32_t strands[2] = {fl, fm ^ kMask};  // LSB-first forward, LSB-first reverse complement
        uint32_t hs[2];
#pragma unroll
        for (int sidx = 0; sidx < 2; ++sidx) {
          uint64_t P[4] = {0, 0, 0, 0};
#pragma unroll
          for (int q = 0; q < kWords; ++q) {
            const uint32_t glo = (strands[sidx] >> (16 * q)) & 0xffu, ghi = (strands[sidx] >> (16 * q + 8)) & 0xffu;
            const uint64_t lo = s_lo[q][glo];
            P[q] = u64_of((uint32_t)lo, (uint32_t)(lo >> 32) + s_hi[q][ghi]);
          }
          hs[sidx] = (uint32_t)murmur3_from_products<K>(P);
        }
        if (hs[0] != hs[1]) h = hs[0] < hs[1] ? hs[0] : hs[1];
      }
      s_h[tid * kPPT + j] = h;
      own[j] = h;
    }
  }
  uint32_t c = 0;
  bool have_c = false;
  // the contig of the thread's first position, and its bounds in registers: they change at a contig boundary only
  // (asking contig_start again for every position was a load, and a wait, per position)
  uint64_t c_beg = 0, c_next = ~0ULL;
  uint32_t c_len = 0;
  if (p0 >= 0 && (uint64_t)p0 < arena_bases) {
    c = contig_of(contig_start, n_contigs, (uint64_t)p0);
    have_c = true;
    c_beg = contig_start[c];
    c_len = contig_len[c];
    c_next = c + 1 < n_contigs ? contig_start[c + 1] : ~0ULL;
  }
  // Residues other than ACGT in a k-mer -- or the padding between contigs, marked the same way: a k-mer that starts in it
  // or runs into it is none.  Few threads ever come here, and it shows nowhere else only if nothing here looks like the
  // hot path: held across the loop above, the residues and mask bits cost the sixteen hashes their registers; a second
  // search through the contigs' starts was merged with every thread's own one (a quarter of the kernel's time, measured
  // variant by variant) -- so the mask words are asked for again (they are in the cache), the residues only by the threads
  // that need them, and the k-mer's contig is found by walking on from the thread's.
  if (have_c) {
    constexpr uint32_t kMask = (K == 16) ? 0xffffffffu : (uint32_t)((1ull << (2 * K)) - 1ull);
    constexpr uint64_t kBadMask = (1ULL << K) - 1;
    const uint64_t mi = (uint64_t)p0 >> 5, nm = arena_bases >> 5;
    const uint64_t m0 = mask[mi], m1 = mi + 1 < nm ? mask[mi + 1] : 0xffffffffu;
    const uint64_t bad = ((m1 << 32) | m0) >> ((uint32_t)p0 & 31u);  // (bits past the two words: never reached by 8 + 16 residues)
    uint32_t unknowns = 0;  // bit j: the k-mer at p0 + j holds a residue that is not ACGT
#pragma unroll
    for (int j = 0; j < kPPT; ++j) unknowns |= (((bad >> j) & kBadMask) != 0 ? 1u : 0u) << j;
    if (__builtin_expect(unknowns != 0u, 0)) {
      const uint64_t wi = (uint64_t)p0 >> 4, nw = arena_bases >> 4;
      const uint64_t w0 = packed[wi], w1 = wi + 1 < nw ? packed[wi + 1] : 0u;
      const uint64_t bases = ((w1 << 32) | w0) >> (2 * ((uint32_t)p0 & 15u));
      uint64_t beg = c_beg, next = c_next;
      uint32_t cc = c, len = c_len;
#pragma unroll 1  // one copy of the code, the thread's positions one after the other; the registers take the results from LDS
      for (uint32_t rest = unknowns; rest; rest &= rest - 1u) {
        const int j = __builtin_ctz(rest);
        const uint64_t pos = (uint64_t)(p0 + j);
        if (pos >= arena_bases) break;
        while (pos >= next) {  // into the next contig
          ++cc;
          beg = next;
          len = contig_len[cc];
          next = cc + 1 < n_contigs ? contig_start[cc + 1] : ~0ULL;
        }
        if (pos - beg + K > len) continue;
        s_h[tid * kPPT + j] = hash_kmer_with_unknowns<K>((uint32_t)(bases >> (2 * j)) & kMask, (uint32_t)((bad >> j) & kBadMask), pos, amb);
      }
#pragma unroll
      for (int j = 0; j < kPPT; ++j) own[j] = s_h[tid * kPPT + j];
    }
  }
  __syncthreads();

  // ---- winnowing minimum (rightmost on ties) for every position that can be asked about, from minima over the
  // groups of eight positions a thread owns (a sliding minimum in two pieces): SUFFIX minima of every group go to LDS
  // (position y: the minimum of y .. end of its group), prefix minima of the own group stay in registers.  A window of
  // w >= 9 positions ending at x = 8 tid + j is then the own prefix up to j, the whole groups to the left of it (their
  // suffix minimum at the group's first position), and the suffix minimum at the window's first position in the group
  // it starts in: three or four LDS reads per position for w = 24 instead of seventeen.
  {
    uint32_t bh = own[kPPT - 1];
    uint32_t bp = tid * kPPT + kPPT - 1;
    s_sufh[bp] = bh;
    s_sufp[bp] = (uint16_t)bp;
#pragma unroll
    for (int j = kPPT - 2; j >= 0; --j) {  // right to left, strictly smaller wins: the rightmost of equal hashes stays
      if (own[j] < bh) { bh = own[j]; bp = tid * kPPT + (uint32_t)j; }
      s_sufh[tid * kPPT + j] = bh;
      s_sufp[tid * kPPT + j] = (uint16_t)bp;
    }
  }
  __syncthreads();
  uint32_t local[kPPT];
  uint32_t cidx[kPPT];
  int32_t mpv[kPPT];     // position of the window minimum (tile coordinates), -1: no window ends here
  uint32_t besth[kPPT];  // its hash
  uint32_t pre_h = kSkip;  // minimum of the own positions 0 .. j, the rightmost of equal ones
  int pre_p = 0;
#pragma unroll
  for (int j = 0; j < kPPT; ++j) {
    const int x = (int)tid * kPPT + j;
    if (own[j] <= pre_h) { pre_h = own[j]; pre_p = x; }
    int32_t mp = -1;
    uint32_t best = 0;
    local[j] = 0; cidx[j] = 0;
    if (have_c && x >= kHalo / 2) {
      const uint64_t pos = (uint64_t)(p0 + j);
      while (pos >= c_next) {  // into the next contig
        ++c;
        c_beg = c_next;
        c_len = contig_len[c];
        c_next = c + 1 < n_contigs ? contig_start[c + 1] : ~0ULL;
      }
      const uint64_t loc = pos - c_beg;
      if (loc < c_len && loc + 1 >= (uint64_t)w && own[j] != kSkip) {
        if (w > kPPT) {
          const int a = x - w + 1, ga = a / kPPT;  // a >= 1: x >= kHalo / 2 and w <= 64; ga < tid
          best = pre_h;
          mp = pre_p;
          for (int g = (int)tid - 1; g > ga; --g) {
            const uint32_t hg = s_sufh[g * kPPT];
            if (hg < best) { best = hg; mp = s_sufp[g * kPPT]; }
          }
          const uint32_t ha = s_sufh[a];
          if (ha < best) { best = ha; mp = s_sufp[a]; }
        } else {
          best = own[j];
          mp = x;
          for (int y = x - 1; y > x - w; --y) {
            const uint32_t hy = s_h[y];
            if (hy < best) { best = hy; mp = y; }
          }
        }
        local[j] = (uint32_t)loc;
        cidx[j] = c;
      }
    }
    mpv[j] = mp;
    besth[j] = best;
  }
  __syncthreads();  // the suffix minima have been read: their memory takes the window minima
#pragma unroll
  for (int j = 0; j < kPPT; ++j) s_mp[tid * kPPT + j] = mpv[j];
  __syncthreads();

  // ---- a minimizer is recorded when it differs from the previous usable window's
  uint32_t flags = 0, cnt = 0;
#pragma unroll
  for (int j = 0; j < kPPT; ++j) {
    const int x = (int)tid * kPPT + j;
    if (x < kHalo) continue;
    const int32_t mp = s_mp[x];
    if (mp < 0) continue;
    int32_t prev = -2;
    for (int y = x - 1; y > x - w; --y) {
      const int32_t my = s_mp[y];
      if (my >= 0) { prev = my; break; }
    }
    if (prev != mp) { flags |= 1u << j; ++cnt; }
  }
  // block exclusive scan of cnt
  const uint32_t lane = tid & 63u, wave = tid >> 6;
  const uint32_t wex = wave_excl_scan(cnt, lane);
  if (lane == 63) s_scan[wave] = wex + cnt;
  __syncthreads();
  uint32_t pre = 0, total = 0;
#pragma unroll
  for (int q = 0; q < kThreads / 64; ++q) {
    if ((uint32_t)q < wave) pre += s_scan[q];
    total += s_scan[q];
  }
  // ---- minimizers of all tiles before this one
  if (wave == 0) {
    uint32_t before = 0;
    if (tile == 0) {
      if (lane == 0) __hip_atomic_store(&look[0], kLookTotal | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if (lane == 0) __hip_atomic_store(&look[tile], kLookTile | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int64_t at = (int64_t)tile - 1;  // lane l looks at tile at - l
      uint32_t spins = 0;
      for (;;) {
        const int64_t idx = at - (int64_t)lane;
        const uint64_t v = idx >= 0 ? __hip_atomic_load(&look[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kLookTotal;
        const uint32_t state = (uint32_t)(v >> 62);
        const uint64_t totals = __ballot(state == 2u), missing = __ballot(state == 0u);
        const uint32_t stop = totals ? (uint32_t)__builtin_ctzll(totals) : 64u;  // first lane holding a running total
        const uint64_t needed = stop >= 63u ? ~0ULL : ((2ULL << stop) - 1ULL);
        if (missing & needed) {  // a tile in front has not published yet
          if (++spins > kLookSpinLimit) { if (lane == 0) scalars[2] = 1u; break; }
          __builtin_amdgcn_s_sleep(1);
          continue;
        }
        before += wave_sum(lane <= stop ? (uint32_t)v : 0u);
        if (stop < 64u) break;
        at -= 64;
      }
      if (lane == 0)
        __hip_atomic_store(&look[tile], kLookTotal | (uint64_t)(before + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (lane == 0) {
      s_before = before;
      if (tile == n_tiles - 1) scalars[1] = before + total;
    }
  }
  __syncthreads();
  uint32_t o = s_before + pre + wex;
#pragma unroll
  for (int j = 0; j < kPPT; ++j) {
    if (!((flags >> j) & 1u)) continue;
    if (o < cap) {  // a run that overflows the estimate is repeated with the exact size
      out_hash[o] = besth[j];
      out_wpos[o] = local[j] - (uint32_t)w + 1u;
      out_contig[o] = cidx[j];
    }
    ++o;
  }
}

__global__ __launch_bounds__(kThreads) void contig_offsets_kernel(const uint32_t *__restrict__ mini_contig, uint32_t m,
                                                                   uint32_t n_contigs, uint32_t *__restrict__ off) {
  const uint32_t c = blockIdx.x * kThreads + threadIdx.x;
  if (c <= n_contigs) off[c] = lower_bound_u32(mini_contig, 0, m, c);
}

// bucket_first[base_c + b] = first minimizer of contig c with window id >= b*256, b = 0 .. n_buckets_c
__global__ __launch_bounds__(kThreads) void bucket_index_kernel(const uint32_t *__restrict__ mini_wpos,
                                                                const uint32_t *__restrict__ contig_mini_off,
                                                                const uint32_t *__restrict__ contig_bucket_off,
                                                                uint32_t n_contigs, uint32_t total_entries,
                                                                uint32_t *__restrict__ bucket_first) {
  const uint32_t e = blockIdx.x * kThreads + threadIdx.x;
  if (e >= total_entries) return;
  uint32_t lo = 0, hi = n_contigs;  // contig owning entry e
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (contig_bucket_off[mid] <= e) lo = mid; else hi = mid;
  }
  const uint32_t c = lo, b = e - contig_bucket_off[c];
  bucket_first[e] = lower_bound_u32(mini_wpos, contig_mini_off[c], contig_mini_off[c + 1], b << kBucketShift);
}

// ============================================================== 2. dictionary of minimizer hashes
// Sort key of a minimizer: its hash in the low word -- the only bits the radix passes look at -- and its window id as a
// passenger in the high word, so that the posting build reads it in posting order instead of gathering it.
// (`base`: the first minimizer the dictionary holds -- those of the reference genomes asked for; values = minimizer indices)
__global__ __launch_bounds__(kThreads) void mini_keys_kernel(const uint32_t *__restrict__ hash, const uint32_t *__restrict__ wpos,
                                                             uint32_t base, uint32_t m, uint64_t *__restrict__ keys,
                                                             uint32_t *__restrict__ vals) {
  const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
  if (i < m) { keys[i] = ((uint64_t)wpos[base + i] << 32) | hash[base + i]; vals[i] = base + i; }
}
// Where the dictionary holds the reference range's minimizers only, a query minimizer finds its hash by value: an open
// table of {hash, first posting, postings | mark of the frequency cut << 31, -} entries, 16 bytes each, at least two slots
// per hash, linear probing -- one load per look-up where a search through the sorted hashes and the list bounds were eight
// (19 ms per batch against 1 for 125 of 1 000 genomes).  The slot comes from the hash multiplied by the golden-ratio
// constant: a minimizer's hash is the MINIMUM of a window of hashes, and minima crowd at the low end -- slotted by their top
// bits they filled the table's first tenth and every probe walked through it (28 s for twenty genomes).
__device__ __forceinline__ uint32_t lookup_slot(uint32_t h, uint32_t table_bits) { return (h * 0x9e3779b1u) >> (32u - table_bits); }
__global__ __launch_bounds__(kThreads) void lookup_insert_kernel(const uint32_t *__restrict__ uniq_hash,
                                                                 const uint32_t *__restrict__ post_start,
                                                                 const uint32_t *__restrict__ hash_cut, uint32_t n_ids,
                                                                 uint32_t table_bits, uint4 *__restrict__ table) {
  const uint32_t id = blockIdx.x * kThreads + threadIdx.x;
  if (id >= n_ids) return;
  const uint32_t h = uniq_hash[id], lo = post_start[id], cnt = post_start[id + 1] - lo;
  const uint32_t mark = (hash_cut[id >> 5] >> (id & 31u)) & 1u;
  const uint32_t mask = (1u << table_bits) - 1u;
  unsigned long long *words = reinterpret_cast<unsigned long long *>(table);
  for (uint32_t slot = lookup_slot(h, table_bits);; slot = (slot + 1u) & mask) {
    if (atomicCAS(&words[2ull * slot], ~0ULL, ((unsigned long long)lo << 32) | h) == ~0ULL) {
      table[slot].z = cnt | (mark << 31);
      return;
    }
  }
}
__global__ __launch_bounds__(kThreads) void key_heads_kernel(const uint64_t *__restrict__ keys, uint32_t m,
                                                             uint32_t *__restrict__ flags) {
  const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
  if (i < m) flags[i] = (i == 0 || (uint32_t)keys[i] != (uint32_t)keys[i - 1]) ? 1u : 0u;
}
__global__ __launch_bounds__(kThreads) void postings_kernel(const uint64_t *__restrict__ keys,
                                                            const uint32_t *__restrict__ sorted_idx,
                                                            const uint32_t *__restrict__ flags,
                                                            const uint32_t *__restrict__ pos, uint32_t m, uint32_t n_ids,
                                                            const uint32_t *__restrict__ mini_contig,
                                                            uint32_t *__restrict__ mini_id, uint32_t *__restrict__ post_start,
                                                            int32_t *__restrict__ prev_same,
                                                            const uint32_t *__restrict__ mini_wpos,
                                                            const uint32_t *__restrict__ contig_genome,
                                                            uint64_t *__restrict__ post_cw,
                                                            uint16_t *__restrict__ post_genome,
                                                            const uint32_t *__restrict__ contig_mini_off, uint32_t n_contigs,
                                                            uint32_t *__restrict__ uniq_hash) {
  // the contig of the posting before this one comes from the thread before it (LDS) instead of a second random read
  __shared__ uint32_t s_contig[kThreads];
  const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
  const bool in = i < m;
  uint32_t id = 0, me = 0, mc = 0;
  if (in) {
    id = pos[i] + flags[i] - 1u;
    me = sorted_idx[i];
    // the minimizers are stored contig by contig: the contig of minimizer `me` is a search in the contigs' first
    // minimizers (a few KB, cached) instead of one more random 4-byte read from HBM
    uint32_t lo = 0, hi = n_contigs;  // largest c with contig_mini_off[c] <= me
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) >> 1;
      if (contig_mini_off[mid] <= me) lo = mid; else hi = mid;
    }
    mc = lo;
  }
  s_contig[threadIdx.x] = mc;
  __syncthreads();
  if (!in) return;
  mini_id[me] = id;
  // the posting as the low 44 bits of a hit key, its genome on top so that bucketing needs no second lookup
  const uint32_t pg = contig_genome[mc];
  const uint64_t key = keys[i];
  post_cw[i] = ((uint64_t)pg << 44) | ((uint64_t)mc << 24) | (uint32_t)(key >> 32);  // the window id rode along in the sort key
  post_genome[i] = (uint16_t)pg;  // (the bucketed seeding and the frequency cut read this; at most 65 535 genomes)
  if (flags[i]) { post_start[id] = i; uniq_hash[id] = (uint32_t)key; }
  if (i == m - 1) post_start[n_ids] = m;
  // "the same hash earlier in this contig": rare (repeats inside a contig), and the array has been filled with -1
  if (i > 0 && (uint32_t)key == (uint32_t)keys[i - 1]) {
    const uint32_t other = sorted_idx[i - 1];  // stable sort: other < me
    const uint32_t oc = threadIdx.x > 0 ? s_contig[threadIdx.x - 1] : mini_contig[other];
    if (oc == mc) prev_same[me] = (int32_t)other;
  }
}

// ---- Mashmap's frequency cut of the seed look-up (fastANI logs it: "ignore minimizers occurring >= N times during
// lookup").  The reference sketch of a fastANI process is ONE genome; the minimizers it holds are counted per hash, and
// the most frequent ones -- as many bars of the histogram of counts, from the top, as stay within 0.001 % of the distinct
// minimizers -- give no seed hits.  Here the postings of a hash are ordered by arena position, so the occurrences of a
// hash in one genome are a RUN of its list: a run of c >= 2 is counted into its genome's histogram (c >= kFreqBins - 1:
// listed exactly instead), the host walks the histograms (a few KB per genome), and the runs at or above their genome's
// threshold are taken out of the posting lists -- everything after the index sees lists that never held them.  The
// minimizers themselves stay where they are: the L2 windows hold every minimizer, as fastANI's do.
constexpr uint32_t kFreqBins = 256;
// The runs are numbered first -- a flag at every run's first posting, their prefix sums, the runs' first postings
// gathered by number -- so that a run's length is a difference of two entries, whoever asks: one thread per POSTING
// everywhere below.  (A thread walking its run cost the run's length in dependent loads, twice per index build: a
// homopolymer or an array of a short unit in a reference is one minimizer per position with one hash, runs of 10^5-10^6
// postings on one lane while the grid idled.)
__global__ __launch_bounds__(kThreads) void posting_run_flags_kernel(const uint32_t *__restrict__ heads,
                                                                     const uint16_t *__restrict__ post_genome, uint32_t m,
                                                                     uint32_t *__restrict__ run_flag) {
  const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
  if (i < m) run_flag[i] = (i == 0 || heads[i] || post_genome[i - 1] != post_genome[i]) ? 1u : 0u;
}
__global__ __launch_bounds__(kThreads) void posting_run_starts_kernel(const uint32_t *__restrict__ run_flag,
                                                                      const uint32_t *__restrict__ runs_before, uint32_t m,
                                                                      uint32_t *__restrict__ run_start) {
  const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= m) return;
  if (run_flag[i]) run_start[runs_before[i]] = i;
  if (i == m - 1) run_start[runs_before[i] + run_flag[i]] = m;  // the closing entry
}
__global__ __launch_bounds__(kThreads) void posting_run_hist_kernel(const uint32_t *__restrict__ run_flag,
                                                                    const uint32_t *__restrict__ runs_before,
                                                                    const uint32_t *__restrict__ run_start,
                                                                    const uint16_t *__restrict__ post_genome, uint32_t m,
                                                                    uint32_t *__restrict__ hist /* [genomes][kFreqBins] */,
                                                                    uint32_t *__restrict__ dups /* [genomes] */,
                                                                    uint2 *__restrict__ over, uint32_t over_cap,
                                                                    uint32_t *__restrict__ over_n) {
  const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= m || !run_flag[i]) return;  // a run's first posting speaks for the run
  const uint32_t c = run_start[runs_before[i] + 1u] - i;
  if (c < 2u) return;  // a run of one: the common case, counted by difference
  const uint16_t g = post_genome[i];
  atomicAdd(&dups[g], c - 1u);
  if (c < kFreqBins - 1u) {
    atomicAdd(&hist[(uint64_t)g * kFreqBins + c], 1u);
  } else {
    atomicAdd(&hist[(uint64_t)g * kFreqBins + kFreqBins - 1u], 1u);
    const uint32_t at = atomicAdd(over_n, 1u);
    if (at < over_cap) over[at] = make_uint2(g, c);
  }
}
// keep[i] = 0 for the postings of runs at or above their genome's threshold, 1 for the others.  `keep` is the memory of
// the run flags: a thread reads its own flag before it writes its own entry.
__global__ __launch_bounds__(kThreads) void posting_cut_flags_kernel(const uint32_t *__restrict__ heads,
                                                                     const uint32_t *__restrict__ ids_before,
                                                                     const uint32_t *__restrict__ runs_before,
                                                                     const uint32_t *__restrict__ run_start,
                                                                     const uint16_t *__restrict__ post_genome, uint32_t m,
                                                                     const uint32_t *__restrict__ threshold,
                                                                     uint32_t *__restrict__ keep /* in: run flags */,
                                                                     uint32_t *__restrict__ hash_cut) {
  const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= m) return;
  const uint32_t first = keep[i];  // the run flag
  const uint32_t run = runs_before[i] + first - 1u;
  const uint32_t c = run_start[run + 1u] - run_start[run];
  const uint32_t thr = threshold[post_genome[i]];
  const bool cut = thr != 0xffffffffu && c >= thr;
  keep[i] = cut ? 0u : 1u;
  if (cut && first) {
    // the hash has lost seed hits somewhere: its matches in an L2 window are no longer all among the seed hits (the mapping
    // kernel's bounds allow for them).  A bit per hash (dense id: the hashes before this posting's own, plus one where it
    // is its hash's first) says so; mark_cut_minimizers_kernel hands it on to the minimizers.
    const uint32_t id = ids_before[i] + heads[i] - 1u;
    atomicOr(&hash_cut[id >> 5], 1u << (id & 31u));
  }
}
// Every minimizer whose hash lost seed hits -- in whatever genome: it is the QUERY's minimizers that are asked -- carries
// the mark in the top bit of its hash id, which the sketch kernel reads anyway.  One thread per posting.
__global__ __launch_bounds__(kThreads) void mark_cut_minimizers_kernel(const uint32_t *__restrict__ heads,
                                                                       const uint32_t *__restrict__ ids_before,
                                                                       const uint32_t *__restrict__ sorted_idx, uint32_t m,
                                                                       const uint32_t *__restrict__ hash_cut,
                                                                       uint32_t *__restrict__ mini_id) {
  const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= m) return;
  const uint32_t id = ids_before[i] + heads[i] - 1u;
  if ((hash_cut[id >> 5] >> (id & 31u)) & 1u) mini_id[sorted_idx[i]] |= 0x80000000u;
}
__global__ __launch_bounds__(kThreads) void posting_compact_kernel(const uint32_t *__restrict__ keep, const uint32_t *__restrict__ at,
                                                                   const uint64_t *__restrict__ post_cw,
                                                                   const uint16_t *__restrict__ post_genome,
                                                                   const uint32_t *__restrict__ sorted_idx, uint32_t m,
                                                                   uint64_t *__restrict__ cw_out, uint16_t *__restrict__ g_out,
                                                                   uint32_t *__restrict__ idx_out) {
  const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
  if (i < m && keep[i]) { cw_out[at[i]] = post_cw[i]; g_out[at[i]] = post_genome[i]; idx_out[at[i]] = sorted_idx[i]; }
}
__global__ __launch_bounds__(kThreads) void posting_starts_kernel(uint32_t *__restrict__ post_start, uint32_t n_ids,
                                                                  const uint32_t *__restrict__ at, uint32_t m, uint32_t kept) {
  const uint32_t id = blockIdx.x * kThreads + threadIdx.x;
  if (id > n_ids) return;
  const uint32_t old = post_start[id];
  post_start[id] = old < m ? at[old] : kept;
}

// Keys ordered in registers (a segment's hits by (contig, window id) in the mapping kernel, a fragment's minimizers by
// hash below): E keys per lane, element e = lane * E + q, bitonic network over 64 E elements.  Strides of E and more exchange between lanes (ds_bpermute, no LDS memory, no barrier),
// the strides below E between the registers of a lane.  Missing elements are keys above any real one.  Key = uint32_t
// (contig relative to the genome's first in 8 bits | window id in 24) when the segment's contigs allow it -- one
// shuffle, a minimum, a maximum and a select per key and stage --, else uint64_t (20 + 24 bits).
template <typename Key>
__device__ __forceinline__ Key lane_exchange(Key v, int partner_byte_address) {
  if constexpr (sizeof(Key) == 4) {
    return (Key)__builtin_amdgcn_ds_bpermute(partner_byte_address, (int)v);
  } else {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute(partner_byte_address, (int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute(partner_byte_address, (int)(uint32_t)(v >> 32));
    return ((Key)hi << 32) | lo;
  }
}
template <int E, int J, typename Key>
__device__ __forceinline__ void bitonic_inside_lane(Key (&k)[E], uint32_t e0, uint32_t span) {
#pragma unroll
  for (int q = 0; q < E; ++q) {
    if ((q & J) == 0) {
      const bool asc = ((e0 + (uint32_t)q) & span) == 0u;
      const Key a = k[q], b = k[q | J];
      const Key lo = a < b ? a : b, hi = a < b ? b : a;
      k[q] = asc ? lo : hi;
      k[q | J] = asc ? hi : lo;
    }
  }
}
template <int E, typename Key>
__device__ __forceinline__ void bitonic_sort_lanes(Key (&k)[E], uint32_t lane) {
  const uint32_t e0 = lane * (uint32_t)E;
  for (uint32_t span = 2; span <= 64u * (uint32_t)E; span <<= 1) {
    for (uint32_t j = span >> 1; j > 0; j >>= 1) {
      if (j >= (uint32_t)E) {
        const uint32_t lj = j / (uint32_t)E;
        const int partner = (int)((lane ^ lj) << 2);  // byte address of the partner lane for ds_bpermute
        // the lower lane of a pair keeps the smaller key in an ascending run: span > j >= E, so the run's direction is a
        // bit of the lane number, the same for all E keys of the lane
        const bool take_min = ((lane & lj) == 0u) == ((lane & (span / (uint32_t)E)) == 0u);
#pragma unroll
        for (int q = 0; q < E; ++q) {
          const Key other = lane_exchange<Key>(k[q], partner);
          if constexpr (sizeof(Key) == 4) {
            const Key lo = min(k[q], other), hi = max(k[q], other);
            k[q] = take_min ? lo : hi;
          } else {
            k[q] = ((other < k[q]) == take_min) ? other : k[q];
          }
        }
      } else if (E > 4 && j == 4u) {
        bitonic_inside_lane<E, (E > 4 ? 4 : 1), Key>(k, e0, span);
      } else if (E > 2 && j == 2u) {
        bitonic_inside_lane<E, (E > 2 ? 2 : 1), Key>(k, e0, span);
      } else if (E > 1) {
        bitonic_inside_lane<E, 1, Key>(k, e0, span);
      }
    }
  }
}
// ============================================================== 3. fragment sketches
// fastANI sketches a fragment on its own: winnowing restarts at the fragment's first residue, and the first minimizer is
// selected at the window of the first USED k-mer at or after the fragment's w-th.  Here a fragment's sketch is a slice of its
// genome's minimizers, and that window is all the slice needs to know: d = the windows at the fragment's start at which
// nothing is selected = the k-mers from the w-th on that equal their own reverse complement (both strands hash alike; a
// residue that is not ACGT counts as the N it is hashed as, and stays where it is in the reverse complement: a run of N is
// such a k-mer), 0 nearly always; count_windows when no k-mer of the fragment is used from there on.
// One thread per fragment of the batch (the answer is in the first k-mer nearly always; the wave that builds the sketch
// would only wait for it).  (x0: arena position of the fragment's w-th k-mer.)
__global__ __launch_bounds__(kThreads) void windows_without_selection_kernel(
    const uint32_t *__restrict__ packed, const uint32_t *__restrict__ mask, uint64_t arena_bases,
    const uint64_t *__restrict__ contig_start, uint32_t k, uint32_t w, const uint32_t *__restrict__ frag_contig,
    const uint32_t *__restrict__ frag_no, uint32_t n_frags, uint32_t frag_len, uint32_t count_windows,
    uint32_t *__restrict__ frag_d, AmbiguousList amb) {
  const uint32_t f = blockIdx.x * kThreads + threadIdx.x;
  if (f >= n_frags) return;
  const uint64_t x0 = contig_start[frag_contig[f]] + (uint64_t)frag_no[f] * frag_len + w - 1u;
  const uint32_t k_mask = k == 16u ? 0xffffffffu : ((1u << (2u * k)) - 1u);
  const uint64_t nw = arena_bases >> 4, nm = arena_bases >> 5;
  uint32_t d = 0;
  for (; d < count_windows; ++d) {
    const uint64_t x = x0 + d, wi = x >> 4, mi = x >> 5;
    const uint64_t w0 = wi < nw ? packed[wi] : 0u, w1 = wi + 1 < nw ? packed[wi + 1] : 0u;
    const uint32_t fwd = (uint32_t)(((w1 << 32) | w0) >> (2u * ((uint32_t)x & 15u))) & k_mask;  // base j in bits 2j, 2j+1
    const uint64_t m0 = mi < nm ? mask[mi] : 0xffffffffu, m1 = mi + 1 < nm ? mask[mi + 1] : 0xffffffffu;
    const uint32_t bad = (uint32_t)(((m1 << 32) | m0) >> ((uint32_t)x & 31u)) & ((1u << k) - 1u);
    // the reverse complement in the same layout: the 2-bit groups in reverse order, complemented; the unknown residues
    // in reverse order, as they are
    uint32_t r = __brev(fwd);
    r = ((r >> 1) & 0x55555555u) | ((r & 0x55555555u) << 1);
    const uint32_t rc = (r >> (32u - 2u * k)) ^ k_mask;
    const uint32_t rbad = __brev(bad) >> (32u - k);
    uint32_t known = ~bad & 0xffffu;  // one bit per residue -> two
    known = (known | (known << 8)) & 0x00ff00ffu;
    known = (known | (known << 4)) & 0x0f0f0f0fu;
    known = (known | (known << 2)) & 0x33333333u;
    known = (known | (known << 1)) & 0x55555555u;
    known |= known << 1;
    if (!(bad == rbad && ((fwd ^ rc) & known & k_mask) == 0u)) break;  // a used k-mer
    // residues that are not ACGT stay where they are in the reverse complement: the strands are the same text only if the
    // letters at mirrored positions are the same letter (all N without a list)
    if (bad && amb.n) {
      bool same = true;
      for (uint32_t rest = bad; rest && same; rest &= rest - 1u) {
        const uint32_t j = (uint32_t)__builtin_ctz(rest);
        if (j < k - 1u - j) same = ambiguous_letter(amb, x + j) == ambiguous_letter(amb, x + (k - 1u - j));
      }
      if (!same) break;  // a used k-mer
    }
  }
  frag_d[f] = d;
}

// one wave per fragment: slice of the contig's minimizers, sorted by (hash, slice index), first of each hash kept
__global__ __launch_bounds__(kThreads) void query_sketch_kernel(
    const uint32_t *__restrict__ frag_d, const uint32_t *__restrict__ frag_contig, const uint32_t *__restrict__ frag_no, uint32_t n_frags, uint32_t frag_len,
    uint32_t count_windows, const uint32_t *__restrict__ contig_mini_off, const uint32_t *__restrict__ contig_bucket_off,
    const uint32_t *__restrict__ bucket_first, const uint32_t *__restrict__ mini_hash,
    const uint32_t *__restrict__ mini_wpos, const uint32_t *__restrict__ mini_id, const uint32_t *__restrict__ post_start,
    uint32_t *__restrict__ q_hash, uint32_t *__restrict__ q_pos /* posting list length */,
    uint32_t *__restrict__ q_id /* first posting */, uint32_t *__restrict__ q_s,
    uint32_t *__restrict__ hit_count, uint32_t *__restrict__ overflow, uint32_t *__restrict__ max_hits,
    uint32_t *__restrict__ q_cut /* hashes of the sketch that lost seed hits to the frequency cut */,
    const uint4 *__restrict__ lookup /* null: every minimizer knows its hash id */, uint32_t lookup_bits) {
  __shared__ uint64_t s_key[kThreads / 64][kQMax];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t f = blockIdx.x * (kThreads / 64) + wave;
  const bool active = f < n_frags;  // every wave takes part in the barriers below
  uint64_t *key = s_key[wave];
  uint32_t p = 0, b0 = 0, n = 0;
  if (active) {
    const uint32_t c = frag_contig[f];
    p = frag_no[f] * frag_len;
    const uint32_t m0 = contig_mini_off[c], m1 = contig_mini_off[c + 1];
    const uint32_t bb = contig_bucket_off[c], nb = contig_bucket_off[c + 1] - bb - 1;
    const uint32_t b = wpos_lower_bound(mini_wpos, bucket_first, bb, nb, p);
    const uint32_t e = wpos_lower_bound(mini_wpos, bucket_first, bb, nb, p + count_windows);
    // the minimizer recorded last before the fragment belongs to its sketch unless a new one is recorded by the first
    // window at which the fragment, sketched alone, selects any
    const uint32_t d = frag_d[f];  // windows_without_selection_kernel
    const bool fresh = b < m1 && mini_wpos[b] <= p + d;
    b0 = (!fresh && b > m0) ? b - 1 : b;
    n = d < count_windows ? e - b0 : 0u;  // (no used k-mer from the w-th on: no sketch)
  }
  // A slice of more than kQMax minimizers is low-complexity sequence: inside a homopolymer run or an array of a short
  // unit every window records its (rightmost) minimum anew -- one minimizer per position, all with the same hash.  Runs of
  // equal hashes are taken as one entry each on the way in (the sketch is the set of hashes); only a slice that still
  // holds more than kQMax entries is refused.
  const bool long_slice = n > (uint32_t)kQMax;  // (uniform: one fragment per wave)
  if (long_slice) {
    uint32_t o = 0, carry = 0;
    for (uint32_t base = 0; base < n; base += 64) {
      const uint32_t e = base + lane;
      const uint32_t h = e < n ? mini_hash[b0 + e] : 0u;
      uint32_t before = (uint32_t)__shfl_up((int)h, 1, 64);
      if (lane == 0) before = carry;
      const bool keep = e < n && (e == 0u || h != before);
      const uint64_t bal = __ballot(keep);
      const uint32_t at = o + (uint32_t)__popcll(bal & ((1ULL << lane) - 1ULL));
      if (keep && at < (uint32_t)kQMax) key[at] = ((uint64_t)h << 16) | e;  // (e <= count_windows <= 65 535)
      o += (uint32_t)__popcll(bal);
      carry = (uint32_t)__shfl((int)h, 63, 64);
    }
    if (o > (uint32_t)kQMax) { if (lane == 0) atomicAdd(overflow, 1u); o = kQMax; }
    n = o;
    __builtin_amdgcn_wave_barrier();
  }
  // the slice's (hash, slice index) keys ordered in registers, E = 1, 2, 4 or 8 per lane by the size of the slice, and
  // left in LDS for the pass below (every wave has its own keys: no barrier between the waves of the workgroup)
  auto sort_slice = [&](auto e_tag) {
    constexpr int E = decltype(e_tag)::value;
    uint64_t k[E];
#pragma unroll
    for (int q = 0; q < E; ++q) {
      const uint32_t e = lane * (uint32_t)E + (uint32_t)q;
      k[q] = e < n ? (long_slice ? key[e] : (((uint64_t)mini_hash[b0 + e] << 16) | e)) : ~0ULL;
    }
    __builtin_amdgcn_wave_barrier();  // (a long slice: every lane has its entries before any is overwritten)
    bitonic_sort_lanes<E, uint64_t>(k, lane);
#pragma unroll
    for (int q = 0; q < E; ++q) key[lane * (uint32_t)E + (uint32_t)q] = k[q];
  };
  if (n <= 64u) sort_slice(std::integral_constant<int, 1>{});
  else if (n <= 128u) sort_slice(std::integral_constant<int, 2>{});
  else if (n <= 256u) sort_slice(std::integral_constant<int, 4>{});
  else sort_slice(std::integral_constant<int, 8>{});
  __builtin_amdgcn_wave_barrier();
  if (!active) return;
  // keep the first entry of every hash run (smallest slice index = smallest window id)
  uint32_t s = 0, hits = 0, cut_hashes = 0;
  for (uint32_t base = 0; base < n; base += 64) {
    const uint32_t i = base + lane;
    bool keep = false;
    uint32_t h = 0, idx = 0;
    if (i < n) {
      h = (uint32_t)(key[i] >> 16);
      idx = (uint32_t)(key[i] & 0xffffu);
      keep = (i == 0) || ((uint32_t)(key[i - 1] >> 16) != h);
    }
    const uint64_t bal = __ballot(keep);
    if (keep) {
      const uint32_t o = s + __popcll(bal & ((1ULL << lane) - 1ULL));
      // the minimizer's posting list as (first posting, length): the seeding kernels then go straight to the postings
      // instead of through two more dependent, uncoalesced reads of post_start per list
      uint32_t lo = 0, cnt = 0, mark = 0;  // (a hash the dictionary does not hold: an empty list)
      if (lookup) {  // the dictionary of the reference range only: by the hash's value
        const uint32_t mask = (1u << lookup_bits) - 1u;
        for (uint32_t slot = lookup_slot(h, lookup_bits);; slot = (slot + 1u) & mask) {
          const uint4 entry = lookup[slot];
          if (entry.y == 0xffffffffu) break;  // an empty slot: not there
          if (entry.x == h) { lo = entry.y; cnt = entry.z & 0x7fffffffu; mark = entry.z >> 31; break; }
        }
      } else {
        const uint32_t id_and_mark = mini_id[b0 + idx], id = id_and_mark & 0x7fffffffu;  // top bit: the hash lost seed hits to the frequency cut
        lo = post_start[id];
        cnt = post_start[id + 1] - lo;
        mark = id_and_mark >> 31;
      }
      q_hash[(uint64_t)f * kQMax + o] = h;
      q_pos[(uint64_t)f * kQMax + o] = cnt;
      q_id[(uint64_t)f * kQMax + o] = lo;
      hits += cnt;
      cut_hashes += mark;
    }
    s += __popcll(bal);
  }
  hits = wave_sum(hits);
  cut_hashes = wave_sum(cut_hashes);
  if (lane == 0) {
    q_s[f] = s;
    q_cut[f] = cut_hashes;
    hit_count[f] = hits;
    // running maxima of the batch; look first, most fragments do not raise them
    if (hits > __builtin_nontemporal_load(max_hits)) atomicMax(max_hits, hits);
    if (s > __builtin_nontemporal_load(max_hits + 1)) atomicMax(max_hits + 1, s);
  }
}

// ============================================================== 4. seed hits
__global__ __launch_bounds__(kThreads) void fill_hits_kernel(
    uint32_t n_frags, const uint32_t *__restrict__ q_pos, const uint32_t *__restrict__ q_id,
    const uint32_t *__restrict__ q_s, const uint32_t *__restrict__ hit_off, const uint32_t *__restrict__ post_start,
    const uint32_t *__restrict__ sorted_idx, const uint32_t *__restrict__ mini_wpos,
    const uint32_t *__restrict__ mini_contig, uint64_t *__restrict__ keys, uint32_t *__restrict__ vals) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t f = blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
  if (f >= n_frags) return;
  const uint32_t s = q_s[f];
  uint32_t base = hit_off[f];
  for (uint32_t i0 = 0; i0 < s; i0 += 64) {
    const uint32_t i = i0 + lane;
    uint32_t lo = 0, n = 0;
    if (i < s) {
      lo = q_id[(uint64_t)f * kQMax + i];  // (first posting, length) as query_sketch_kernel left them
      n = q_pos[(uint64_t)f * kQMax + i];
    }
    const uint32_t ex = wave_excl_scan(n, lane);
    for (uint32_t t = 0; t < n; ++t) {
      const uint32_t g = sorted_idx[lo + t];
      keys[base + ex + t] = ((uint64_t)f << 44) | ((uint64_t)mini_contig[g] << 24) | mini_wpos[g];
      vals[base + ex + t] = i;  // the rank of the hit's hash among the fragment's (the sketch is in hash order)
    }
    base += wave_sum(n);
  }
}

__global__ __launch_bounds__(kThreads) void segment_heads_kernel(const uint64_t *__restrict__ keys, uint32_t n,
                                                                 const uint32_t *__restrict__ contig_genome,
                                                                 uint32_t *__restrict__ flags) {
  const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  bool head = i == 0;
  if (!head) {
    const uint64_t a = keys[i - 1], b = keys[i];
    head = (a >> 44) != (b >> 44) || contig_genome[(a >> 24) & 0xfffffu] != contig_genome[(b >> 24) & 0xfffffu];
  }
  flags[i] = head ? 1u : 0u;
}
__global__ __launch_bounds__(kThreads) void segment_starts_kernel(const uint32_t *__restrict__ flags,
                                                                  const uint32_t *__restrict__ pos, uint32_t n,
                                                                  uint32_t *__restrict__ seg_start) {
  const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  if (flags[i]) seg_start[pos[i]] = i;
  if (i == n - 1) seg_start[pos[i] + flags[i]] = n;
}

// ---- hits of one fragment, bucketed by reference genome --------------------------------------------
// One wave per fragment and an LDS counter per reference genome: count the postings of the fragment's
// minimizers per genome (from the 2-byte genome of every posting: a list of ~15 is one 64-byte sector; the list
// bounds come coalesced from query_sketch_kernel), scan, then write every hit into its genome's slice of the
// fragment's hit range.
// The (fragment, genome) segments fall out of the scan, so the ones that can hold an L1 run (>= min_hits
// seed hits) are listed right here; nothing is sorted -- the mapping kernel orders the <= kHitCap hits of a
// segment in LDS, longer segments (repeats) are listed for segment_sort.  Eight lanes walk one posting
// list, so a wave reads 8 lists at a time in 64-byte pieces.
constexpr int kBucketWaves = 4;
constexpr uint32_t kHitRankShift = 55;  // a bucketed hit: rank of its hash in the fragment's sketch (9 bits) << 55 | contig << 24 | window id
__global__ __launch_bounds__(kBucketWaves * 64) void bucket_hits_kernel(
    uint32_t n_frags, const uint32_t *__restrict__ q_pos, const uint32_t *__restrict__ q_id,
    const uint32_t *__restrict__ q_s, const uint32_t *__restrict__ hit_off, const uint16_t *__restrict__ post_genome,
    const uint64_t *__restrict__ post_cw, uint32_t n_genomes, const uint32_t *__restrict__ tab_min_hits,
    uint64_t *__restrict__ keys, uint32_t *__restrict__ vals,
    uint32_t *__restrict__ seg_a0, uint32_t *__restrict__ seg_nh, uint32_t *__restrict__ seg_f, uint32_t seg_cap,
    uint32_t *__restrict__ counters, unsigned long long *__restrict__ cursor64, uint32_t ref0, uint32_t ref1, bool write_all) {
  extern __shared__ uint32_t bk_lds[];
  __shared__ uint32_t s_part[kBucketWaves][2], s_draw[2];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t f = blockIdx.x * kBucketWaves + wave;
  const bool active = f < n_frags;  // a wave past the last fragment has an empty sketch and still takes part in the barriers
  uint32_t *hist = bk_lds + (uint64_t)wave * n_genomes;
  const uint32_t s = active ? q_s[f] : 0u;
  const uint32_t base = active ? hit_off[f] : 0u;
  for (uint32_t g = lane; g < n_genomes; g += 64) hist[g] = 0;
  // The posting list of every minimizer of the fragment, (first posting, length), into registers: lane l holds the
  // lists of minimizers l, l + 64, ...  Two rounds of independent loads instead of a chain of three dependent ones
  // per group of eight lists.
  constexpr int kListRegs = kQMax / 64;
  uint32_t lo_r[kListRegs], n_r[kListRegs];
#pragma unroll
  for (int j = 0; j < kListRegs; ++j) {
    const uint32_t i = (uint32_t)j * 64u + lane;
    const bool in = i < s;
    lo_r[j] = in ? q_id[(uint64_t)f * kQMax + i] : 0u;  // coalesced: (first posting, length) from query_sketch_kernel
    n_r[j] = in ? q_pos[(uint64_t)f * kQMax + i] : 0u;
  }
  __builtin_amdgcn_wave_barrier();
  // Sixteen lists per step, 32 lanes' worth of slots each: a lane has eight independent posting loads in flight (the
  // kernel waits on memory 93 % of the time; what counts is how many loads are outstanding).  Lists longer than 32
  // (repeat families) take further rounds of the same step.
  constexpr int kLoads = 8;
  auto for_each_posting = [&](const auto *__restrict__ postings, auto &&visit) {
    using Elem = std::remove_cv_t<std::remove_reference_t<decltype(postings[0])>>;
    for (uint32_t i0 = 0; i0 < s; i0 += 2u * kLoads) {
      uint32_t lo_blk = 0, n_blk = 0;  // this lane's pair of the 64 lists i0 belongs to
#pragma unroll
      for (int j = 0; j < kListRegs; ++j)
        if ((i0 >> 6) == (uint32_t)j) { lo_blk = lo_r[j]; n_blk = n_r[j]; }
      uint32_t lo_u[kLoads], n_u[kLoads];
      uint32_t longest = 0;
#pragma unroll
      for (int u = 0; u < kLoads; ++u) {
        const uint32_t i = i0 + 2u * (uint32_t)u + (lane >> 5);  // i0 is a multiple of 16: the 16 lists share one block of 64
        lo_u[u] = __shfl(lo_blk, (int)(i & 63u), 64);
        n_u[u] = __shfl(n_blk, (int)(i & 63u), 64);  // 0 past the fragment's last minimizer; every lane takes part in the shuffle
        longest = max(longest, n_u[u]);
      }
      longest = pa_dev::wave_max_dpp(longest);
      for (uint32_t r = 0; r < longest; r += 32) {
        const uint32_t slot = r + (lane & 31u);
        Elem cw[kLoads];
#pragma unroll
        for (int u = 0; u < kLoads; ++u) cw[u] = slot < n_u[u] ? postings[lo_u[u] + slot] : Elem(0);
#pragma unroll
        for (int u = 0; u < kLoads; ++u)
          if (slot < n_u[u]) visit(i0 + 2u * (uint32_t)u + (lane >> 5), cw[u]);
      }
    }
  };
  // counting pass over the genome of every posting only (2 bytes each: a list of ~15 fits one 64-byte sector)
  for_each_posting(post_genome, [&](uint32_t, uint16_t g) { atomicAdd(&hist[g], 1u); });
  __builtin_amdgcn_wave_barrier();
  // exclusive scan over the genomes; then list the segments worth mapping.  The list cursors are global counters that
  // every workgroup of the launch draws from: ONE fetch-and-add per workgroup for both lists (atomics on one address
  // are served one after the other by the L2 -- with a draw per group of 64 genomes and wave, as this kernel had it,
  // 22 of its 23.7 ms per batch were spent queueing for that one address; profiles/r03_bucket_hits_ablation.txt).
  const uint32_t mh = s ? tab_min_hits[s] : 0xffffffffu;
  uint32_t carry = 0, n_small = 0, n_large = 0, n_big = 0, max_big = 0;
  for (uint32_t g0 = 0; g0 < n_genomes; g0 += 64) {
    const uint32_t g = g0 + lane;
    const uint32_t cnt = g < n_genomes ? hist[g] : 0u;
    const uint32_t off = carry + wave_excl_scan(cnt, lane);
    const bool keep = cnt >= mh && g >= ref0 && g < ref1;  // only the reference genomes asked for are mapped
    // the top bit marks a genome whose hits nobody will read (fewer than a run needs, or not asked for): the scatter
    // pass then does not write them -- lone 8-byte stores, a 64-byte sector each (offsets stay below 2^31)
    if (g < n_genomes) hist[g] = off | (keep ? 0u : 0x80000000u);
    const bool small = keep && cnt <= (uint32_t)kHitCapSmall;
    n_small += (uint32_t)__popcll(__ballot(small));
    n_large += (uint32_t)__popcll(__ballot(keep && !small));
    n_big += (uint32_t)__popcll(__ballot(keep && cnt > (uint32_t)kHitCap));
    if (keep && cnt > (uint32_t)kHitCap) max_big = max(max_big, cnt);
    carry += wave_sum(cnt);
  }
  const uint32_t total_hits = carry;
  if (n_big) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) max_big = max(max_big, (uint32_t)__shfl_xor((int)max_big, o, 64));
  }
  if (lane == 0) {
    s_part[wave][0] = n_small;
    s_part[wave][1] = n_large;
    if (n_big) { atomicAdd(&counters[1], n_big); atomicMax(&counters[2], max_big); }
  }
  __syncthreads();
  if (threadIdx.x == 0) {  // both cursors in one 64-bit word: one draw per workgroup
    uint32_t ws = 0, wl = 0;
#pragma unroll
    for (int q = 0; q < kBucketWaves; ++q) { ws += s_part[q][0]; wl += s_part[q][1]; }
    unsigned long long got = 0;
    if (ws | wl) got = atomicAdd(cursor64, ((unsigned long long)wl << 32) | ws);
    s_draw[0] = (uint32_t)got;
    s_draw[1] = (uint32_t)(got >> 32);
  }
  __syncthreads();
  uint32_t s0 = s_draw[0], l0 = s_draw[1];
#pragma unroll
  for (int q = 0; q < kBucketWaves; ++q)
    if ((uint32_t)q < wave) { s0 += s_part[q][0]; l0 += s_part[q][1]; }
  if (n_small | n_large) {
    // short segments fill the list from the front, the ones over kHitCapSmall hits from the back: the two
    // classes are mapped by launches with different LDS footprints
    for (uint32_t g0 = 0; g0 < n_genomes; g0 += 64) {
      const uint32_t g = g0 + lane;
      uint32_t off = 0, cnt = 0;
      if (g < n_genomes) {
        off = hist[g] & 0x7fffffffu;
        cnt = (g + 1 < n_genomes ? (hist[g + 1] & 0x7fffffffu) : total_hits) - off;
      }
      const bool keep = cnt >= mh && g >= ref0 && g < ref1;
      const bool small = keep && cnt <= (uint32_t)kHitCapSmall, large = keep && !small;
      const uint64_t sm = __ballot(small), lm = __ballot(large);
      if (keep) {
        const uint64_t below = (1ULL << lane) - 1ULL;
        const uint32_t slot = small ? s0 + (uint32_t)__popcll(sm & below) : seg_cap - 1u - (l0 + (uint32_t)__popcll(lm & below));
        if (slot < seg_cap) { seg_a0[slot] = base + off; seg_nh[slot] = cnt; seg_f[slot] = f; }
      }
      s0 += (uint32_t)__popcll(sm);
      l0 += (uint32_t)__popcll(lm);
    }
  }
  __builtin_amdgcn_wave_barrier();
  // A hit carries the RANK of its hash among the fragment's hashes -- the sketch is in hash order, so that is the number i
  // of the posting list it comes from -- on top of its (contig, window id): the mapping kernel's bound on what a window
  // can share asks which of a window's hits lie below a pivot rank (kHitRankShift; the fragment itself is named by the
  // segment lists).
  // write_all: the batch is about to be ordered as a whole (a repeat family too long for an LDS sort), and that keeps
  // the (fragment, genome) slices in place only when every slot holds its own key with the fragment on top; the rank
  // then rides in the sort's payload.
  for_each_posting(post_cw, [&](uint32_t i, uint64_t cw) {
    const uint32_t at = atomicAdd(&hist[(uint32_t)(cw >> 44)], 1u);
    const uint64_t cw44 = cw & ((1ULL << 44) - 1ULL);
    if (write_all) {
      keys[base + (at & 0x7fffffffu)] = ((uint64_t)f << 44) | cw44;
      vals[base + (at & 0x7fffffffu)] = i;
    } else if (!(at & 0x80000000u)) {
      keys[base + at] = ((uint64_t)i << kHitRankShift) | cw44;
    }
  });
}

// ---- the same for one fragment per WORKGROUP, the hits of the listed pairs staged in LDS ----------------------------
// bucket_hits_kernel above writes every hit of a listed pair with a lone 8-byte store into its genome's slice of the
// fragment's hit range: the eight hits of a 64-byte sector arrive at eight different times from eight posting lists, the
// L2 cannot hold the lines of all fragments in flight until they are full, and the counters show 19.6 written bytes per
// hit against 8 (profiles/r04_pmc_bucket_hits_summary.txt).  Here the eight waves of a workgroup share one fragment:
// they split its posting lists (a step of sixteen lists per wave and turn), count into ONE histogram in LDS, place the
// hits of the LISTED genomes in a compact LDS copy of their slices (21 KB on average at 1 000 genomes: the pairs that
// are not listed -- fewer hits than a run needs -- are not written at all), and then every listed slice leaves as a run
// of consecutive 8-byte stores -- whole sectors but for a slice's two ends.  A fragment whose listed hits do not fit the
// staging area (repeat families) writes them directly, as above.
constexpr int kStageWaves = 8;
__global__ __launch_bounds__(kStageWaves * 64) void bucket_hits_staged_kernel(
    uint32_t n_frags, const uint32_t *__restrict__ q_pos, const uint32_t *__restrict__ q_id,
    const uint32_t *__restrict__ q_s, const uint32_t *__restrict__ hit_off,
    const uint16_t *__restrict__ post_genome, const uint64_t *__restrict__ post_cw, uint32_t n_genomes,
    const uint32_t *__restrict__ tab_min_hits, uint64_t *__restrict__ keys, uint32_t *__restrict__ seg_a0,
    uint32_t *__restrict__ seg_nh, uint32_t *__restrict__ seg_f, uint32_t seg_cap, uint32_t *__restrict__ counters,
    unsigned long long *__restrict__ cursor64, uint32_t ref0, uint32_t ref1, uint32_t stage_cap) {
  extern __shared__ uint64_t st_lds[];
  uint64_t *stage = st_lds;                                           // [stage_cap] the listed slices, one after the other
  uint32_t *hist = reinterpret_cast<uint32_t *>(st_lds + stage_cap);  // [n_genomes] counts, then offsets in the fragment's hit range
  uint32_t *coff = hist + n_genomes;                                  // [n_genomes] cursors into `stage` (listed genomes)
  __shared__ uint32_t s_tot[kStageWaves][2], s_part[kStageWaves][2], s_draw[2];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint32_t f = blockIdx.x;
  if (f >= n_frags) return;
  const uint32_t s = q_s[f], base = hit_off[f];
  for (uint32_t g = tid; g < n_genomes; g += kStageWaves * 64) hist[g] = 0;
  __syncthreads();
  // sixteen lists per step and wave, 32 lanes' worth of slots each, eight independent posting loads per lane in flight
  constexpr int kLoads = 8;
  auto for_each_posting = [&](const auto *__restrict__ postings, auto &&visit) {
    using Elem = std::remove_cv_t<std::remove_reference_t<decltype(postings[0])>>;
    for (uint32_t i0 = 2u * kLoads * wave; i0 < s; i0 += 2u * kLoads * kStageWaves) {
      uint32_t lo_u[kLoads], n_u[kLoads], longest = 0;
#pragma unroll
      for (int u = 0; u < kLoads; ++u) {
        const uint32_t i = i0 + 2u * (uint32_t)u + (lane >> 5);
        lo_u[u] = i < s ? q_id[(uint64_t)f * kQMax + i] : 0u;  // (first posting, length) from query_sketch_kernel
        n_u[u] = i < s ? q_pos[(uint64_t)f * kQMax + i] : 0u;
        longest = max(longest, n_u[u]);
      }
      longest = pa_dev::wave_max_dpp(longest);
      for (uint32_t r = 0; r < longest; r += 32) {
        const uint32_t slot = r + (lane & 31u);
        Elem cw[kLoads];
#pragma unroll
        for (int u = 0; u < kLoads; ++u) cw[u] = slot < n_u[u] ? postings[lo_u[u] + slot] : Elem(0);
#pragma unroll
        for (int u = 0; u < kLoads; ++u)
          if (slot < n_u[u]) visit(i0 + 2u * (uint32_t)u + (lane >> 5), cw[u]);
      }
    }
  };
  for_each_posting(post_genome, [&](uint32_t, uint16_t g) { atomicAdd(&hist[g], 1u); });
  __syncthreads();
  // exclusive scans over the genomes, 512 at a time (a wave per 64 genomes, the waves chained through their totals): of all
  // counts -- the offsets in the fragment's hit range -- and of the listed genomes' counts -- the offsets in `stage`
  const uint32_t mh = s ? tab_min_hits[s] : 0xffffffffu;
  uint32_t carry = 0, ccarry = 0, n_small = 0, n_large = 0, n_big = 0, max_big = 0;
  for (uint32_t g0 = 0; g0 < n_genomes; g0 += kStageWaves * 64) {
    const uint32_t g = g0 + tid;
    const uint32_t cnt = g < n_genomes ? hist[g] : 0u;
    const bool keep = cnt >= mh && g >= ref0 && g < ref1;
    const uint32_t kcnt = keep ? cnt : 0u;
    const uint32_t ex = wave_excl_scan(cnt, lane), wsum = wave_sum(cnt);
    const uint32_t kex = wave_excl_scan(kcnt, lane), kwsum = wave_sum(kcnt);
    if (lane == 0) { s_tot[wave][0] = wsum; s_tot[wave][1] = kwsum; }
    __syncthreads();
    uint32_t before = 0, chunk = 0, kbefore = 0, kchunk = 0;
#pragma unroll
    for (int q = 0; q < kStageWaves; ++q) {
      const uint32_t t = s_tot[q][0], kt = s_tot[q][1];
      before += (uint32_t)q < wave ? t : 0u; chunk += t;
      kbefore += (uint32_t)q < wave ? kt : 0u; kchunk += kt;
    }
    if (g < n_genomes) {
      hist[g] = (carry + before + ex) | (keep ? 0u : 0x80000000u);  // top bit: nobody will read this genome's hits
      coff[g] = ccarry + kbefore + kex;
    }
    const bool small = keep && cnt <= (uint32_t)kHitCapSmall;
    n_small += (uint32_t)__popcll(__ballot(small));
    n_large += (uint32_t)__popcll(__ballot(keep && !small));
    n_big += (uint32_t)__popcll(__ballot(keep && cnt > (uint32_t)kHitCap));
    if (keep && cnt > (uint32_t)kHitCap) max_big = max(max_big, cnt);
    carry += chunk;
    ccarry += kchunk;
    __syncthreads();  // s_tot is written again in the next turn
  }
  const uint32_t total_hits = carry;
  const bool staged = ccarry <= stage_cap;  // the listed hits fit the staging area (uniform over the workgroup)
  if (n_big) max_big = pa_dev::wave_max_dpp(max_big);
  if (lane == 0) {
    s_part[wave][0] = n_small;
    s_part[wave][1] = n_large;
    if (n_big) { atomicAdd(&counters[1], n_big); atomicMax(&counters[2], max_big); }
  }
  __syncthreads();
  if (tid == 0) {  // both list cursors in one 64-bit word: one draw per workgroup
    uint32_t ws = 0, wl = 0;
#pragma unroll
    for (int q = 0; q < kStageWaves; ++q) { ws += s_part[q][0]; wl += s_part[q][1]; }
    unsigned long long got = 0;
    if (ws | wl) got = atomicAdd(cursor64, ((unsigned long long)wl << 32) | ws);
    s_draw[0] = (uint32_t)got;
    s_draw[1] = (uint32_t)(got >> 32);
  }
  __syncthreads();
  uint32_t s0 = s_draw[0], l0 = s_draw[1];
#pragma unroll
  for (int q = 0; q < kStageWaves; ++q)
    if ((uint32_t)q < wave) { s0 += s_part[q][0]; l0 += s_part[q][1]; }
  if (n_small | n_large) {  // (uniform over the wave)
    for (uint32_t g0 = 0; g0 < n_genomes; g0 += kStageWaves * 64) {
      const uint32_t g = g0 + tid;
      uint32_t off = 0, cnt = 0;
      if (g < n_genomes) {
        off = hist[g] & 0x7fffffffu;
        cnt = (g + 1 < n_genomes ? (hist[g + 1] & 0x7fffffffu) : total_hits) - off;
      }
      const bool keep = cnt >= mh && g >= ref0 && g < ref1;
      const bool small = keep && cnt <= (uint32_t)kHitCapSmall, large = keep && !small;
      const uint64_t sm = __ballot(small), lm = __ballot(large);
      if (keep) {
        const uint64_t below = (1ULL << lane) - 1ULL;
        const uint32_t slot = small ? s0 + (uint32_t)__popcll(sm & below) : seg_cap - 1u - (l0 + (uint32_t)__popcll(lm & below));
        if (slot < seg_cap) { seg_a0[slot] = base + off; seg_nh[slot] = cnt; seg_f[slot] = f; }
      }
      s0 += (uint32_t)__popcll(sm);
      l0 += (uint32_t)__popcll(lm);
    }
  }
  // scatter pass.  Staged: a hit of a listed genome draws its place from the genome's cursor into `stage` (the offsets
  // in the hit range stay as they are: the copy below reads them); else from the genome's offset, which moves.
  if (staged) {
    for_each_posting(post_cw, [&](uint32_t i, uint64_t cw) {
      const uint32_t g = (uint32_t)(cw >> 44);
      if (!(hist[g] & 0x80000000u)) stage[atomicAdd(&coff[g], 1u)] = ((uint64_t)i << kHitRankShift) | (cw & ((1ULL << 44) - 1ULL));
    });
    __syncthreads();
    // the listed slices, from LDS to the hit array: a wave per genome, consecutive lanes consecutive hits.  A genome's
    // cursor stands at the END of its slice in `stage` now; its length is the distance to the next genome's offset.
    for (uint32_t g0 = 0; g0 < n_genomes; g0 += kStageWaves * 64) {
      const uint32_t g = g0 + tid;
      uint32_t off = 0, cnt = 0, cend = 0;
      bool keep = false;
      if (g < n_genomes) {
        const uint32_t v = hist[g];
        keep = !(v & 0x80000000u);
        off = v & 0x7fffffffu;
        cnt = (g + 1 < n_genomes ? (hist[g + 1] & 0x7fffffffu) : total_hits) - off;
        cend = coff[g];
      }
      for (uint64_t todo = __ballot(keep && cnt > 0u); todo; todo &= todo - 1) {
        const int src = __builtin_ctzll(todo);
        const uint32_t o0 = (uint32_t)__builtin_amdgcn_readlane((int)off, src), n0 = (uint32_t)__builtin_amdgcn_readlane((int)cnt, src);
        const uint32_t c0 = (uint32_t)__builtin_amdgcn_readlane((int)cend, src) - n0;
        for (uint32_t i = lane; i < n0; i += 64) keys[base + o0 + i] = stage[c0 + i];
      }
    }
  } else {
    __syncthreads();  // every offset has been read before the scatter pass moves it
    for_each_posting(post_cw, [&](uint32_t i, uint64_t cw) {
      const uint32_t at = atomicAdd(&hist[(uint32_t)(cw >> 44)], 1u);
      if (!(at & 0x80000000u)) keys[base + at] = ((uint64_t)i << kHitRankShift) | (cw & ((1ULL << 44) - 1ULL));
    });
  }
}

// Chance matches with unrelated genomes still leave millions of listed segments with a handful of hits and
// no valid L1 run.  One THREAD settles each segment of <= 8 hits here (sort its keys in registers, test the
// run condition of map_segments_kernel exactly) so that the mapping kernel does not spend a workgroup
// launch and a dozen dependent loads on it; longer segments pass through.
constexpr uint32_t kTinySegment = 8;
__global__ __launch_bounds__(kThreads) void prefilter_segments_kernel(
    const uint64_t *__restrict__ keys, const uint32_t *__restrict__ seg_a0, const uint32_t *__restrict__ seg_nh,
    const uint32_t *__restrict__ seg_f, uint32_t n_segs, const uint32_t *__restrict__ q_s,
    const uint32_t *__restrict__ tab_min_hits, const uint32_t *__restrict__ q_cut, uint32_t frag_len, uint32_t *__restrict__ out_a0,
    uint32_t *__restrict__ out_nh, uint32_t *__restrict__ out_f, unsigned long long *__restrict__ counter, uint32_t sparse_max) {
  // Kept segments of at most `sparse_max` hits (0: none) whose fragment lost no hash to the frequency cut are listed from
  // the BACK of the out arrays: map_sparse_kernel takes them, map_segments_kernel the ones listed from the front.
  const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t a0 = 0, nh = 0, f = 0;
  bool keep = false;
  if (i < n_segs) {
    a0 = seg_a0[i];
    nh = seg_nh[i];
    f = seg_f[i];
    if (nh > kTinySegment) {
      keep = true;
    } else {
      uint64_t k[kTinySegment];
#pragma unroll
      for (uint32_t j = 0; j < kTinySegment; ++j) k[j] = j < nh ? (keys[a0 + j] & ((1ULL << 44) - 1ULL)) : ~0ULL;  // (contig, window id)
#pragma unroll
      for (uint32_t pass = 0; pass < kTinySegment; ++pass)  // odd-even transposition sort: 8 passes sort 8 keys
#pragma unroll
        for (uint32_t j = pass & 1u; j + 1 < kTinySegment; j += 2) {
          const uint64_t lo = k[j] < k[j + 1] ? k[j] : k[j + 1], hi = k[j] < k[j + 1] ? k[j + 1] : k[j];
          k[j] = lo;
          k[j + 1] = hi;
        }
      const uint32_t s = q_s[f];
      const uint32_t mh = s ? tab_min_hits[s] : 0xffffffffu;
#pragma unroll
      for (uint32_t a = 0; a < kTinySegment; ++a) {
#pragma unroll
        for (uint32_t z = a; z < kTinySegment; ++z) {  // z = a + mh - 1
          if (z + 1 == a + mh && z < nh && ((k[a] >> 24) & 0xfffffu) == ((k[z] >> 24) & 0xfffffu) &&
              (uint32_t)(k[z] & 0xffffffu) - (uint32_t)(k[a] & 0xffffffu) < frag_len)
            keep = true;
        }
      }
    }
  }
  const bool sparse = keep && nh <= sparse_max && q_cut[f] == 0u;
  // one draw from the two list cursors (one 64-bit word) per workgroup (same-address atomics queue up in the L2, ~11 ns each)
  __shared__ uint32_t s_kept[kThreads / 64][2], s_base[2];
  const uint64_t km = __ballot(keep && !sparse), sm = __ballot(sparse);
  const uint32_t wave = threadIdx.x >> 6;
  if (lane == 0) { s_kept[wave][0] = (uint32_t)__popcll(km); s_kept[wave][1] = (uint32_t)__popcll(sm); }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t sum = 0, ssum = 0;
#pragma unroll
    for (int q = 0; q < kThreads / 64; ++q) { sum += s_kept[q][0]; ssum += s_kept[q][1]; }
    const unsigned long long got = (sum | ssum) ? atomicAdd(counter, ((unsigned long long)ssum << 32) | sum) : 0ull;
    s_base[0] = (uint32_t)got;
    s_base[1] = (uint32_t)(got >> 32);
  }
  __syncthreads();
  if (keep) {
    uint32_t slot = s_base[sparse ? 1 : 0] + (uint32_t)__popcll((sparse ? sm : km) & ((1ULL << lane) - 1ULL));
#pragma unroll
    for (int q = 0; q < kThreads / 64; ++q)
      if ((uint32_t)q < wave) slot += s_kept[q][sparse ? 1 : 0];
    if (sparse) slot = n_segs - 1u - slot;  // from the back (n_segs slots in all: the two lists cannot meet)
    out_a0[slot] = a0;
    out_nh[slot] = nh;
    out_f[slot] = f;
  }
}

// the listed segments longer than kHitCap, as (start, length) pairs for frag_sort_kernel
__global__ __launch_bounds__(kThreads) void big_segments_kernel(const uint32_t *__restrict__ seg_a0,
                                                                const uint32_t *__restrict__ seg_nh, uint32_t n_segs,
                                                                uint32_t *__restrict__ big_a0,
                                                                uint32_t *__restrict__ big_nh,
                                                                uint32_t *__restrict__ counter) {
  const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= n_segs || seg_nh[i] <= (uint32_t)kHitCap) return;
  const uint32_t slot = atomicAdd(counter, 1u);
  big_a0[slot] = seg_a0[i];
  big_nh[slot] = seg_nh[i];
}

// ---- hits of one fragment, sorted by (contig, window) in LDS --------------------------------------
// fill_hits_kernel leaves the hits of fragment f contiguous at hit_off[f]; they only need ordering
// inside the fragment (the fragment number is the top of the key), so one workgroup sorts one fragment
// in LDS (bitonic on the 64-bit key with its 32-bit payload) instead of 8 radix passes over the
// whole batch.  Used when the busiest fragment has <= kFragSortMax hits.
constexpr uint32_t kFragSortMax = 8192;
constexpr int kFragSortThreads = 256;

__global__ __launch_bounds__(kFragSortThreads) void frag_sort_kernel(uint64_t *__restrict__ keys,
                                                                     uint32_t *__restrict__ vals,
                                                                     const uint32_t *__restrict__ hit_off,
                                                                     const uint32_t *__restrict__ hit_count,
                                                                     uint32_t np2_max, uint32_t rot) {
  // rot: bits the keys are turned left by while they are sorted -- 0 for keys with the fragment on top (one fragment's
  // hits: by contig and window id), 20 for the bucketed hits of one long segment, whose top bits hold the rank of the
  // hit's hash: ordered by (contig, window id) all the same
  extern __shared__ uint64_t fs_key[];
  uint32_t *fs_val = reinterpret_cast<uint32_t *>(fs_key + np2_max);
  const uint32_t f = blockIdx.x, tid = threadIdx.x;
  const uint32_t n = hit_count[f];
  if (n < 2) return;
  const uint32_t a0 = hit_off[f];
  uint32_t np2 = 2;
  while (np2 < n) np2 <<= 1;
  for (uint32_t i = tid; i < np2; i += kFragSortThreads) {
    const uint64_t key = i < n ? keys[a0 + i] : 0ULL;
    fs_key[i] = i < n ? (rot ? (key << rot) | (key >> (64u - rot)) : key) : ~0ULL;
    fs_val[i] = i < n ? vals[a0 + i] : 0u;
  }
  __syncthreads();
  for (uint32_t k = 2; k <= np2; k <<= 1) {
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      for (uint32_t t = tid; t < (np2 >> 1); t += kFragSortThreads) {
        const uint32_t i = 2u * t - (t & (j - 1u));
        const uint32_t l = i + j;
        const uint64_t a = fs_key[i], b = fs_key[l];
        const bool up = (i & k) == 0u;
        if ((a > b) == up && a != b) {
          fs_key[i] = b;
          fs_key[l] = a;
          const uint32_t va = fs_val[i];
          fs_val[i] = fs_val[l];
          fs_val[l] = va;
        }
      }
      __syncthreads();
    }
  }
  for (uint32_t i = tid; i < n; i += kFragSortThreads) {
    const uint64_t key = fs_key[i];
    keys[a0 + i] = rot ? (key >> rot) | (key << (64u - rot)) : key;
    vals[a0 + i] = fs_val[i];
  }
}

// keep the segments that can hold an L1 run at all: at least min_hits(s) seed hits
__global__ __launch_bounds__(kThreads) void segment_keep_kernel(const uint64_t *__restrict__ keys,
                                                                const uint32_t *__restrict__ seg_start, uint32_t n_segs,
                                                                const uint32_t *__restrict__ q_s,
                                                                const uint32_t *__restrict__ tab_min_hits,
                                                                const uint32_t *__restrict__ contig_genome, uint32_t ref0,
                                                                uint32_t ref1, uint32_t *__restrict__ keep) {
  const uint32_t seg = blockIdx.x * kThreads + threadIdx.x;
  if (seg >= n_segs) return;
  const uint32_t a0 = seg_start[seg], nh = seg_start[seg + 1] - a0;
  const uint64_t key = keys[a0];
  const uint32_t s = q_s[(uint32_t)(key >> 44)];
  const uint32_t g = contig_genome[(key >> 24) & 0xfffffu];
  keep[seg] = (s != 0 && nh >= tab_min_hits[s] && g >= ref0 && g < ref1) ? 1u : 0u;
}

__global__ __launch_bounds__(kThreads) void segment_list_kernel(const uint32_t *__restrict__ keep,
                                                                const uint32_t *__restrict__ pos, uint32_t n_segs,
                                                                uint32_t *__restrict__ seg_list) {
  const uint32_t seg = blockIdx.x * kThreads + threadIdx.x;
  if (seg < n_segs && keep[seg]) seg_list[pos[seg]] = seg;
}

// old path: the kept segments of the fully sorted hit list as (start, length) pairs
__global__ __launch_bounds__(kThreads) void segments_from_list_kernel(const uint32_t *__restrict__ seg_start,
                                                                      const uint32_t *__restrict__ seg_list,
                                                                      uint32_t n, uint32_t *__restrict__ seg_a0,
                                                                      uint32_t *__restrict__ seg_nh) {
  const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  const uint32_t seg = seg_list[i];
  seg_a0[i] = seg_start[seg];
  seg_nh[i] = seg_start[seg + 1] - seg_start[seg];
}

// Reference minimizers of one stretch (the windows of up to 64 starts) held in LDS: a compile-time parameter of the
// mapping kernel, chosen per call from the expected minimizers per window (map_ref_cap) -- 320 for fastANI's defaults
// (237 per window): with 512 the per-lane arrays cost registers and waves (1.04 s against 0.99 s for the 1 000-genome
// run), with 256 most windows no longer fit and take the cooperative path (2.9 s).
constexpr uint32_t kRefCapMax = 512;
// LDS of one segment's wave, carved from dynamic shared memory so that the query-hash arrays are only as
// long as the longest fragment sketch of the batch (s_cap): ~8 KB per wave instead of 17 KB, which is what
// sets how many of these latency-bound waves a CU keeps in flight.
struct EvalShared {
  uint32_t *qh;        // [s_cap] the fragment's sketch, ascending
  uint32_t *cnt;       // [s_cap + 64] reference-only hashes per rank gap (cooperative evaluation of one over-long window);
  uint32_t *tab;       //   the same memory: the bit tables of the windowed evaluation (eval_tab_words)
  uint32_t *matched;   // [kQMax / 32] bitset over query ranks (cooperative evaluation of one over-long window)
  uint32_t *cand;      // [4 * 8] candidates of the L1 scan waiting for their evaluation: contig, first and last start, first hit
  uint32_t *scan;      // [24] the L1 scan's state while the candidates it has listed are evaluated [0..10], the best mapping so far [16..23]
  uint32_t *lmask;     // [kLmaskWords] one bit per staged hit: the rank of its hash lies below the pivot of the tight bound;
  uint32_t *lpre;      // [kLmaskWords] set bits in the words before (so "hits below the pivot among the first i" is two reads)
  uint32_t *hw;        // [kHitCap] window id of each staged hit << 8 | rank of its hash in the fragment's sketch >> 1
  uint16_t *hc;        // [kHitCap] contig of each staged hit, relative to the segment's first
  uint16_t *ref_w;     // [kRefCap] window id relative to the first minimizer's of the stretch
  uint16_t *prev;      // [kRefCap] 1 + stretch position of the same hash earlier in the stretch, 0: none
  uint16_t *qt;        // [kQtBuckets] the fragment's sketch bucketed by the top bits of the hash: first rank (10 bits) | hashes in the bucket (6 bits)
};
constexpr uint32_t kQtBits = 9, kQtBuckets = 1u << kQtBits, kQtShift = 32u - kQtBits;
constexpr uint32_t kLmaskWords = (uint32_t)kHitCap / 32u + 2u;  // a word per 32 staged hits, one more for "all of them", even
// Bit tables of the windowed evaluation: one row per query rank r = the stretch positions (one bit each, kRefCap / 32
// words) whose minimizer has rank <= r among the fragment's hashes.  Coarse rows stand at every kCoarse-th rank; the
// fine rows cover kFineGroups coarse groups at a time, every rank of them.
constexpr uint32_t kCoarseShift = 4, kCoarse = 1u << kCoarseShift;
constexpr uint32_t kFineGroups = 2, kFineRows = kFineGroups * kCoarse + 1u;
__host__ __device__ inline uint32_t eval_tab_words(uint32_t s_cap, uint32_t ref_cap) {
  const uint32_t t = (s_cap / kCoarse + 1u + kFineRows) * (ref_cap / 16u), c = s_cap + 64u;  // coarse and fine rows of two halves
  return ((t > c ? t : c) + 3u) & ~3u;
}
__host__ __device__ inline uint32_t eval_lds_bytes(uint32_t s_cap, uint32_t hit_cap, uint32_t ref_cap) {
  return 4u * s_cap + 4u * eval_tab_words(s_cap, ref_cap) + 4u * (kQMax / 32) + 128u + 96u + 8u * kLmaskWords + hit_cap * 6u + ref_cap * 4u +
         2u * kQtBuckets;
}
// The arrays whose length is known at compile time come first, so that their addresses are constants of the kernel
// (immediate offsets of the LDS instructions, no registers); the fragment's sketch and the tables follow.
__device__ __forceinline__ EvalShared eval_carve(uint32_t *base, uint32_t s_cap, uint32_t hit_cap, uint32_t kRefCap) {
  EvalShared sh;
  sh.matched = base;
  sh.cand = sh.matched + kQMax / 32;
  sh.scan = sh.cand + 32;
  sh.lmask = sh.scan + 24;
  sh.lpre = sh.lmask + kLmaskWords;
  sh.qt = reinterpret_cast<uint16_t *>(sh.lpre + kLmaskWords);  // (2 kLmaskWords is a multiple of 4: still on a 16-byte boundary)
  sh.hw = reinterpret_cast<uint32_t *>(sh.qt + kQtBuckets);
  sh.hc = reinterpret_cast<uint16_t *>(sh.hw + hit_cap);
  sh.ref_w = sh.hc + hit_cap;  // hit_cap and kRefCap are multiples of 64: everything stays on 16-byte boundaries
  sh.prev = sh.ref_w + kRefCap;
  sh.qh = reinterpret_cast<uint32_t *>(sh.prev + kRefCap);
  sh.cnt = sh.qh + s_cap;  // s_cap is a multiple of 64: the tables start on a 16-byte boundary
  sh.tab = sh.cnt;
  return sh;
}

// fetch-and-add on a 16-bit LDS counter through a 32-bit atomic on the word that holds it (little-endian halves;
// the counts stay below 2^16, so the low half never carries into the high one)
__device__ __forceinline__ uint32_t atomicAdd_u16(uint16_t *counters, uint32_t idx) {
  uint32_t *word = reinterpret_cast<uint32_t *>(counters) + (idx >> 1);
  const uint32_t old = atomicAdd(word, (idx & 1u) ? 0x10000u : 1u);
  return (idx & 1u) ? (old >> 16) : (old & 0xffffu);
}

// One record per listed segment: what a mapping wave needs before it can load anything else -- where the hits are, the
// fragment, its sketch size, the seed hits a run needs, the first contig of the reference genome -- gathered by one
// thread per segment, so that the wave starts with one scalar load instead of a chain of five dependent ones.
// Two uint4: {first hit, hits, fragment, sketch size} {hits a run needs, first contig, hashes of the sketch that lost seed hits to the
// frequency cut, 0}; sketch size 0 = nothing to do.
__global__ __launch_bounds__(kThreads) void segment_records_kernel(
    const uint64_t *__restrict__ keys, const uint32_t *__restrict__ seg_a0, const uint32_t *__restrict__ seg_nh,
    const uint32_t *__restrict__ seg_f /* null: the fragment is the top of the key */, uint32_t n_segs,
    const uint32_t *__restrict__ q_s, const uint32_t *__restrict__ q_cut, const uint32_t *__restrict__ tab_min_hits,
    const uint32_t *__restrict__ contig_genome, const uint32_t *__restrict__ genome_first_contig, uint4 *__restrict__ rec) {
  const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= n_segs) return;
  const uint32_t a0 = seg_a0[i], nh = seg_nh[i];
  const uint64_t key = keys[a0];
  const uint32_t f = seg_f ? seg_f[i] : (uint32_t)(key >> 44);
  uint32_t s = q_s[f];
  const uint32_t mh = s ? tab_min_hits[s] : 0u;
  if (nh < mh) s = 0;  // chance matches with unrelated genomes: fewer seed hits than any L1 run needs
  const uint32_t hc_base = genome_first_contig[contig_genome[(uint32_t)(key >> 24) & 0xfffffu]];
  rec[2 * (uint64_t)i] = make_uint4(a0, nh, f, s);
  rec[2 * (uint64_t)i + 1] = make_uint4(mh, hc_base, q_cut[f], 0u);
}

// the segment's hits from the hit array into LDS, in order: window id and rank (see EvalShared::hw), contig relative to
// the genome's first.  The hits come as the bucketing pass left them: rank << kHitRankShift | contig << 24 | window id.
template <int E>
__device__ __forceinline__ void stage_hits_sorted(const uint64_t *__restrict__ seg_keys, uint32_t nh, uint32_t hc_base,
                                                  uint32_t lane, uint32_t *hw, uint16_t *hc) {
  uint64_t k[E];
  uint32_t widest = 0, furthest = 0;  // largest relative contig and largest window id of the segment
#pragma unroll
  for (int q = 0; q < E; ++q) {
    const uint32_t e = lane * (uint32_t)E + (uint32_t)q;
    const uint64_t raw = e < nh ? seg_keys[e] : 0ULL;
    // sort key: (contig, window id) on top of the rank -- a (contig, window id) occurs once, the rank decides nothing
    k[q] = e < nh ? ((raw & 0xfffffffffffULL) << 9) | (raw >> kHitRankShift) : ~0ULL;
    if (e < nh) {
      widest = max(widest, ((uint32_t)(raw >> 24) & 0xfffffu) - hc_base);
      furthest = max(furthest, (uint32_t)raw & 0xffffffu);
    }
  }
  widest = pa_dev::wave_max_dpp(widest);
  furthest = pa_dev::wave_max_dpp(furthest);
  // 32-bit keys where contig, window id and rank fit (uniform): one shuffle, a minimum, a maximum and a select per key and
  // stage.  A genome of one contig of up to 8 Mb always does.
  const uint32_t w_bits = 32u - (uint32_t)__builtin_clz(furthest | 1u);
  if (((uint64_t)(widest + 1u) << w_bits) <= (1ull << 23)) {
    uint32_t k32[E];
#pragma unroll
    for (int q = 0; q < E; ++q) {
      const uint32_t e = lane * (uint32_t)E + (uint32_t)q;
      k32[q] = 0xffffffffu;
      if (e < nh) {
        const uint32_t c_rel = ((uint32_t)(k[q] >> 33) & 0xfffffu) - hc_base, wpos = (uint32_t)(k[q] >> 9) & 0xffffffu;
        k32[q] = ((((c_rel << w_bits) | wpos)) << 9) | ((uint32_t)k[q] & 0x1ffu);
      }
    }
    bitonic_sort_lanes<E, uint32_t>(k32, lane);
#pragma unroll
    for (int q = 0; q < E; ++q) {
      const uint32_t e = lane * (uint32_t)E + (uint32_t)q;
      if (e < nh) {
        const uint32_t cw = k32[q] >> 9;
        hw[e] = ((cw & ((1u << w_bits) - 1u)) << 8) | ((k32[q] & 0x1ffu) >> 1);
        hc[e] = (uint16_t)(cw >> w_bits);
      }
    }
    return;
  }
  bitonic_sort_lanes<E, uint64_t>(k, lane);
#pragma unroll
  for (int q = 0; q < E; ++q) {
    const uint32_t e = lane * (uint32_t)E + (uint32_t)q;
    if (e < nh) {
      hw[e] = (((uint32_t)(k[q] >> 9) & 0xffffffu) << 8) | (((uint32_t)k[q] & 0x1ffu) >> 1);
      hc[e] = (uint16_t)(((uint32_t)(k[q] >> 33) & 0xfffffu) - hc_base);
    }
  }
}

// The phases of map_segments_kernel in the order they run.  In the tools build the kernel takes one of these as an argument
// and ends after that phase (or leaves a part out): tools/map_cut.py times the phases by difference -- results wrong --
// and READS THIS LIST (name = number, // description), so the kernel and the tool cannot disagree about what a cut means.
enum MapCut : uint32_t {
  kCutHeader = 10,         // segment record and sketch loaded
  kCutStaged = 11,         // hits loaded, ordered, staged
  kCutSketchTable = 1,     // bucket table of the sketch
  kCutL1 = 2,              // L1 scan: candidates listed, none evaluated
  kCutCandidate = 3,       // candidate set-up: begins, end of the slide, hit range
  kCutGroupBounds = 4,     // seed-hit bounds of every group of begins, no rounds (nothing raises the bar: an upper estimate)
  kCutStretch = 5,         // first round only: stretch loaded, window ends found
  kCutRanks = 6,           // first round only: stretch ranked against the sketch
  kCutCoarseTable = 7,     // first round only: coarse bit table built
  kCutCoarseSearch = 8,    // first round only: window masks and coarse search
  kCutNone = 9,            // the whole kernel
  kCutNoSecondPass = 21,   // the whole kernel without the second pass of rounds with more than 64 items
  kCutNoFinePass = 22,     // the whole kernel without the fine passes
  kCutFirstGroupOnly = 23, // the whole kernel, the group of the expected optimum only (one group per candidate)
};
#ifdef PA_TOOLS
#define PA_MAP_CUT_PARAM , uint32_t cut
#define PA_MAP_CUT_ARG , map_cut
#define PA_CUT(k) do { if (cut == (k)) return; } while (0)
#define PA_CUT_IS(k) (cut == (k))
#else
#define PA_MAP_CUT_PARAM
#define PA_MAP_CUT_ARG
#define PA_CUT(k) do { } while (0)
#define PA_CUT_IS(k) false
#endif
// one wave per (fragment, reference genome) segment
#ifndef PA_MAP_WAVES
#define PA_MAP_WAVES 4  // waves per SIMD the register allocation aims at: 128 VGPRs, the kernel needs 117 without spilling (80 registers / 6 waves: 0.64 s instead of 0.48 s for the 1 000-genome run)
#endif
// kAllStaged: the launch holds only segments of at most hit_cap hits (the bucketed path's list of short segments), so
// every access to a hit is an LDS read and the choice is not made per access.
template <uint32_t kRefCap, bool kAllStaged>
__global__ __launch_bounds__(64, PA_MAP_WAVES) void map_segments_kernel(
    uint64_t *__restrict__ keys, const uint32_t *__restrict__ vals, const uint4 *__restrict__ seg_rec, uint32_t n_segs,
    bool presorted, const uint32_t *__restrict__ q_hash, const uint32_t *__restrict__ frag_genome_local, uint32_t frag_len,
    uint32_t count_windows, const uint32_t *__restrict__ tab_min_shared,
    const uint32_t *__restrict__ contig_mini_off, const uint32_t *__restrict__ contig_bucket_off,
    const uint32_t *__restrict__ bucket_first, const uint32_t *__restrict__ mini_hash,
    const uint32_t *__restrict__ mini_wpos, const int32_t *__restrict__ prev_same,
    const uint32_t *__restrict__ contig_bin_off, uint64_t table_stride, unsigned long long *__restrict__ table,
    uint32_t *__restrict__ run_g, uint32_t s_cap, uint32_t hit_cap PA_MAP_CUT_PARAM) {
  extern __shared__ uint32_t eval_lds[];
  // the short-segment launch stages up to kHitCapSmall hits, the other one up to kHitCap (what the host passes as hit_cap)
  constexpr uint32_t kStageCap = kAllStaged ? (uint32_t)kHitCapSmall : (uint32_t)kHitCap;
  const EvalShared sh = eval_carve(eval_lds, s_cap, kStageCap, kRefCap);
  const uint32_t lane = threadIdx.x;
  if (blockIdx.x >= n_segs) return;
#ifdef PA_MAP_STATS  // event counts of the mapping kernel in run_g[0 .. 15] (tools: -DPA_MAP_STATS, PA_FRAGANI_TRACE=1)
#define PA_STAT(slot, v) do { const uint32_t pa_stat_v = (uint32_t)(v); if (lane == 0) atomicAdd(&run_g[slot], pa_stat_v); } while (0)  // (v may hold a ballot: every lane evaluates it)
#else
#define PA_STAT(slot, v) do { } while (0)
  (void)run_g;
#endif
  const uint4 rec0 = seg_rec[2 * (uint64_t)blockIdx.x], rec1 = seg_rec[2 * (uint64_t)blockIdx.x + 1];  // segment_records_kernel
  const uint32_t a0 = rec0.x, nh = rec0.y, f = rec0.z, s = rec0.w, mh = rec1.x;
  const int32_t unseeded = (int32_t)rec1.z;  // hashes of the sketch whose reference occurrences are not all seed hits (frequency cut)
  if (s == 0) return;  // no sketch, or fewer seed hits than any L1 run needs
  PA_STAT(0, 1);   // segments that reach L1
  PA_STAT(25 + min(6u, (uint32_t)(32 - __builtin_clz(nh | 1u)) > 3u ? (uint32_t)(32 - __builtin_clz(nh | 1u)) - 3u : 0u), 1);  // ... by their seed hits
  PA_STAT(1, nh);  // their seed hits
  // segments of up to kHitCap hits are staged in LDS; larger ones (repeats: rRNA operons, IS elements)
  // are read in place from the sorted hit arrays
  // contigs are kept relative to the reference genome's first one, window ids of the query as 16 bits: the
  // host takes this kernel only when both fit
  const uint32_t hc_base = rec1.y;
  const bool staged = kAllStaged || nh <= kStageCap;
  (void)hit_cap;
  auto HW = [&](uint32_t i) -> uint32_t { return staged ? sh.hw[i] >> 8 : (uint32_t)(keys[a0 + i] & 0xffffffu); };
  // "the window id of hit i is below w" without the shift (the searches of the bounds ask nothing else)
  // (window ids stay below 2^24 - 1: contigs are shorter than 2^24)
  auto HW_below = [&](uint32_t i, uint32_t w) -> bool {
    return staged ? sh.hw[i] < ((w < 0xffffffu ? w : 0xffffffu) << 8) : (uint32_t)(keys[a0 + i] & 0xffffffu) < w;
  };
  auto HC = [&](uint32_t i) -> uint32_t {
    return staged ? hc_base + sh.hc[i] : (uint32_t)(keys[a0 + i] >> 24) & 0xfffffu;
  };
  {  // the fragment's sketch, 16 bytes per lane and turn (s_cap is a multiple of 64; a row of q_hash holds kQMax hashes)
    const uint4 *src4 = reinterpret_cast<const uint4 *>(q_hash + (uint64_t)f * kQMax);
    uint4 *dst4 = reinterpret_cast<uint4 *>(sh.qh);
    for (uint32_t i = lane; i < s_cap / 4u; i += 64) dst4[i] = src4[i];
  }
  PA_CUT(kCutHeader);  // segment header and sketch
  if (staged) {
    if (presorted) {
      for (uint32_t i = lane; i < nh; i += 64) {  // hits ordered as a whole: the fragment on top of the key, the rank in the payload
        const uint64_t key = keys[a0 + i];
        sh.hw[i] = ((uint32_t)(key & 0xffffffu) << 8) | ((vals[a0 + i] & 0x1ffu) >> 1);
        sh.hc[i] = (uint16_t)(((uint32_t)(key >> 24) & 0xfffffu) - hc_base);
      }
    } else {
      // the bucketing pass leaves a segment's hits in no particular order: ordered in registers on their way to LDS
      if (nh <= 64u) stage_hits_sorted<1>(keys + a0, nh, hc_base, lane, sh.hw, sh.hc);
      else if (nh <= 128u) stage_hits_sorted<2>(keys + a0, nh, hc_base, lane, sh.hw, sh.hc);
      else if (kAllStaged || nh <= 256u) stage_hits_sorted<4>(keys + a0, nh, hc_base, lane, sh.hw, sh.hc);
      else if constexpr (!kAllStaged) stage_hits_sorted<8>(keys + a0, nh, hc_base, lane, sh.hw, sh.hc);
    }
  }
  PA_CUT(kCutStaged);  // hits staged in order
  // (sh.cnt is zeroed where the cooperative evaluation uses it: the bit tables of the rounds live in the same memory)
  if (lane < (uint32_t)kQMax / 32) sh.matched[lane] = 0;
  // The fragment's hashes bucketed by their top kQtBits bits: a reference minimizer's rank among them is then the
  // bucket's first rank plus a search among the bucket's few hashes (a fraction of a hash per bucket on average)
  // instead of log2(s) dependent LDS reads.  Buckets with more than 63 hashes (degenerate sketches) switch it off.
  {
    uint32_t *qt32 = reinterpret_cast<uint32_t *>(sh.qt);
    for (uint32_t i = lane; i < kQtBuckets / 2u; i += 64) qt32[i] = 0;
  }
  __syncthreads();
  for (uint32_t i = lane; i < s; i += 64) atomicAdd_u16(sh.qt, sh.qh[i] >> kQtShift);
  __syncthreads();
  uint32_t qsteps = 0;  // halving steps of the in-bucket search; 0xffffffff: table not usable
  {
    constexpr uint32_t kOwnB = kQtBuckets / 64u;  // consecutive buckets per lane
    uint32_t cntb[kOwnB], local = 0, most = 0;
#pragma unroll
    for (uint32_t q = 0; q < kOwnB; ++q) { cntb[q] = sh.qt[lane * kOwnB + q]; local += cntb[q]; most = max(most, cntb[q]); }
    uint32_t run = wave_excl_scan(local, lane);
    most = pa_dev::wave_max_dpp(most);
#pragma unroll
    for (uint32_t q = 0; q < kOwnB; ++q) { sh.qt[lane * kOwnB + q] = (uint16_t)(run | (cntb[q] << 10)); run += cntb[q]; }
    qsteps = most > 63u ? 0xffffffffu : (most ? 32u - (uint32_t)__builtin_clz(most) : 0u);
  }
  __syncthreads();

  // number of hits whose (contig, window id) is below (c, w0) and below (c, w1) -- the hits are in that order, so these
  // are the bounds of the hits on contig c with window id in [w0, w1).  Uniform arguments: the wave counts side by
  // side, every lane its own hits, instead of searching (no chain of dependent reads).
  auto hit_range = [&](uint32_t c, uint32_t w0, uint32_t w1, uint32_t &below0, uint32_t &below1) {
    const uint64_t k0 = ((uint64_t)c << 32) | w0, k1 = ((uint64_t)c << 32) | w1;
    uint32_t n0 = 0, n1 = 0;
    for (uint32_t chunk = 0; chunk < nh; chunk += 64) {
      const uint32_t i = min(chunk + lane, nh - 1u);
      const uint64_t k = ((uint64_t)HC(i) << 32) | HW(i);
      const bool in = chunk + lane < nh;
      n0 += (uint32_t)__popcll(__ballot(in & (k < k0)));
      n1 += (uint32_t)__popcll(__ballot(in & (k < k1)));
    }
    below0 = n0;
    below1 = n1;
  };
  // first hit index in [lo, lo + 2^steps) and below hi whose window id is >= w, inside an index range that lies on one
  // contig; `steps` is uniform (the longest range any lane has), every lane takes that many halvings without a branch
  auto hit_lower_bound_w = [&](uint32_t lo, uint32_t hi, uint32_t w, uint32_t steps) -> uint32_t {
    uint32_t pos = lo;  // hits [lo, pos) are below w
    for (uint32_t step = steps ? 1u << (steps - 1u) : 0u; step > 0u; step >>= 1) {
      const uint32_t idx = pos + step;
      const bool ok = (idx <= hi) & HW_below(min(idx, nh) - 1u, w);
      pos = ok ? idx : pos;
    }
    return pos;
  };

  // One over-long window (more than kRefCap minimizers: low-complexity or N-riddled sequence) straight from HBM,
  // the whole wave on it: reference-only hashes are counted per query-rank gap, matches set a bit, one scan
  // gives how many of the fragment's smallest hashes sit in the bottom-s of the union.  Uniform result.
  auto eval_window_coop = [&](uint32_t b0, uint32_t e) -> uint32_t {
    for (uint32_t t = b0 + lane; t < e; t += 64) {
      if (prev_same[t] >= (int32_t)b0) continue;  // the same hash already counted inside this window
      const uint32_t h = mini_hash[t];
      const uint32_t r = lower_bound_u32(sh.qh, 0, s, h);
      if (r < s && sh.qh[r] == h) atomicOr(&sh.matched[r >> 5], 1u << (r & 31u));
      else atomicAdd(&sh.cnt[r], 1u);
    }
    __syncthreads();
    uint32_t x;
    {
      const uint32_t per = s / 64u + 1u;  // buckets per lane: s + 1 of them
      uint32_t local_sum = 0;
      for (uint32_t q = 0; q < per; ++q) {
        const uint32_t r = lane * per + q;
        if (r <= s) local_sum += sh.cnt[r];
      }
      uint32_t prefix = wave_excl_scan(local_sum, lane), acc = 0;
      for (uint32_t q = 0; q < per; ++q) {
        const uint32_t r = lane * per + q;
        if (r <= s) {
          const uint32_t cr = sh.cnt[r];
          sh.cnt[r] = 0;  // ready for the next use
          const int32_t room = (int32_t)s - (int32_t)r - (int32_t)prefix;
          if (room > 0) acc += cr < (uint32_t)room ? cr : (uint32_t)room;
          prefix += cr;
        }
      }
      x = wave_sum(acc);
    }
    const uint32_t take = s - x;  // the `take` smallest query hashes are in the bottom-s of the union
    uint32_t shared = 0;
    if (lane < (uint32_t)kQMax / 32) {
      const uint32_t lo_bit = lane * 32u;
      uint32_t m = sh.matched[lane];
      sh.matched[lane] = 0;
      if (take <= lo_bit) m = 0;
      else if (take < lo_bit + 32u) m &= (1u << (take - lo_bit)) - 1u;
      shared = __popc(m);
    }
    shared = wave_sum(shared);
    __syncthreads();
    return shared;
  };

  PA_CUT(kCutSketchTable);  // staging, sort, bucket table
  // The best mapping so far lives in LDS (sh.scan[16..19]: it is looked at once per candidate, and values kept in
  // registers across the whole kernel were spills): shared minimizers; contig; window ids of the first minimizers of its
  // first and of its last optimal state.
  // kBestT: how many of the fragment's smallest hashes lie in the bottom-s of the union with the best window so far (T
  // below) -- the pivot rank of the tight bound; 0xffffffff: not known.
  // kLmaskPivot: the pivot sh.lmask stands for (the hits are the segment's: the bits outlive a candidate).
  enum { kBestShared = 16, kBestC, kBestFirst, kBestLast, kBestT, kLmaskPivot };
  constexpr uint32_t kNoT = 0xffffffffu;
  if (lane == 0) {
    sh.scan[kBestShared] = 0xffffffffu;  // -1
    sh.scan[kBestC] = 0xffffffffu;
    sh.scan[kBestT] = kNoT;
    sh.scan[kLmaskPivot] = kNoT;
  }
  uint32_t half0 = 1;
  while (2u * half0 <= s) half0 *= 2u;

  // ---- L2, the exact slide (oracle/fragani_oracle.c, L2 rule 2).  The window at position i of the candidate's contig
  // holds the minimizers of the reference windows [i, i + count_windows): from b = the last minimizer recorded at or
  // before i (still active in window i) to e = the first one recorded at or after i + count_windows.  The slide starts
  // at the first minimizer of the candidate range and ends as soon as e reaches the first minimizer at or past
  // rangeEnd + fragLen (or the contig's end): positions up to i_max = (window id of the minimizer before that) -
  // count_windows.  A STATE is a maximal run of positions with the same (b, e); its position is the window id of its first
  // minimizer b (Mashmap's), and per candidate the mapping position is the mean of the positions of the first and of the
  // last state with the most shared minimizers.
  // One lane per begin b, as a group of 64 begins is taken up: its states are the ends e from "first minimizer at or
  // after P[b] + count_windows" to "first at or after min(P[b+1] - 1, i_max) + count_windows", usually one or two.  In a
  // round the states of the group's pending begins are spread over the lanes in slide order (up to 64 of them: ITEMS),
  // the stretch of minimizers they cover (~300) is ranked against the fragment's hashes once and entered into bit tables
  // over (query rank x stretch position), and every lane finds, by two short searches over rows of those tables, how many
  // of the fragment's smallest hashes lie in the bottom-s of the union with its own window and how many of them the window
  // holds.  Begins none of whose windows can hold as many seed hits as the best so far shares are never evaluated.
  auto process_candidate = [&](uint32_t c, uint32_t cs, uint32_t ce, uint32_t first_hit_w) __attribute__((always_inline)) {
    PA_CUT(kCutL1);  // L1 only
    const int32_t best_shared = (int32_t)__builtin_amdgcn_readfirstlane((int)sh.scan[kBestShared]);  // (fixed while this candidate is evaluated)
    const uint32_t m1 = contig_mini_off[c + 1];
    const uint32_t bb = contig_bucket_off[c], nb = contig_bucket_off[c + 1] - bb - 1;
    // first begin: through the bucket index, the bucket itself searched by the whole wave (two memory round trips
    // instead of the six or so of a binary search)
    uint32_t b_lo;
    {
      uint32_t bk = cs >> kBucketShift;
      if (bk >= nb) bk = nb;
      const uint32_t lo = bucket_first[bb + bk];
      const uint32_t hi = bk < nb ? bucket_first[bb + bk + 1] : lo;
      b_lo = hi;
      for (uint32_t base = lo; base < hi; base += 64) {
        const uint32_t t = base + lane;
        const uint64_t ge = __ballot(t < hi && mini_wpos[t] >= cs);
        if (ge) { b_lo = base + (uint32_t)__builtin_ctzll(ge); break; }
      }
    }
    if (b_lo >= m1) return;
    // window ids of the first 512 begins in one batch of loads: they say where the slide ends, where the range of begins
    // ends and where the first seed hit sits (the groups re-read their own 64 window ids later: they are in L2 by then,
    // and eight registers are free)
    constexpr int kStartBatch = 8;
    uint32_t b_hi = 0xffffffffu, at = 0xffffffffu, i_max;
    uint32_t h_lo, h_hi;  // the seed hits any window of this candidate can hold: hits on contig c with window id in [cs, i_max + count_windows)
    uint32_t est = first_hit_w;  // where the optimum is expected
    {
      uint32_t wpv[kStartBatch];
#pragma unroll
      for (int q = 0; q < kStartBatch; ++q) {
        const uint32_t t = b_lo + (uint32_t)q * 64u + lane;
        wpv[q] = t < m1 ? mini_wpos[t] : 0xffffffffu;
      }
      // where the slide ends: the window's end may not reach the first minimizer at or past rangeEnd + fragLen (or the
      // contig's end); z = the window id of the minimizer before that one -- the largest of the batch below the limit when
      // the batch reaches the limit (it nearly always does: a candidate range spans at most two fragment lengths)
      const uint32_t limit = ce + frag_len;
      uint32_t z = 0;
      bool reached = false;
#pragma unroll
      for (int q = 0; q < kStartBatch; ++q) {
        z = wpv[q] < limit ? max(z, wpv[q]) : z;
        reached = reached || wpv[q] >= limit;
      }
      const uint32_t wp_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)wpv[0]);  // window id of the first begin
      if (wp_lo >= limit) return;  // no minimizer between the range's start and the limit
      if (__any(reached)) {
        z = pa_dev::wave_max_dpp(z);
      } else {
        const uint32_t last_end = wpos_lower_bound(mini_wpos, bucket_first, bb, nb, limit);
        z = mini_wpos[last_end - 1];
      }
      if (z < wp_lo + count_windows) return;  // the first window's end is already at the limit: nothing is evaluated
      i_max = z - count_windows;
      hit_range(c, cs, i_max + count_windows, h_lo, h_hi);
      // Where the optimum is expected: a fragment that maps at P leaves its hits in [P, P + count_windows), so P is about the
      // middle of the candidate's first and last hit less half a window -- for a diverged pair, whose first matching minimizer
      // sits anywhere in the fragment, a better guess than the first hit itself (never past it: the optimum holds hits).
      if (h_hi > h_lo) {
        const uint32_t mid = (HW(h_lo) + HW(h_hi - 1u)) / 2u, half = count_windows / 2u;
        est = min(first_hit_w, max(cs, mid > half ? mid - half : 0u));
      }
      // the first begin past the slide's last position and the first one at the expected optimum: every lane the first of
      // its own eight, then the minimum over the wave (the window ids ascend with the begin index)
      uint32_t my_over = 0xffffffffu, my_reach = 0xffffffffu;
#pragma unroll
      for (int q = kStartBatch - 1; q >= 0; --q) {
        my_over = wpv[q] > i_max ? (uint32_t)q * 64u + lane : my_over;
        my_reach = wpv[q] >= est ? (uint32_t)q * 64u + lane : my_reach;
      }
      const uint32_t w_over = pa_dev::wave_min_dpp(my_over), w_reach = pa_dev::wave_min_dpp(my_reach);
      if (w_over != 0xffffffffu) b_hi = b_lo + w_over;
      if (w_reach != 0xffffffffu) at = b_lo + w_reach;
    }
    if (b_hi == 0xffffffffu) b_hi = wpos_lower_bound(mini_wpos, bucket_first, bb, nb, i_max + 1u);  // a range of more than 512 begins
    if (at == 0xffffffffu) at = wpos_lower_bound(mini_wpos, bucket_first, bb, nb, est);
    if (b_lo >= b_hi) return;
    PA_STAT(2, 1);  // candidates with begins
#ifdef PA_MAP_STATS
    // the work model's units (profiles/README.md): the minimizers of the candidate's range, each of which a perfect bound
    // still has to look at once (from the first begin to the end of the last window), and -- below -- the states that tie the
    // candidate's optimum, which no bound can spare
    PA_STAT(32, wpos_lower_bound(mini_wpos, bucket_first, bb, nb, i_max + count_windows) - b_lo);
    uint32_t stat_ties = 0;
    int32_t stat_best_of_candidate = -1;
#endif
    int32_t c_best = -1;
    uint32_t c_first = 0, c_last = 0;
    // T of this candidate's best state so far, until there is one that of the fragment's best mapping so far: the pivot of the tight bound
    uint32_t pivot_T = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh.scan[kBestT]);
    // The begins are taken up in groups of 64 (one lane each), tiled so that the begin at the expected optimum sits in the
    // MIDDLE of its group (the tiles start `pad` begins before the first begin): that group goes first and sets the bar, and
    // with the tight bound below a begin more than a few matches away from the optimum is dropped -- the optimum in the
    // middle of the first group, the other groups are passed over or end with the bound.  (With the count of seed hits as
    // the only bound the tiling did not matter: the begins that passed spanned two groups by their number.)
    // A begin matters only if one of its windows can hold min_shared minimizers of the fragment (less is never reported)
    // and reach the best so far: first asked per group (seed hits between its first begin and the end of its last
    // window), then per begin.
    const uint32_t pad = (32u - ((at - b_lo) & 63u)) & 63u;
    const uint32_t tile0 = b_lo - pad;  // (may wrap below zero: begins before b_lo take no part)
    const uint32_t n_groups = (b_hi - tile0 + 63u) / 64u;
    const int32_t floor_bar = (int32_t)tab_min_shared[s];
    const uint32_t h_steps = 32u - (uint32_t)__builtin_clz(h_hi - h_lo + 1u);  // 2^steps > the number of hits: enough halvings
    PA_CUT(kCutCandidate);  // candidate set-up
    const uint32_t g_first = min((at - tile0) / 64u, n_groups - 1u);
    for (uint32_t gi = 0; gi < n_groups; ++gi) {
      const uint32_t g = gi == 0 ? g_first : (gi <= g_first ? gi - 1u : gi);
      if (gi > 0 && PA_CUT_IS(kCutFirstGroupOnly)) break;  // (timing experiment: the group of the first seed hit only; results wrong)
      const uint32_t sb = tile0 + g * 64u;
      const uint32_t b = sb + lane;
      const bool has = b >= b_lo && b < b_hi;
      const uint32_t wp = has ? mini_wpos[b] : 0u;
      const uint32_t wp_next = (has && b + 1u < m1) ? mini_wpos[b + 1u] : 0xffffffffu;
      const uint32_t w_end = min(wp_next - 1u, i_max) + count_windows;  // window ids below this: the begin's widest window
      // what a window must reach to matter (ties matter) -- in SEED HITS: a window shares no more than the seed hits it holds
      // plus the sketch's hashes that the frequency cut took out of the seeds (nearly always none)
      int32_t bar = c_best > best_shared ? c_best : best_shared;
      if (bar < floor_bar) bar = floor_bar;
      bar -= unseeded;
      // A window with `bar` of the candidate's hits ends after the bar-th hit and does not begin after the bar-th hit
      // from the end: groups without such a begin are passed over before any counting
      if (bar > 0) {
        if ((uint32_t)bar > h_hi - h_lo) break;  // no window of this candidate holds that many (the bar only rises)
        const uint32_t w_after = HW(h_lo + (uint32_t)bar - 1u), w_upto = HW(h_hi - (uint32_t)bar);
        if (!__any(has && w_end > w_after && wp <= w_upto)) continue;
      }
      // Seed hits inside the begin's widest window: every occurrence of every query hash is a hit (but for the hashes the
      // frequency cut took out: `unseeded`, allowed for in the bar), so no window shares more.  Only "at least b of them" is ever asked: with i0 = the first hit at or after the begin, that is "hit
      // i0 + b - 1 exists and lies before the window's end" -- one search and one read instead of two searches.
      const uint32_t i0 = hit_lower_bound_w(h_lo, h_hi, wp, h_steps);
      auto holds_hits = [&](int32_t b) -> bool {
        const uint32_t idx = i0 + (uint32_t)max(b, 1) - 1u;
        return has & (idx < h_hi) & HW_below(min(idx, nh - 1u), w_end);
      };
      bool pending = bar > 0 ? holds_hits(bar) : has;
      if (PA_CUT_IS(kCutGroupBounds)) pending = false;  // seed-hit bounds of every group
      PA_STAT(3, 1);                              // groups of 64 begins
      PA_STAT(4, __popcll(__ballot(pending)));    // begins that pass the seed-hit bound
      uint32_t e_next = 0;  // end (minimizer index) of the begin's next state; 0: none of its states has been evaluated yet
      while (__any(pending)) {
        PA_STAT(5, 1);  // rounds
        PA_STAT(18 + min(6u, (uint32_t)(32 - __builtin_clz(nh | 1u)) > 3u ? (uint32_t)(32 - __builtin_clz(nh | 1u)) - 3u : 0u), 1);  // ... by the segment's seed hits: <= 7, 8-15, 16-31, ..., 256 and more
        const uint32_t first_lane = (uint32_t)__builtin_ctzll(__ballot(pending));
        const uint32_t base = sb + first_lane;  // stretch = minimizers [base, base + n)
        const uint32_t n = min(m1 - base, kRefCap);
        constexpr int kPer = (int)(kRefCap / 64u);
        uint32_t hh[kPer];
        uint32_t dup_q = 0;  // bit q: the lane's q-th entry repeats a hash met earlier in the stretch
        const uint32_t wbase = mini_wpos[base];
        __syncthreads();
        // the stretch: hashes into registers, window ids (relative to the first) and duplicate links into LDS
#pragma unroll
        for (int q = 0; q < kPer; ++q) {
          const uint32_t x = (uint32_t)q * 64u + lane;
          const bool in = x < n;
          const uint32_t t = base + min(x, n - 1u);  // loads without a branch: beyond the stretch its last entry, discarded
          const uint32_t h = mini_hash[t], dw = mini_wpos[t] - wbase;
          const int32_t pv = prev_same[t];
          hh[q] = in ? h : 0u;
          // the same hash earlier in the stretch (position + 1): a window keeps this occurrence only if it starts after that one
          const uint32_t p1 = (in && pv >= (int32_t)base) ? (uint32_t)(pv - (int32_t)base) + 1u : 0u;
          sh.prev[x] = (uint16_t)p1;
          dup_q |= (p1 ? 1u : 0u) << q;
          sh.ref_w[x] = (uint16_t)((dw > 0xfffeu || !in) ? 0xffffu : dw);  // 0xffff: far beyond any window of this stretch
        }
        __syncthreads();
        // per begin: the ends of its first and last state inside the stretch, and how many of its remaining states the
        // stretch holds (a state with end x is held when entry x is in the stretch, or the stretch runs to the contig's end)
        const bool at_end = base + n == m1;
        const uint32_t lim = at_end ? n : n - 1u;
        const bool lane_on = pending && lane >= first_lane;  // (pending lanes are at or after first_lane by definition)
        uint32_t cnt = 0, xs0 = 0, xe_lo = 0, xe_hi = 0, top = 0;
        bool hi_known = false;
        {
          auto first_at_or_after = [&](uint32_t lo, uint32_t target) -> uint32_t {
            uint32_t hi = n;
            while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if ((uint32_t)sh.ref_w[mid] < target) lo = mid + 1; else hi = mid; }
            return lo;
          };
          const bool lane_has = has && lane >= first_lane;
          if (lane_has) xe_lo = first_at_or_after(min(b - base, n), wp + count_windows - wbase);
          // the last state of a begin ends where the first state of the next begin does, or one entry before that when
          // this very entry comes in at the next begin's position (the next lane has searched for it); the last begin of
          // the slide and the last lane search themselves
          const uint32_t xe_lo_next = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(min(lane + 1u, 63u) << 2), (int)xe_lo);
          if (lane_on) {
            if (lane < 63u && b + 1u < b_hi) {
              const bool comes_in_there = xe_lo_next > 0u && (uint32_t)sh.ref_w[xe_lo_next - 1u] == wp_next + count_windows - 1u - wbase;
              xe_hi = xe_lo_next - (comes_in_there ? 1u : 0u);
            } else {
              xe_hi = first_at_or_after(xe_lo, w_end - wbase);
            }
            xs0 = e_next ? e_next - base : xe_lo;
            top = min(xe_hi, lim);
            cnt = xs0 <= top ? top - xs0 + 1u : 0u;  // states of the begin the stretch holds
            hi_known = xe_hi < n || at_end;
          }
        }
        // ---- The tight bound.  With T = the number of the fragment's smallest hashes that lie in the bottom-s of the union
        // with a window (rank r is one of them iff r + c(r) < s, c(r) = the window's reference-only hashes below the
        // fragment's hash of rank r), the window shares its matches of rank < T.  For ANY pivot rank r0: c(r) >= c(r0) from
        // r0 on, hence T <= max(r0, s - c(r0)), and
        //     shared <= matches of rank < r0  +  min(matches of rank >= r0, max(0, s - r0 - c(r0))).
        // The matches of a begin's windows are seed hits, and a hit knows the rank of its hash: those below r0 among the
        // hits inside the begin's WIDEST window come from a bit per hit and two prefix counts (sh.lmask).  c(r0) is bounded
        // from below over the begin's NARROWEST window: stretch entries there whose hash is below the fragment's hash of
        // rank r0 -- entries that repeat a hash of the stretch left out, so that no hash counts twice -- less the hits below
        // r0 (they are such entries; taking those of the widest window only lowers the count).  With r0 = T of the best
        // window so far the bound is what the window would share if its reference-only hashes were as dense as that
        // window's: begins a few matches away from the optimum fail it, where the count of seed hits alone lets windows a
        // third of a fragment away pass (a window shares ~0.7 of its hits).  Asked before the stretch is ranked and the bit
        // tables are built -- the expensive part of a round --, for every pending begin at once.  Not asked when the
        // frequency cut took hashes of the sketch out of the seeds (their matches are not among the hits) and for segments
        // whose hits are not staged.
        {
          if (staged && unseeded == 0 && pivot_T != kNoT) {
            const uint32_t r0 = min(pivot_T, s - 1u) & ~1u;  // even: a staged hit keeps its rank without the lowest bit
            if ((uint32_t)__builtin_amdgcn_readfirstlane((int)sh.scan[kLmaskPivot]) != r0) {  // the bits stand for another pivot
              for (uint32_t chunk = 0; chunk <= nh; chunk += 64) {
                const uint32_t i = chunk + lane;
                const uint64_t lm = __ballot(i < nh && (sh.hw[min(i, nh - 1u)] & 0xffu) < (r0 >> 1));
                if (lane == 0) { sh.lmask[chunk / 32u] = (uint32_t)lm; sh.lmask[chunk / 32u + 1u] = (uint32_t)(lm >> 32); }
              }
              __builtin_amdgcn_wave_barrier();
              const uint32_t n_words = 2u * (nh / 64u) + 2u;
              const uint32_t pc = lane < n_words ? (uint32_t)__popc(sh.lmask[min(lane, kLmaskWords - 1u)]) : 0u;
              const uint32_t ex = pa_dev::wave_incl_scan_dpp(pc) - pc;
              if (lane < n_words) sh.lpre[lane] = ex;
              if (lane == 0) sh.scan[kLmaskPivot] = r0;
              __builtin_amdgcn_wave_barrier();
            }
            const uint32_t ph = sh.qh[r0];
            uint32_t *bw = sh.matched;   // 2 kPer words: stretch entries below the pivot hash; their prefix counts in sh.tab
#pragma unroll
            for (int q = 0; q < kPer; ++q) {
              const uint32_t x = (uint32_t)q * 64u + lane;
              const uint64_t mb = __ballot((x < n) & (hh[q] < ph) & (((dup_q >> q) & 1u) == 0u));
              if (lane == 0) { bw[2 * q] = (uint32_t)mb; bw[2 * q + 1] = (uint32_t)(mb >> 32); }
            }
            __builtin_amdgcn_wave_barrier();
            {
              const uint32_t pc = lane < 2u * (uint32_t)kPer ? (uint32_t)__popc(bw[min(lane, 2u * (uint32_t)kPer - 1u)]) : 0u;
              const uint32_t ex = pa_dev::wave_incl_scan_dpp(pc) - pc;
              if (lane < 2u * (uint32_t)kPer) sh.tab[lane] = ex;
            }
            __builtin_amdgcn_wave_barrier();
            auto below_upto = [&](uint32_t x) -> uint32_t {  // entries [0, x) below the pivot hash, x <= n
              const uint32_t w = min(x >> 5, 2u * (uint32_t)kPer - 1u), bits = x - (w << 5);
              const uint32_t m = bits >= 32u ? 0xffffffffu : (1u << bits) - 1u;
              return sh.tab[w] + (uint32_t)__popc(bw[w] & m);
            };
            auto hits_below = [&](uint32_t i) -> uint32_t {  // hits [0, i) whose rank lies below the pivot
              const uint32_t w = i >> 5;
              return sh.lpre[w] + (uint32_t)__popc(sh.lmask[w] & ((1u << (i & 31u)) - 1u));
            };
            int32_t bar_full = c_best > best_shared ? c_best : best_shared;
            if (bar_full < floor_bar) bar_full = floor_bar;
            bool fails = false;
            if (lane_on) {
              const uint32_t cb = below_upto(min(xe_lo, n)) - below_upto(min(b - base, n));
              const uint32_t i1 = hit_lower_bound_w(i0, h_hi, w_end, h_steps);
              const uint32_t mlow = hits_below(i1) - hits_below(i0), mhigh = (i1 - i0) - mlow;
              const uint32_t c2 = cb > mlow ? cb - mlow : 0u;
              const uint32_t room = s - r0 > c2 ? s - r0 - c2 : 0u;
              fails = (int32_t)(mlow + min(mhigh, room)) < bar_full;
            }
            PA_STAT(16, __popcll(__ballot(fails)));  // begins the tight bound drops
            pending = pending && !fails;
            if (!((__ballot(pending) >> first_lane) & 1ULL)) {  // the stretch was loaded for a begin that is gone: the next one, if any
              PA_STAT(17, 1);  // rounds that end here
              continue;
            }
          }
        }
        uint32_t f_shared = 0, p_state = 0;  // p_state: the position of the lane's state = the window id of its begin
        uint32_t t_state = kNoT;             // its T, where the fine search found it
        bool counted = false;  // the lane holds the exact value of a state (a window found out of reach of the bar is done, but not counted)
        uint32_t taken = 0;
        bool complete = false;  // the begin's last state is behind it
        // fold the evaluated states into the candidate's optimum: most shared; position of the first and of the last
        // state that has it (the lanes hold the states in slide order; the groups of a candidate do not come in that order).
        // The states that are not evaluated -- reached by taking in a minimizer the fragment does not hold -- share as many
        // as the state before them or fewer and have that state's begin, hence its position: they change neither.
#ifdef PA_MAP_STATS
        int32_t &stat_best = stat_best_of_candidate;
#endif
        auto fold_items = [&]() {
          const uint64_t dm = __ballot(counted);
          const int32_t group_best = (int32_t)pa_dev::wave_max_dpp(counted ? f_shared + 1u : 0u) - 1;
          if (dm && group_best >= c_best) {
            const uint64_t top_items = __ballot(counted && (int32_t)f_shared == group_best);
            const uint32_t w_first = __shfl(p_state, __builtin_ctzll(top_items), 64), w_last = __shfl(p_state, 63 - __builtin_clzll(top_items), 64);
            c_last = group_best > c_best ? w_last : max(c_last, w_last);
            c_first = group_best > c_best ? w_first : min(c_first, w_first);
            c_best = group_best;
            const uint32_t tt = (uint32_t)__builtin_amdgcn_readlane((int)t_state, __builtin_ctzll(top_items));
            if (tt != kNoT) pivot_T = tt;
#ifdef PA_MAP_STATS
            stat_ties = (group_best > stat_best ? 0u : stat_ties) + (uint32_t)__popcll(top_items);
            stat_best = group_best;
#endif
          }
        };
        if (__builtin_amdgcn_readlane((int)cnt, (int)first_lane) == 0) {
          PA_STAT(9, 1);  // cooperative evaluations
          // the first pending begin's next window is longer than the stretch: the whole wave takes that one state from HBM
          const uint32_t wp0 = __shfl(wp, (int)first_lane, 64), we0 = __shfl(w_end, (int)first_lane, 64);
          const uint32_t en0 = __shfl(e_next, (int)first_lane, 64);
          const uint32_t e_abs = en0 ? en0 : wpos_lower_bound(mini_wpos, bucket_first, bb, nb, wp0 + count_windows);
          const uint32_t e_last = wpos_lower_bound(mini_wpos, bucket_first, bb, nb, we0);
          if (e_abs <= e_last) {
            __syncthreads();
            for (uint32_t i = lane; i <= s; i += 64) sh.cnt[i] = 0;  // the cooperative form counts in the memory of the tables
            if (lane < (uint32_t)kQMax / 32) sh.matched[lane] = 0;    // (the rounds keep their bitmap of matching entries there)
            __syncthreads();
            const uint32_t v = eval_window_coop(base, e_abs);
            if (lane == first_lane) {
              f_shared = v;
              counted = true;
              p_state = wp0;
              e_next = e_abs + 1u;
            }
          }
          complete = lane == first_lane && e_abs + 1u > e_last;
          fold_items();
        } else {
          // ---- ranks of the lane's kPer minimizers among the fragment's hashes, for the part of the stretch that some
          // window of the pending begins reaches: the binary searches advance together, one halving step for all of them at
          // a time, so the LDS reads of a step are in flight at once
          const uint32_t n_rank = pa_dev::wave_max_dpp(lane_on ? top : 0u);
          uint32_t rank[kPer];
#pragma unroll
          for (int q = 0; q < kPer; ++q) rank[q] = 0;
          if (qsteps != 0xffffffffu) {
            uint32_t hi_r[kPer];
#pragma unroll
            for (int q = 0; q < kPer; ++q) {
              const uint32_t e = sh.qt[hh[q] >> kQtShift];
              rank[q] = e & 0x3ffu;          // hashes in the buckets below: all smaller
              hi_r[q] = rank[q] + (e >> 10);  // the hashes from here on are in higher buckets: all larger
            }
            for (uint32_t half = qsteps ? 1u << (qsteps - 1u) : 0u; half > 0; half >>= 1) {
#pragma unroll
              for (int q = 0; q < kPer; ++q) {
                const uint32_t idx = rank[q] + half;  // number of hashes below h is >= idx iff qh[idx - 1] < h
                if (idx <= hi_r[q] && sh.qh[idx - 1] < hh[q]) rank[q] = idx;
              }
            }
          } else {
            for (uint32_t half = half0; half > 0; half >>= 1) {  // half0 = largest power of two <= s: positions 0 .. 2*half0 - 1 >= s
#pragma unroll
              for (int q = 0; q < kPer; ++q) {
                const uint32_t idx = rank[q] + half;  // number of hashes below h is >= idx iff qh[idx - 1] < h
                if ((uint32_t)q * 64u + lane < n_rank && idx <= s && sh.qh[idx - 1] < hh[q]) rank[q] = idx;
              }
            }
          }
          // which entries match a hash of the fragment, as a bitmap over the stretch (scratch in the memory of the tables,
          // which are built later in the round)
          uint32_t match_q = 0;
          uint32_t *bm = sh.matched;  // (the cooperative evaluation, the other user of these words, clears them before it counts)
#pragma unroll
          for (int q = 0; q < kPer; ++q) {
            const uint32_t x = (uint32_t)q * 64u + lane;
            const bool is_match = (x < n_rank) & (rank[q] < s) & (sh.qh[min(rank[q], s - 1u)] == hh[q]);
            match_q |= (is_match ? 1u : 0u) << q;
            const uint64_t mb = __ballot(is_match);
            if (lane == 0) { bm[2 * q] = (uint32_t)mb; bm[2 * q + 1] = (uint32_t)(mb >> 32); }
          }
          __builtin_amdgcn_wave_barrier();
          // ---- which states are evaluated.  A state reached by taking in a minimizer that matches no hash of the fragment
          // (or repeats a hash the window holds) shares no more than the state before it, so the first state with the most
          // shared minimizers is never such a state, and the last one is the last evaluated state that has them or one of
          // the states right after it -- those are looked at once, when the candidate's other states are through (below).
          // Evaluated here: a begin's first state and every state reached by taking in a MATCHING minimizer.  Per round a
          // begin contributes the run of states from its first to its last such state among the next 16 (the few between
          // them ride along).
          // bit j: state xs0 + j is such a state
          uint32_t xa = xs0, len = 0;
          if (cnt) {
            const uint32_t p0 = xs0 - 1u, w0 = p0 >> 5;  // (xs0 >= 1: a window holds its begin)
            const uint64_t two = ((uint64_t)(w0 + 1u < (uint32_t)kQMax / 32u ? bm[w0 + 1u] : 0u) << 32) | bm[w0];
            uint32_t m = (uint32_t)(two >> (p0 & 31u));
            if (xs0 == xe_lo) m |= 1u;
            m &= (2u << min(top - xs0, 15u)) - 1u;
            if (m == 0u) {
              xa = xs0 + min(top - xs0, 15u) + 1u;  // nothing to evaluate among them
            } else {
              const uint32_t first = (uint32_t)__builtin_ctz(m), last = 31u - (uint32_t)__builtin_clz(m);
              xa = xs0 + first;
              len = last - first + 1u;
            }
          }
          // ---- the chosen states of the pending begins, in slide order, spread over the lanes: item t is state t - off of
          // the begin whose run of items [off, incl) holds t (runs laid out by a prefix sum of the counts).  A round takes up
          // to 128 items: the tables are built once, the lanes go through them twice when there are more than 64.
          const uint32_t incl = pa_dev::wave_incl_scan_dpp(len), off = incl - len;
          const uint32_t n_items = min(128u, (uint32_t)__builtin_amdgcn_readlane((int)incl, 63));
          taken = off < 128u ? min(len, 128u - off) : 0u;
          // the begins: where their next state ends (past the states after the run that are not evaluated), and whether
          // the last state is behind them -- settled here, so that none of this is alive across the evaluation
          {
            uint32_t x_next = xs0;
            if (cnt) {
              x_next = taken == len ? xs0 + min(top - xs0, 15u) + 1u : xa + taken;
              e_next = base + x_next;
            }
            complete = lane_on && hi_known && x_next > xe_hi;
          }
          PA_CUT(kCutStretch);  // stretch loads and window ends
          PA_STAT(6, n_rank);                           // stretch entries ranked
          PA_STAT(7, n_items);                          // windows evaluated in the round
          PA_STAT(12, n_items == 0u ? 1u : 0u);
          PA_STAT(13, n_items > 64u ? 1u : 0u);
          if (n_items) {
            PA_CUT(kCutRanks);  // ranks
            // Every window of the round, one lane each, without ordering the stretch.  A window holds the stretch
            // positions [xs, xw) minus later occurrences of a hash it already holds: a bit mask W over the positions.  With
            // R_r / M_r = the positions of reference-only / matching minimizers of rank <= r among the fragment's hashes, the
            // reference-only minimizers below the fragment's hash of rank r number c(r) = |R_r & W|; that hash lies in the
            // bottom-s of the union iff r + c(r) < s, which holds for r < T and no other (r + c(r) grows strictly), and the
            // window shares |M_(T-1) & W| minimizers.  T comes from two searches per lane: over the rows at every kCoarse-th
            // rank, then over all ranks of the coarse group that holds it.  A row is R_r followed by M_r.
            uint32_t xs = 0, xw = 0;
            bool it_on = false;
            // the begin of item t: the first lane whose run ends past t (the run ends do not decrease over the lanes)
            auto item_setup = [&](uint32_t pass) {
              const uint32_t t = pass * 64u + lane;
              it_on = t < n_items;
              uint32_t src = 0;
              if (pass == 0u) {  // the run starts marked in LDS (scratch in the memory of the tables), then a running maximum over the lanes
                sh.tab[lane] = 0u;
                __builtin_amdgcn_wave_barrier();
                if (len && off < 64u) sh.tab[off] = lane + 1u;
                __builtin_amdgcn_wave_barrier();
                src = max(pa_dev::wave_incl_max_scan_dpp(sh.tab[lane]), 1u) - 1u;
              } else {  // a search: the first lane whose run ends past t
#pragma unroll
                for (uint32_t step = 32; step > 0; step >>= 1) {
                  const uint32_t probe = src + step - 1u;
                  const uint32_t v = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(min(probe, 63u) << 2), (int)incl);
                  src += (probe < 64u && v <= t) ? step : 0u;
                }
              }
              const int src_addr = (int)(min(src, 63u) << 2);
              const uint32_t packed = (uint32_t)__builtin_amdgcn_ds_bpermute(src_addr, (int)(((b - base) & 0x3ffu) | (xa << 10)));
              const uint32_t off_s = (uint32_t)__builtin_amdgcn_ds_bpermute(src_addr, (int)off);
              p_state = (uint32_t)__builtin_amdgcn_ds_bpermute(src_addr, (int)wp);  // the state's position: its begin's window id
              xs = it_on ? (packed & 0x3ffu) : 0u;                  // the window: stretch entries [xs, xw)
              xw = it_on ? (packed >> 10) + (t - off_s) : 0u;
            };
            item_setup(0u);
          constexpr uint32_t kW = kRefCap / 32u;  // words per half row; stretch position q * 64 + lane is bit (lane & 31) of word 2 q + (lane >> 5)
          constexpr uint32_t kRow = 2u * kW;
          const uint32_t n_coarse = s / kCoarse + 1u;  // the last row stands at a rank >= s: r + c(r) >= s holds there
          uint32_t *bc = sh.tab, *bf = sh.tab + n_coarse * kRow;
          __syncthreads();
          {
            uint4 *t4 = reinterpret_cast<uint4 *>(sh.tab);
            for (uint32_t i = lane; i < n_coarse * kRow / 4u; i += 64) t4[i] = make_uint4(0u, 0u, 0u, 0u);  // kRow is a multiple of 4
          }
          const uint16_t *prev16 = sh.prev;
          // word of an entry's bit inside a row (the matching half comes second)
          auto col_of = [&](int q) -> uint32_t { return 2u * (uint32_t)q + (lane >> 5) + (((match_q >> q) & 1u) ? kW : 0u); };
#pragma unroll
          for (int q = 0; q < kPer; ++q) {
            const uint32_t x = (uint32_t)q * 64u + lane;
            const uint32_t r = rank[q];
            const bool valid = (x < n_rank) & (r < s);  // inside the used part and below some hash of the fragment: in the rows from r on
            rank[q] = valid ? r : 0xffffffffu;
          }
          const bool any_dup = __any(dup_q != 0u);
          __syncthreads();
#pragma unroll
          for (int q = 0; q < kPer; ++q) {  // no branch: entries that are in no row add nothing to a word of the (not yet filled) fine rows
            const bool valid = rank[q] != 0xffffffffu;
            atomicOr(valid ? &bc[(rank[q] >> kCoarseShift) * kRow + col_of(q)] : &bf[lane], valid ? 1u << (lane & 31u) : 0u);
          }
          __syncthreads();
          // rows become prefixes: row g |= rows below it.  Lane = (pair of columns, run of rows); the runs of one column
          // pair are chained through the totals of the runs before
          auto prefix_or_rows = [&](uint32_t *rows, uint32_t n_rows) {
            constexpr uint32_t kRuns = 64u / kW;
            uint2 *rows2 = reinterpret_cast<uint2 *>(rows);  // kW column pairs per row
            const uint32_t cp = lane % kW, run = lane / kW;
            const uint32_t per = (n_rows + kRuns - 1u) / kRuns;
            const uint32_t r0 = min(n_rows, run * per), r1 = run < kRuns ? min(n_rows, r0 + per) : r0;
            uint2 acc = make_uint2(0u, 0u);
            for (uint32_t r = r0; r < r1; ++r) { const uint2 v = rows2[r * kW + cp]; acc.x |= v.x; acc.y |= v.y; }
            uint2 before = make_uint2(0u, 0u);
#pragma unroll
            for (uint32_t j = 1; j < kRuns; ++j) {
              const int from = (int)(lane >= kW * j ? lane - kW * j : lane);
              const uint32_t vx = (uint32_t)__shfl((int)acc.x, from, 64), vy = (uint32_t)__shfl((int)acc.y, from, 64);
              if (run >= j) { before.x |= vx; before.y |= vy; }
            }
            for (uint32_t r = r0; r < r1; ++r) {
              const uint2 v = rows2[r * kW + cp];
              before.x |= v.x; before.y |= v.y;
              rows2[r * kW + cp] = before;
            }
          };
          prefix_or_rows(bc, n_coarse);
          PA_CUT(kCutCoarseTable);  // coarse table
            for (uint32_t pass = 0; pass * 64u < n_items && !(pass && PA_CUT_IS(kCutNoSecondPass)); ++pass) {
              if (pass) item_setup(pass);
              // the lane's window as a mask over the stretch positions
              uint32_t wm[kW];
              {
                // bits [xs, xw) of the row, two words at a time: xs is below 64, so only the first pair has a lower end
                auto below = [](int32_t b) -> uint64_t {  // the b lowest bits of a pair of words, b clamped to 0 .. 64
                  const uint64_t m = b >= 64 ? ~0ULL : (1ULL << (b & 63)) - 1ULL;
                  return b <= 0 ? 0ULL : m;
                };
    #pragma unroll
                for (uint32_t w2 = 0; w2 < kW / 2u; ++w2) {
                  uint64_t m = below((int32_t)xw - (int32_t)(64u * w2));
                  if (w2 == 0) m &= ~below((int32_t)xs);
                  wm[2 * w2] = (uint32_t)m;
                  wm[2 * w2 + 1] = (uint32_t)(m >> 32);
                }
              }
              if (any_dup) {
    #pragma unroll
                for (int q = 0; q < kPer; ++q) {
                  for (uint64_t dmask = __ballot((dup_q >> q) & 1u); dmask; dmask &= dmask - 1) {
                    const uint32_t bit = (uint32_t)__builtin_ctzll(dmask);
                    if ((uint32_t)prev16[(uint32_t)q * 64u + bit] > xs) {  // the earlier occurrence lies inside this lane's window
                      if (bit < 32u) wm[2 * q] &= ~(1u << bit); else wm[2 * q + 1] &= ~(1u << (bit - 32u));
                    }
                  }
                }
              }
              auto count_in = [&](const uint32_t *half_row) -> uint32_t {
                const uint2 *row2 = reinterpret_cast<const uint2 *>(half_row);  // half rows start on 8-byte boundaries (kW is even)
                uint32_t c = 0;
    #pragma unroll
                for (uint32_t w = 0; w < kW / 2u; ++w) {
                  const uint2 v = row2[w];
                  c += __popc(v.x & wm[2 * w]) + __popc(v.y & wm[2 * w + 1]);
                }
                return c;
              };
              __syncthreads();
              // coarse: the first group g whose last rank r = kCoarse g + kCoarse - 1 has r + c(r) >= s (the last row always has)
              uint32_t g_lo = 0, g_hi = n_coarse - 1u;
              for (uint32_t span = n_coarse - 1u; span > 0u; span >>= 1) {  // as many halvings as the widest range needs
                const uint32_t mid = (g_lo + g_hi) >> 1;
                const bool ge = mid * kCoarse + (kCoarse - 1u) + count_in(bc + mid * kRow) >= s;
                const bool open = g_lo < g_hi;
                g_hi = (open & ge) ? mid : g_hi;
                g_lo = (open & !ge) ? mid + 1u : g_lo;
              }
              PA_CUT(kCutCoarseSearch);  // window masks, coarse search
              // fine: the groups of the lanes lie next to each other as a rule; kFineGroups of them per pass
              // A window shares |M_(T-1) & W| minimizers and T - 1 lies in the coarse group just found, so the matches up to the
              // group's last rank bound it from above (by the matches of at most 15 more ranks): windows that cannot reach
              // the bar any more -- most of a candidate's windows away from its optimum -- are done here, without the fine
              // tables, and are left out of the fold below (their exact value is below the bar, which is all that matters).
              int32_t bar_now = c_best > best_shared ? c_best : best_shared;
              if (bar_now < floor_bar) bar_now = floor_bar;
              const bool in_reach = (int32_t)count_in(bc + g_lo * kRow + kW) >= bar_now;
              counted = it_on && in_reach;
              bool unresolved = counted && !PA_CUT_IS(kCutNoFinePass);
              while (__any(unresolved)) {
                PA_STAT(8, 1);  // fine passes
                const uint32_t g_cur = pa_dev::wave_min_dpp(unresolved ? g_lo : 0xffffffffu);
                const uint32_t band0 = g_cur * kCoarse;  // row t of the band: ranks <= band0 + t - 1; row 0 is the coarse row below
                __syncthreads();
                {
                  uint4 *f4 = reinterpret_cast<uint4 *>(bf);  // bf starts on a 16-byte boundary: kRow is a multiple of 4 words
                  const uint4 *below = reinterpret_cast<const uint4 *>(bc + (g_cur ? g_cur - 1u : 0u) * kRow);
                  for (uint32_t i = lane; i < kFineRows * kRow / 4u; i += 64)
                    f4[i] = (i < kRow / 4u && g_cur > 0u) ? below[i] : make_uint4(0u, 0u, 0u, 0u);
                }
                __syncthreads();
    #pragma unroll
                for (int q = 0; q < kPer; ++q) {
                  const uint32_t t = rank[q] - band0;  // wraps to something huge below the band
                  const bool valid = (rank[q] != 0xffffffffu) & (t < kFineRows - 1u);
                  atomicOr(valid ? &bf[(t + 1u) * kRow + col_of(q)] : &bc[lane], valid ? 1u << (lane & 31u) : 0u);
                }
                __syncthreads();
                prefix_or_rows(bf, kFineRows);
                __syncthreads();
                const bool now = unresolved && g_lo - g_cur < kFineGroups;
                // first rank r of the lane's group with r + c(r) >= s: the group's last rank has it
                uint32_t r_lo = g_lo * kCoarse, r_hi = r_lo + kCoarse - 1u;
                if (!now) r_lo = r_hi = band0;
                for (uint32_t step = 0; step < kCoarseShift; ++step) {
                  const uint32_t mid = (r_lo + r_hi) >> 1;
                  const bool ge = mid + count_in(bf + (mid - band0 + 1u) * kRow) >= s;
                  const bool open = r_lo < r_hi;
                  r_hi = (open & ge) ? mid : r_hi;
                  r_lo = (open & !ge) ? mid + 1u : r_lo;
                }
                const uint32_t c = count_in(bf + (r_lo - band0) * kRow + kW);  // matches of rank < T = r_lo
                if (now) { f_shared = c; t_state = r_lo; unresolved = false; }
              }
              __syncthreads();
              PA_STAT(14, __popcll(__ballot(counted)));  // windows whose exact value was found
              PA_STAT(15, __popcll(__ballot(counted && (int32_t)f_shared >= max(max(c_best, best_shared), floor_bar))));  // ... at or above the bar
              fold_items();
            }
          }
        }
        PA_STAT(10, __popcll(__ballot(lane_on)));   // begins taking part in the rounds
        PA_STAT(11, __popcll(__ballot(complete)));  // begins finished by the rounds
        pending = pending && !complete;
        // whoever can no longer reach the bar drops out
        {
          int32_t bar2 = c_best > best_shared ? c_best : best_shared;
          if (bar2 < floor_bar) bar2 = floor_bar;
          bar2 -= unseeded;
          pending = pending && (bar2 > 0 ? holds_hits(bar2) : true);
        }
      }
    }
#ifdef PA_MAP_STATS
    PA_STAT(33, stat_ties ? stat_ties : 1u);  // states tying the candidate's optimum (one probe where nothing was evaluated)
#endif
    if (c_best < 0) return;
    // fastANI keeps every candidate as a mapping, orders a fragment's mappings by identity and lets each overwrite the one
    // before: of several candidates that share equally many minimizers the LAST one -- the candidates come in (contig,
    // position) order -- is the fragment's mapping.
    __syncthreads();
    if (c_best >= best_shared && lane == 0) {
      sh.scan[kBestShared] = (uint32_t)c_best; sh.scan[kBestC] = c; sh.scan[kBestFirst] = c_first; sh.scan[kBestLast] = c_last;
      sh.scan[kBestT] = pivot_T;
    }
    __syncthreads();
  };

  // ---- L1: run a is valid when hits a .. a+mh-1 share a contig and span < frag_len window ids; its candidate range
  // of window starts is [y.w - fragLen + 1, x.w]; ranges that touch on one contig merge (hits are in (contig, window)
  // order, so both ends only grow and "touches the merged range" is "touches the previous run's")
  // The candidates a chunk of 64 runs closes are first listed (LDS, kListCap at a time), then evaluated: the evaluation
  // is in the code once and none of the scan's per-lane state is alive across it.  A chunk that closes more than
  // kListCap candidates is scanned again for the rest (`handled` = its breaks already listed).
#ifndef PA_MAP_LIST_CAP
#define PA_MAP_LIST_CAP 8  // 1 in a test build: every chunk with two closed candidates is then scanned twice
#endif
  constexpr uint32_t kListCap = PA_MAP_LIST_CAP;
  bool have_cur = false, have_prev = false;
  uint32_t cur_c = 0, cur_cs = 0, cur_ce = 0, cur_fw = 0, prev_c = 0, prev_ce = 0;
  uint64_t handled = 0;
  uint32_t chunk = 0;
  for (;;) {
    const bool tail = chunk >= nh;  // one more turn after the last chunk lists the candidate still open
    uint32_t n_list = 0;
    auto list_current = [&]() {
      if (lane == 0) {
        uint4 *slot = reinterpret_cast<uint4 *>(sh.cand) + n_list;
        *slot = make_uint4(cur_c, cur_cs, cur_ce, cur_fw);
      }
      ++n_list;
    };
    if (tail) {
      if (have_cur) list_current();
    } else {
      const uint32_t i = chunk + lane;
      bool v = false;
      uint32_t c_i = 0, cs_i = 0, ce_i = 0;
      if (i + mh <= nh) {
        c_i = HC(i);
        ce_i = HW(i);
        const uint32_t yw = HW(i + mh - 1);
        v = HC(i + mh - 1) == c_i && yw - ce_i < frag_len;
        cs_i = yw + 1u > frag_len ? yw + 1u - frag_len : 0u;
      }
      const uint64_t vm = __ballot(v);
      if (!vm) { chunk += 64; continue; }
      // the valid run before this lane's: in this chunk, or carried over from the chunks before
      const uint64_t below = vm & ((1ULL << lane) - 1ULL);
      const int pl = below ? 63 - __builtin_clzll(below) : 0;
      const uint32_t sc = __shfl(c_i, pl, 64), se = __shfl(ce_i, pl, 64);
      const bool hp = below ? true : have_prev;
      const uint32_t pc = below ? sc : prev_c, pe = below ? se : prev_ce;
      const bool brk = v && (!hp || pc != c_i || cs_i > pe);
      const uint64_t bm_all = __ballot(brk);
      uint64_t bm = bm_all & ~handled;
      if (!handled) {
        // valid runs before the first break of the chunk extend the carried candidate
        const uint64_t head = bm_all ? vm & ((1ULL << __builtin_ctzll(bm_all)) - 1ULL) : vm;
        if (head && have_cur) cur_ce = max(cur_ce, (uint32_t)__builtin_amdgcn_readlane((int)ce_i, 63 - __builtin_clzll(head)));
      }
      while (bm && n_list < kListCap) {
        const int bit = __builtin_ctzll(bm);
        bm &= bm - 1;
        handled |= 1ULL << bit;
        if (have_cur) list_current();
        const uint64_t upto = bm ? ((1ULL << __builtin_ctzll(bm)) - 1ULL) : ~0ULL;
        const uint64_t mine = vm & upto & ~((1ULL << bit) - 1ULL);  // the valid runs of this group inside the chunk
        // (uniform lane numbers: v_readlane puts the values into scalar registers, where the scan's state belongs)
        cur_c = (uint32_t)__builtin_amdgcn_readlane((int)c_i, bit);
        cur_cs = (uint32_t)__builtin_amdgcn_readlane((int)cs_i, bit);
        cur_fw = (uint32_t)__builtin_amdgcn_readlane((int)ce_i, bit);  // window id of the first hit of the candidate's first run
        cur_ce = (uint32_t)__builtin_amdgcn_readlane((int)ce_i, 63 - __builtin_clzll(mine));
        have_cur = true;
      }
      if (!bm) {  // the chunk is done
        const int last = 63 - __builtin_clzll(vm);
        prev_c = (uint32_t)__builtin_amdgcn_readlane((int)c_i, last);
        prev_ce = (uint32_t)__builtin_amdgcn_readlane((int)ce_i, last);
        have_prev = true;
        handled = 0;
        chunk += 64;
      }
    }
    // A turn that lists nothing (most do: a segment of 139 hits is three chunks and ends with ONE candidate) goes straight on.
    if (n_list == 0u) {
      if (tail) break;
      continue;
    }
    // The scan's state sits in LDS while the listed candidates are evaluated: kept in registers across the evaluation --
    // the register-hungriest part of the kernel -- it was spilled to scratch memory (HBM traffic, and a wait) at every turn.
    if (lane == 0) {
      sh.scan[0] = chunk; sh.scan[1] = (have_cur ? 1u : 0u) | (have_prev ? 2u : 0u) | (tail ? 4u : 0u);
      sh.scan[2] = cur_c; sh.scan[3] = cur_cs; sh.scan[4] = cur_ce; sh.scan[5] = cur_fw; sh.scan[6] = prev_c; sh.scan[7] = prev_ce;
      sh.scan[8] = (uint32_t)handled; sh.scan[9] = (uint32_t)(handled >> 32); sh.scan[10] = n_list;
    }
    __syncthreads();
    for (uint32_t t = 0; t < sh.scan[10]; ++t) {
      const uint4 cand = reinterpret_cast<const uint4 *>(sh.cand)[t];
      process_candidate((uint32_t)__builtin_amdgcn_readfirstlane((int)cand.x), (uint32_t)__builtin_amdgcn_readfirstlane((int)cand.y),
                        (uint32_t)__builtin_amdgcn_readfirstlane((int)cand.z), (uint32_t)__builtin_amdgcn_readfirstlane((int)cand.w));
    }
    __syncthreads();
    {
      const uint32_t flags1 = sh.scan[1];
      chunk = sh.scan[0]; have_cur = flags1 & 1u; have_prev = (flags1 >> 1) & 1u;
      cur_c = sh.scan[2]; cur_cs = sh.scan[3]; cur_ce = sh.scan[4]; cur_fw = sh.scan[5]; prev_c = sh.scan[6]; prev_ce = sh.scan[7];
      handled = ((uint64_t)sh.scan[9] << 32) | sh.scan[8];
      if (flags1 & 4u) break;
    }
    __syncthreads();
  }

  __syncthreads();
  const int32_t best_shared = (int32_t)sh.scan[kBestShared];
  if (lane == 0 && best_shared >= 0 && (uint32_t)best_shared >= tab_min_shared[s]) {
    // fastANI buckets the reference by fragLen - 20
    const uint64_t jq = ((uint64_t)best_shared << 30) / s;
    const unsigned long long packed = ((unsigned long long)jq << 32) | ((unsigned long long)best_shared << 16) | s;
    const uint64_t bin = contig_bin_off[sh.scan[kBestC]] + (sh.scan[kBestFirst] + sh.scan[kBestLast]) / 2u / (frag_len - 20u);
    atomicMax(&table[(uint64_t)frag_genome_local[f] * table_stride + bin], packed);
  }
}

// ============================================================== 4b. segments of a handful of seed hits
// A pair of the same species that has diverged far leaves a fragment two to eight seed hits in a reference genome.  Such a
// segment is the mapping kernel's worst customer: every window that holds its few hits shares the same few minimizers or
// nearly, so the states that tie the optimum span hundreds of begins -- four or five rounds of ranking a stretch and
// building bit tables where a window holds eight matches at most (13 % of the segments, a quarter of the rounds at 1 000
// genomes).  With so few matches the windowed MinHash has a direct form: hit j, of rank r_j among the fragment's hashes,
// is shared by a window iff it lies in it and r_j + c_j < s, c_j = the window's reference-only minimizers below the hit's
// hash -- a count over a range of the stretch, two prefix sums per hit and state.  One wave per segment: no sketch, no
// rank, no table; per group of 64 begins the stretch's hashes are compared with the (at most eight) hit hashes, one bit
// mask and its prefix counts per hit go to LDS, and every begin evaluates all its states.  Same candidates, same slide,
// same positions and ties as map_segments_kernel (of which this is the small-segment form); whatever does not fit the
// simple form -- a hash met twice in a stretch, windows longer than the stretch -- sends the segment to
// map_segments_kernel through the overflow list.
constexpr uint32_t kSparseHits = kTinySegment;  // (16: 426 ms of mapping per 1 000-genome run against 419)
constexpr uint32_t kSparseCap = 384;  // stretch entries of a group of 64 begins (a window holds ~237: 5 sigma to spare)
__global__ __launch_bounds__(64) void map_sparse_kernel(
    const uint64_t *__restrict__ keys, const uint32_t *__restrict__ seg_a0, const uint32_t *__restrict__ seg_nh,
    const uint32_t *__restrict__ seg_f, uint32_t n_segs, const uint32_t *__restrict__ q_s, const uint32_t *__restrict__ q_hash,
    const uint32_t *__restrict__ frag_genome_local, uint32_t frag_len, uint32_t count_windows,
    const uint32_t *__restrict__ tab_min_hits, const uint32_t *__restrict__ tab_min_shared,
    const uint32_t *__restrict__ contig_mini_off, const uint32_t *__restrict__ contig_bucket_off,
    const uint32_t *__restrict__ bucket_first, const uint32_t *__restrict__ mini_hash, const uint32_t *__restrict__ mini_wpos,
    const int32_t *__restrict__ prev_same, const uint32_t *__restrict__ contig_bin_off, uint64_t table_stride,
    unsigned long long *__restrict__ table, uint32_t *__restrict__ over_a0, uint32_t *__restrict__ over_nh,
    uint32_t *__restrict__ over_f, uint32_t *__restrict__ over_n) {
  constexpr int kPer = (int)(kSparseCap / 64u);
  constexpr uint32_t kWords = kSparseCap / 32u;  // 12
  __shared__ uint32_t s_hc[kSparseHits], s_hw[kSparseHits], s_hr[kSparseHits], s_ph[kSparseHits], s_pos[kSparseHits];
  __shared__ uint32_t s_cand[kSparseHits][4];
  __shared__ uint16_t s_refw[kSparseCap];
  __shared__ uint32_t s_B[kSparseHits][kWords + 1], s_P[kSparseHits][kWords + 1];
  const uint32_t lane = threadIdx.x;
  if (blockIdx.x >= n_segs) return;
  const uint32_t a0 = seg_a0[blockIdx.x], nh = seg_nh[blockIdx.x], f = seg_f[blockIdx.x];
  const uint32_t s = q_s[f];
  if (s == 0 || nh > kSparseHits) return;
  const uint32_t mh = tab_min_hits[s];
  if (nh < mh) return;
  const int32_t floor_bar = (int32_t)tab_min_shared[s];
  {  // the hits in (contig, window id) order, with the rank and the hash of each
    uint64_t k1[1];
    const uint64_t raw = lane < nh ? keys[a0 + lane] : 0ULL;
    k1[0] = lane < nh ? ((raw & 0xfffffffffffULL) << 9) | (raw >> kHitRankShift) : ~0ULL;
    bitonic_sort_lanes<1, uint64_t>(k1, lane);
    if (lane < nh) {
      const uint32_t r = (uint32_t)k1[0] & 0x1ffu;
      s_hw[lane] = (uint32_t)(k1[0] >> 9) & 0xffffffu;
      s_hc[lane] = (uint32_t)(k1[0] >> 33) & 0xfffffu;
      s_hr[lane] = r;
      s_ph[lane] = q_hash[(uint64_t)f * kQMax + r];
    }
  }
  __syncthreads();
  // ---- L1 (map_segments_kernel's rule, one after the other: at most eight hits)
  uint32_t n_cand = 0;
  {
    bool have = false, have_prev = false;
    uint32_t cur_c = 0, cur_cs = 0, cur_ce = 0, cur_fw = 0, prev_c = 0, prev_ce = 0;
    for (uint32_t i = 0; i + mh <= nh; ++i) {
      const uint32_t c_i = s_hc[i], ce_i = s_hw[i], yw = s_hw[i + mh - 1u];
      if (!(s_hc[i + mh - 1u] == c_i && yw - ce_i < frag_len)) continue;
      const uint32_t cs_i = yw + 1u > frag_len ? yw + 1u - frag_len : 0u;
      if (!have_prev || prev_c != c_i || cs_i > prev_ce) {
        if (have) { if (lane == 0) { s_cand[n_cand][0] = cur_c; s_cand[n_cand][1] = cur_cs; s_cand[n_cand][2] = cur_ce; s_cand[n_cand][3] = cur_fw; } ++n_cand; }
        cur_c = c_i; cur_cs = cs_i; cur_ce = ce_i; cur_fw = ce_i; have = true;
      } else {
        cur_ce = max(cur_ce, ce_i);
      }
      prev_c = c_i; prev_ce = ce_i; have_prev = true;
    }
    if (have) { if (lane == 0) { s_cand[n_cand][0] = cur_c; s_cand[n_cand][1] = cur_cs; s_cand[n_cand][2] = cur_ce; s_cand[n_cand][3] = cur_fw; } ++n_cand; }
  }
  __syncthreads();
  bool overflow = false;
  int32_t best_shared = -1;
  uint32_t best_c = 0, best_first = 0, best_last = 0;
  for (uint32_t ci = 0; ci < n_cand && !overflow; ++ci) {
    const uint32_t c = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_cand[ci][0]);
    const uint32_t cs = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_cand[ci][1]);
    const uint32_t ce = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_cand[ci][2]);
    const uint32_t m1 = contig_mini_off[c + 1];
    const uint32_t bb = contig_bucket_off[c], nb = contig_bucket_off[c + 1] - bb - 1;
    uint32_t b_lo;
    {
      uint32_t bk = cs >> kBucketShift;
      if (bk >= nb) bk = nb;
      const uint32_t lo = bucket_first[bb + bk];
      const uint32_t hi = bk < nb ? bucket_first[bb + bk + 1] : lo;
      b_lo = hi;
      for (uint32_t base = lo; base < hi; base += 64) {
        const uint32_t t = base + lane;
        const uint64_t ge = __ballot(t < hi && mini_wpos[t] >= cs);
        if (ge) { b_lo = base + (uint32_t)__builtin_ctzll(ge); break; }
      }
    }
    if (b_lo >= m1) continue;
    uint32_t b_hi, i_max;
    {
      constexpr int kStartBatch = 8;
      uint32_t wpv[kStartBatch];
#pragma unroll
      for (int q = 0; q < kStartBatch; ++q) {
        const uint32_t t = b_lo + (uint32_t)q * 64u + lane;
        wpv[q] = t < m1 ? mini_wpos[t] : 0xffffffffu;
      }
      const uint32_t limit = ce + frag_len;
      uint32_t z = 0;
      bool reached = false;
#pragma unroll
      for (int q = 0; q < kStartBatch; ++q) {
        z = wpv[q] < limit ? max(z, wpv[q]) : z;
        reached = reached || wpv[q] >= limit;
      }
      const uint32_t wp_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)wpv[0]);
      if (wp_lo >= limit) continue;
      if (__any(reached)) {
        z = pa_dev::wave_max_dpp(z);
      } else {  // a range of more than 512 minimizers (two far-apart runs merged): through the bucket index, as the general kernel does
        const uint32_t last_end = wpos_lower_bound(mini_wpos, bucket_first, bb, nb, limit);
        z = mini_wpos[last_end - 1];
      }
      if (z < wp_lo + count_windows) continue;
      i_max = z - count_windows;
      uint32_t my_over = 0xffffffffu;
#pragma unroll
      for (int q = kStartBatch - 1; q >= 0; --q) my_over = wpv[q] > i_max ? (uint32_t)q * 64u + lane : my_over;
      const uint32_t w_over = pa_dev::wave_min_dpp(my_over);
      b_hi = w_over != 0xffffffffu ? b_lo + w_over : wpos_lower_bound(mini_wpos, bucket_first, bb, nb, i_max + 1u);
    }
    if (b_lo >= b_hi) continue;
    int32_t c_best = -1;
    uint32_t c_first = 0, c_last = 0;
    const uint32_t n_groups = (b_hi - b_lo + 63u) / 64u;
    for (uint32_t g = 0; g < n_groups && !overflow; ++g) {
      const uint32_t sb = b_lo + g * 64u, b = sb + lane;
      const bool has = b < b_hi;
      const uint32_t wp = has ? mini_wpos[b] : 0u;
      const uint32_t wp_next = (has && b + 1u < m1) ? mini_wpos[b + 1u] : 0xffffffffu;
      const uint32_t w_end = min(wp_next - 1u, i_max) + count_windows;  // window ids below this: the begin's widest window
      int32_t bar = c_best > best_shared ? c_best : best_shared;
      if (bar < floor_bar) bar = floor_bar;
      uint32_t held = 0;  // seed hits inside the begin's widest window: no window of the begin shares more
      for (uint32_t i = 0; i < nh; ++i) held += (s_hc[i] == c && s_hw[i] >= wp && s_hw[i] < w_end) ? 1u : 0u;
      const bool pending = has && (int32_t)held >= bar;
      if (!__any(pending)) continue;
      const uint32_t first_lane = (uint32_t)__builtin_ctzll(__ballot(pending));
      const uint32_t base = sb + first_lane, n = min(m1 - base, kSparseCap);
      const uint32_t wbase = mini_wpos[base];
      __syncthreads();
      uint32_t hh[kPer];
      bool any_dup = false;
#pragma unroll
      for (int q = 0; q < kPer; ++q) {
        const uint32_t x = (uint32_t)q * 64u + lane;
        const bool in = x < n;
        const uint32_t t = base + min(x, n - 1u);
        const uint32_t h = mini_hash[t], dw = mini_wpos[t] - wbase;
        const int32_t pv = prev_same[t];
        hh[q] = in ? h : 0xffffffffu;
        any_dup = any_dup || (in && pv >= (int32_t)base);
        s_refw[x] = (uint16_t)((dw > 0xfffeu || !in) ? 0xffffu : dw);
      }
      if (__any(any_dup)) { overflow = true; break; }  // a hash twice in the stretch: the windows' distinct hashes are the general kernel's to count
      __syncthreads();
      const bool at_end = base + n == m1;
      auto first_at_or_after = [&](uint32_t lo, uint32_t target) -> uint32_t {
        uint32_t hi = n;
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if ((uint32_t)s_refw[mid] < target) lo = mid + 1; else hi = mid; }
        return lo;
      };
      const bool lane_on = pending && lane >= first_lane;
      uint32_t xs = 0, xe_lo = 0, xe_hi = 0;
      if (lane_on) {
        xs = b - base;
        xe_lo = first_at_or_after(min(xs, n), wp + count_windows - wbase);  // (a target past 0xfffe finds nothing: the window runs past the stretch)
        xe_hi = first_at_or_after(xe_lo, w_end - wbase);
      }
      if (__any(lane_on && !(xe_hi < n || at_end))) { overflow = true; break; }  // a window runs past the stretch
      // where the hits sit in the stretch
      if (lane < nh) {
        uint32_t p = 0xffffffffu;
        if (s_hc[lane] == c && s_hw[lane] >= wbase && s_hw[lane] - wbase < 0xffffu) {
          const uint32_t x = first_at_or_after(0u, s_hw[lane] - wbase);
          if (x < n && (uint32_t)s_refw[x] == s_hw[lane] - wbase) p = x;
        }
        s_pos[lane] = p;
      }
      // per hit: the stretch entries that are reference-only (no hit's hash) and below the hit's hash, as bits, with the
      // counts of the bits before each word
      uint32_t is_match = 0;
      for (uint32_t i = 0; i < nh; ++i) {
        const uint32_t ph = s_ph[i];
#pragma unroll
        for (int q = 0; q < kPer; ++q) is_match |= (hh[q] == ph ? 1u : 0u) << q;
      }
      for (uint32_t i = 0; i < nh; ++i) {
        const uint32_t ph = s_ph[i];
#pragma unroll
        for (int q = 0; q < kPer; ++q) {
          const uint64_t mb = __ballot(((uint32_t)q * 64u + lane < n) & (hh[q] < ph) & (((is_match >> q) & 1u) == 0u));
          if (lane == 0) { s_B[i][2 * q] = (uint32_t)mb; s_B[i][2 * q + 1] = (uint32_t)(mb >> 32); }
        }
      }
      __syncthreads();
      for (uint32_t i = 0; i < nh; ++i) {
        const uint32_t pc = lane < kWords ? (uint32_t)__popc(s_B[i][lane]) : 0u;
        const uint32_t inc = pa_dev::wave_incl_scan_dpp(pc);
        if (lane < kWords) s_P[i][lane] = inc - pc;
        if (lane == kWords - 1u) { s_P[i][kWords] = inc; s_B[i][kWords] = 0u; }
      }
      __syncthreads();
      // every state of every pending begin: the window holds the stretch entries [xs, e), e from the begin's first end to its last
      int32_t lane_best = -1;
      const uint32_t most = pa_dev::wave_max_dpp(lane_on ? xe_hi - xe_lo + 1u : 0u);
      for (uint32_t t = 0; t < most; ++t) {
        const uint32_t e = xe_lo + t;
        if (!(lane_on && e <= xe_hi)) continue;
        int32_t shared = 0;
        for (uint32_t i = 0; i < nh; ++i) {
          const uint32_t p = s_pos[i];
          if (!(p != 0xffffffffu && p >= xs && p < e)) continue;
          const uint32_t we = e >> 5, ws = xs >> 5;
          const uint32_t ce_ = s_P[i][we] + (uint32_t)__popc(s_B[i][we] & ((1u << (e & 31u)) - 1u));
          const uint32_t cs_ = s_P[i][ws] + (uint32_t)__popc(s_B[i][ws] & ((1u << (xs & 31u)) - 1u));
          shared += (s_hr[i] + (ce_ - cs_) < s) ? 1 : 0;
        }
        lane_best = max(lane_best, shared);
      }
      // fold: most shared; window ids of the first and of the last begin that has it (a state's position is its begin's)
      const int32_t group_best = (int32_t)pa_dev::wave_max_dpp(lane_on ? (uint32_t)(lane_best + 1) : 0u) - 1;
      if (__any(lane_on) && group_best >= c_best) {
        const uint64_t top = __ballot(lane_on && lane_best == group_best);
        const uint32_t w_first = __shfl(wp, __builtin_ctzll(top), 64), w_last = __shfl(wp, 63 - __builtin_clzll(top), 64);
        c_last = group_best > c_best ? w_last : max(c_last, w_last);
        c_first = group_best > c_best ? w_first : min(c_first, w_first);
        c_best = group_best;
      }
    }
    if (overflow) break;
    if (c_best >= 0 && c_best >= best_shared) { best_shared = c_best; best_c = c; best_first = c_first; best_last = c_last; }  // of equals, the last
  }
  if (overflow) {  // (rare: one draw per such segment)
    if (lane == 0) {
      const uint32_t slot = atomicAdd(over_n, 1u);
      over_a0[slot] = a0; over_nh[slot] = nh; over_f[slot] = f;
    }
    return;
  }
  if (lane == 0 && best_shared >= 0 && (uint32_t)best_shared >= tab_min_shared[s]) {
    const uint64_t jq = ((uint64_t)best_shared << 30) / s;
    const unsigned long long packed = ((unsigned long long)jq << 32) | ((unsigned long long)best_shared << 16) | s;
    const uint64_t bin = contig_bin_off[best_c] + (best_first + best_last) / 2u / (frag_len - 20u);
    atomicMax(&table[(uint64_t)frag_genome_local[f] * table_stride + bin], packed);
  }
}

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
      contig_bucket_off, bucket_first, post_g, seg_rec, hash_cut, q_cut, post_cw2, post_g2, run_hist, long_runs, frag_d, uniq_hash, lookup_at, seg_f, seg2_f, amb_pos, amb_byte, seg_over;
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
  uint64_t index_key_room = 0;              // keys of one half of the sort's key buffer (W.keys[0] holds both halves)
  int index_which = 0;
  ~FragWork() {
    DevBuf *all[] = {&contig_start, &contig_len, &contig_genome, &block_counts, &block_offsets, &mini_hash, &mini_wpos,
                     &mini_contig, &contig_mini_off, &keys[0], &keys[1], &vals[0], &vals[1], &flags, &mini_id,
                     &post_start, &prev_same, &sorted_idx, &frag_contig, &frag_no, &frag_genome_local, &q_hash, &q_pos,
                     &q_id, &q_s, &hit_count, &hit_off, &hkeys[0], &hkeys[1], &hvals[0], &hvals[1], &seg_start,
                     &tab_min_hits, &tab_min_shared, &ident_tab, &contig_bin_off, &genome_bin_off, &table, &matched,
                     &ident_sum, &scalars, &run_g, &seg_list, &seg2_a0, &seg2_nh, &post_cw, &seg_a0, &seg_nh, &genome_first_contig, &contig_bucket_off, &bucket_first, &post_g, &seg_rec, &hash_cut, &q_cut, &post_cw2, &post_g2, &run_hist, &long_runs, &frag_d, &uniq_hash, &lookup_at, &seg_f, &seg2_f, &amb_pos, &amb_byte, &seg_over};
    for (DevBuf *b : all) b->release();
  }
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
                   uint32_t n_contigs, int w, uint32_t *m_out) {
  const uint32_t blocks = ceil_div_u64(arena_bases, kOwn);
  PA_TRY(W.block_counts.reserve((uint64_t)blocks * 8));  // the look-back words of minimizer_kernel
  PA_TRY(W.scalars.reserve(64));
  // expected density of winnowed minimizers is 2 / (w + 1); the arrays are sized a quarter above that and the run is
  // repeated with the exact size should a low-complexity data set need more
  uint64_t cap = (uint64_t)((double)arena_bases * 2.5 / (double)(w + 1)) + (1u << 20);
  if (const char *v = PA_TOOL_ENV("PA_FRAGANI_MINIMIZER_ROOM")) cap = std::max<uint64_t>(1, strtoull(v, nullptr, 10));  // tests: force the repeat
  for (int attempt = 0; attempt < 2; ++attempt) {
    PA_REQUIRE(cap < (1ULL << 31), "fragment ANI: room for %llu minimizers exceeds the 31-bit index space", (unsigned long long)cap);
    PA_TRY(W.mini_hash.reserve(cap * 4 + 16));
    PA_TRY(W.mini_wpos.reserve(cap * 4 + 16));
    PA_TRY(W.mini_contig.reserve(cap * 4 + 16));
    PA_HIP(hipMemsetAsync(W.block_counts.p, 0, (uint64_t)blocks * 8, c->stream));
    PA_HIP(hipMemsetAsync(W.scalars.p, 0, 16, c->stream));
    hipLaunchKernelGGL((minimizer_kernel<K>), dim3(blocks), dim3(kThreads), 0, c->stream,
                       d_packed, d_mask, arena_bases, W.contig_start.as<uint64_t>(), W.contig_len.as<uint32_t>(), n_contigs, w,
                       W.block_counts.as<unsigned long long>(), W.scalars.as<uint32_t>(), (uint32_t)cap,
                       W.mini_hash.as<uint32_t>(), W.mini_wpos.as<uint32_t>(), W.mini_contig.as<uint32_t>(), blocks, W.ambiguous(d_packed));
    PA_HIP(hipGetLastError());
    PA_HIP(hipMemcpyAsync(c->h_pinned, W.scalars.p, 16, hipMemcpyDeviceToHost, c->stream));
    PA_HIP(hipStreamSynchronize(c->stream));
    const uint32_t *h = reinterpret_cast<const uint32_t *>(c->h_pinned);
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
                        uint32_t n_contigs, uint32_t k, int w, uint32_t *m_out) {
  switch (k) {
    case 8: return run_minimizers<8>(c, W, d_packed, d_mask, arena_bases, n_contigs, w, m_out);
    case 9: return run_minimizers<9>(c, W, d_packed, d_mask, arena_bases, n_contigs, w, m_out);
    case 10: return run_minimizers<10>(c, W, d_packed, d_mask, arena_bases, n_contigs, w, m_out);
    case 11: return run_minimizers<11>(c, W, d_packed, d_mask, arena_bases, n_contigs, w, m_out);
    case 12: return run_minimizers<12>(c, W, d_packed, d_mask, arena_bases, n_contigs, w, m_out);
    case 13: return run_minimizers<13>(c, W, d_packed, d_mask, arena_bases, n_contigs, w, m_out);
    case 14: return run_minimizers<14>(c, W, d_packed, d_mask, arena_bases, n_contigs, w, m_out);
    case 15: return run_minimizers<15>(c, W, d_packed, d_mask, arena_bases, n_contigs, w, m_out);
    case 16: return run_minimizers<16>(c, W, d_packed, d_mask, arena_bases, n_contigs, w, m_out);
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

int pa_fragani_ex(pa_ctx *c, const uint32_t *d_packed, const uint32_t *d_mask, uint64_t arena_bases,
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
  W.index_valid = false;  // until this call has passed stage 2 (or taken it over)
  prof.emplace(c, PA_PROF_FRAG_INDEX);
  uint32_t m = reuse ? W.index_m : 0;
  if (!reuse) {
  PA_TRY(dispatch_minimizers(c, W, d_packed, d_mask, arena_bases, n_contigs, k, w, &m));
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

  // ---- fragments and reference bins (host bookkeeping)
  std::vector<uint32_t> frag_contig, frag_no, genome_frag_off(n_genomes + 1, 0), contig_bin_off(n_contigs + 1, 0),
      genome_bin_off(n_genomes + 1, 0);
  for (uint32_t g = 0; g < n_genomes; ++g) h_total_frags[g] = 0;
  for (uint32_t ci = 0; ci < n_contigs; ++ci) {
    const uint32_t nf = h_contig_len[ci] / frag_len;
    for (uint32_t f = 0; f < nf; ++f) { frag_contig.push_back(ci); frag_no.push_back(f); }
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
  const uint32_t bin_base = genome_bin_off[ref0];
  const uint64_t range_bins = std::max<uint32_t>(genome_bin_off[ref1] - bin_base, 1u);  // bins of the reference genomes asked for
  const uint32_t n_frags = (uint32_t)frag_contig.size();
  for (uint64_t i = (uint64_t)qry0 * out_cols; i < (uint64_t)qry1 * out_cols; ++i) { h_matched[i] = 0; h_ident_sum[i] = 0.0; }
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
      hipLaunchKernelGGL(postings_kernel, dim3(gm), dim3(kThreads), 0, c->stream, keys[which], vals[which], d_flags, d_pos, md,
                         n_ids, W.mini_contig.as<uint32_t>(), W.mini_id.as<uint32_t>(), W.post_start.as<uint32_t>(),
                         W.prev_same.as<int32_t>(), W.mini_wpos.as<uint32_t>(), W.contig_genome.as<uint32_t>(),
                         W.post_cw.as<uint64_t>(), W.post_g.as<uint16_t>(), W.contig_mini_off.as<uint32_t>(), n_contigs,
                         W.uniq_hash.as<uint32_t>());
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
    std::vector<uint32_t> mh(kQMax + 1), ms(kQMax + 1);
    PA_TRY(pa_fragani_tables(k, kQMax, mh.data(), ms.data()));
    std::vector<float> ident((uint64_t)(kQMax + 1) * (kQMax + 1), 0.0f);
    for (uint32_t s = 1; s <= (uint32_t)kQMax; ++s)
      for (uint32_t x = 0; x <= s; ++x) ident[(uint64_t)s * (kQMax + 1) + x] = (float)pa_fragani_identity(x, s, k);
    PA_TRY(upload(c, W.tab_min_hits, mh));
    PA_TRY(upload(c, W.tab_min_shared, ms));
    PA_TRY(upload(c, W.ident_tab, ident));
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
  uint32_t batch_frags = 1u << 17;
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
                       W.q_cut.as<uint32_t>(), restricted ? W.lookup_at.as<uint4>() : nullptr, W.index_lookup_bits);
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
        PA_TRY(W.seg_rec.reserve((uint64_t)count * 32));
        hipLaunchKernelGGL(segment_records_kernel, dim3(ceil_div_u64(count, kThreads)), dim3(kThreads), 0, c->stream, hk[hw],
                           list_a0, list_nh, list_f, count, W.q_s.as<uint32_t>(), W.q_cut.as<uint32_t>(), W.tab_min_hits.as<uint32_t>(),
                           W.contig_genome.as<uint32_t>(), W.genome_first_contig.as<uint32_t>(), W.seg_rec.as<uint4>());
#define PA_MAP_CASE(CAP)                                                                                                  \
  case CAP:                                                                                                               \
    hipLaunchKernelGGL((map_segments_kernel<CAP, kAll>), dim3(count), dim3(64), eval_lds_bytes(s_cap, hit_cap, CAP), c->stream,  \
                       hk[hw], hv[hw], W.seg_rec.as<uint4>(), count, presorted, q_hash_p,                                  \
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
          if (const char *v = PA_TOOL_ENV("PA_FRAGANI_SPARSE")) sparse_max = atoi(v) ? kSparseHits : 0u;  // tests: 0 = every segment through the general kernel
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
                               d_over_n);
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
                        "%u states tying their candidate's optimum\n",
                st[0], st[1], st[2], st[3], st[4], st[5], st[6], st[7], st[8], st[9], st[10], st[11], st[12], st[13], st[14], st[15], st[16], st[17],
                st[18], st[19], st[20], st[21], st[22], st[23], st[24], st[25], st[26], st[27], st[28], st[29], st[30], st[31], st[32], st[33]);
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
