// kmer_hash.hip -- kernel 1 of the sketch phase (gfx950).
//
// Replaces the k-mer loop inside `sourmash scripts singlesketch`
// (call site pyani_plus/methods/sourmash.py:67-83): for every window of K valid
// bases, hash the canonical k-mer (upper-case ASCII) with MurmurHash3_x64_128,
// seed 42, keep the first word when it is <= max_hash.
//
// Work decomposition: one thread per 64-base arena block (one 16-byte packed
// load, perfectly coalesced), plus a 32-base look-back into the previous block.
// The invalid-position mask is read only by the blocks the arena's `dirty`
// bitmap flags (one bit per block: the block or the 32 positions before it hold
// an invalid position) -- a handful per genome -- so the kernel streams the
// packed bases and nothing else.
// The kernel is integer-VALU bound (8 arithmetic 64-bit multiplies + ~60 other
// VALU per window against 0.25 byte of HBM input; VALUBusy 105 % by counters), so
// there is no LDS tiling of the input; LDS holds the first-multiply tables and
// stages the rare survivors (1 in `scaled`) so that the global append is one
// atomic per workgroup and the stores are coalesced.
#include <cstdlib>
#include <type_traits>

#include "pa_internal.h"
#include "murmur_dev.h"

namespace {

using namespace pa_dev;

constexpr int kThreads = 256;
constexpr uint32_t kStageCap = 1024;  // LDS staging slots per workgroup

constexpr int pow2_floor(int k) { int c = 1; while (2 * c <= k) c *= 2; return c; }

// largest g with genome_blk[g] <= blk   (genome_blk has n+1 ascending entries)
__device__ __forceinline__ uint32_t find_genome(const uint32_t *__restrict__ genome_blk, uint32_t n, uint32_t blk) {
  uint32_t lo = 0, hi = n;  // invariant: genome_blk[lo] <= blk < genome_blk[hi]
  while (hi - lo > 1) {
    uint32_t mid = (lo + hi) >> 1;
    if (genome_blk[mid] <= blk) lo = mid; else hi = mid;
  }
  return lo;
}

// one candidate into the region of genome g (slow path: one atomic per candidate)
__device__ __forceinline__ void region_append(uint64_t *__restrict__ regions, const uint64_t *__restrict__ region_off,
                                              uint32_t *__restrict__ cursor, uint32_t *__restrict__ overflow,
                                              uint32_t g, uint64_t h) {
  const uint64_t slot = atomicAdd(&cursor[g], 1u);
  if (slot < region_off[g + 1] - region_off[g]) regions[region_off[g] + slot] = h; else *overflow = 1u;
}

// LUT = true (default): the first multiply of every murmur word is looked up.  A word is
// two 4-base groups (lo, hi); word*c = (ascii(lo) + ascii(hi)*2^32)*c
//                                    = s_lo[j][lo] + (s_hi[j][hi] << 32)   (mod 2^64)
// with s_lo[j][b] = ascii(b)*c_j (64 bit) and s_hi[j][b] = low32(ascii(b)*c_j): two LDS reads and
// one add replace the ASCII expansion (8 VALU) and a 64-bit multiply (3 quarter-rate VALU).
// LUT = false keeps the arithmetic form (ablation / cross-check, PA_KMER_VARIANT=0).
template <int K, bool LUT>
__global__ __launch_bounds__(kThreads) void kmer_hash_kernel(
    const uint4 *__restrict__ packed, const uint2 *__restrict__ mask, const uint64_t *__restrict__ dirty, uint32_t n_blocks64,
    const uint32_t *__restrict__ genome_blk, uint32_t n_genomes, uint64_t max_hash,
    uint64_t *__restrict__ cand_hash, uint32_t *__restrict__ cand_genome, uint64_t cap,
    unsigned long long *__restrict__ count, const uint64_t *__restrict__ region_off, uint32_t *__restrict__ cursor,
    uint32_t *__restrict__ overflow, uint32_t blk0) {
  static_assert(K >= 1 && K <= 32, "a k-mer is one 64-bit register pair");
  __shared__ uint64_t s_hash[kStageCap];
  __shared__ uint32_t s_blk[kStageCap];
  __shared__ uint32_t s_n;
  __shared__ unsigned long long s_base;
  constexpr int kWords = (K + 7) / 8;
  __shared__ uint64_t s_lo[LUT ? kWords : 1][256];
  __shared__ uint32_t s_hi[LUT ? kWords : 1][256];

  const uint32_t tid = threadIdx.x;
  const uint32_t max_hi = (uint32_t)(max_hash >> 32);
  const bool take_all = max_hi == 0xffffffffu;  // scaled = 1: no screen
  const uint32_t screen_hi = max_hi + 1u;
  if (tid == 0) s_n = 0;
  if constexpr (LUT) {
#pragma unroll
    for (int j = 0; j < kWords; ++j) {
      const uint64_t cj = (j & 1) ? kC2 : kC1;
      s_lo[j][tid] = (uint64_t)ascii_group(tid, K - 8 * j) * cj;
      s_hi[j][tid] = (uint32_t)((uint64_t)ascii_group(tid, K - 8 * j - 4) * cj);
    }
  }
  __syncthreads();

  const uint32_t t = blk0 + blockIdx.x * kThreads + tid;  // blk0: first arena block of this launch (streamed uploads)
  if (t < n_blocks64) {
    const uint4 cur = packed[t];
    uint2 pw = make_uint2(0u, 0u);
    if (t > 0) pw = reinterpret_cast<const uint2 *>(packed)[2 * (uint64_t)t - 1];  // bases -32..-1
    // ---- which windows are usable.  Clean blocks (all but a few per genome) have none to exclude and never
    // touch the mask; the dirty word of a wave's 64 blocks is one uniform load (blk0 is a multiple of 64).
    uint32_t d1 = 0u, d2 = 0u;
    const uint32_t wave_blk = __builtin_amdgcn_readfirstlane(t & ~63u);
    const uint64_t dirty_word = dirty[wave_blk >> 6];
    if ((dirty_word >> (t & 63u)) & 1ull) {
      // dilate the invalid-position bitset by K: bit i of (d0,d1,d2) <-> base i-32; the window ending at base e
      // is bad iff any of bases e-K+1..e is invalid, i.e. bit (32+e) of OR_{s<K} (M << s).
      const uint2 m = mask[t];
      uint32_t d0 = t > 0 ? mask[t - 1].y : 0xffffffffu;
      d1 = m.x;
      d2 = m.y;
      int c = 1;
#pragma unroll
      for (; 2 * c <= K; c *= 2) {
        d2 |= alignbit(d2, d1, 32 - c);
        d1 |= alignbit(d1, d0, 32 - c);
        d0 |= d0 << c;
      }
      constexpr int cc = pow2_floor(K);  // width covered by the doubling rounds
      constexpr int r = K - cc;
      if constexpr (r > 0) {
        d2 |= alignbit(d2, d1, 32 - r);
        d1 |= alignbit(d1, d0, 32 - r);
      }
    }
    const uint32_t bad[2] = {d1, d2};
    if ((d1 & d2) != 0xffffffffu) {  // at least one usable window in this block
      constexpr uint64_t kMask = (K == 32) ? ~0ULL : ((1ULL << (2 * K)) - 1);
      constexpr uint32_t kMaskHi = (uint32_t)(kMask >> 32);
      // Two packed streams over the 96 positions this thread sees (32 of look-back + its 64): F, the
      // bases as stored (base j at bits 2j: already the byte order murmur wants), and R, their reverse
      // complement (reverse the order of the 2-bit groups, flip the bits).  The window ending at block
      // position e is K groups of F starting at group e + 33 - K and, as its reverse complement, K groups of
      // R starting at group 63 - e.  Both are funnel shifts -- no rolling registers, no warm-up over the
      // K-1 preceding bases, no dependency from one window to the next.  And because
      // the MSB-first (lexicographic) form of one strand is the complement of the LSB-first form of the
      // other, "forward <= reverse complement" in lexicographic order is simply F-window <= R-window.
      uint32_t fw[8] = {pw.x, pw.y, cur.x, cur.y, cur.z, cur.w, 0u, 0u};
      uint32_t rw[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const uint32_t x = __builtin_bitreverse32(fw[5 - j]);  // groups reversed, bits inside a group too
        rw[j] = ~(((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1));
      }
      // Windows are taken column by column: the four windows e = i, i+16, i+32, i+48 sit at the same bit offset of
      // consecutive dwords, so the five funnel shifts A[0..4] of the forward stream (and five of the reverse one) serve
      // all four -- the high dword of one window is the low dword of the next: 2.5 shifts per window and stream pair
      // instead of 4.  The shift amounts are uniform run-time values; the dword a column starts in changes once over
      // the 16 columns (where i + 33 - K crosses a multiple of 16), hence the two loops with a constant base each.
      auto columns = [&](auto base_tag, int i_begin, int i_end) {
        constexpr int kBase = decltype(base_tag)::value;
#pragma unroll 1
        for (int i = i_begin; i < i_end; ++i) {
          constexpr int kOff = 33 - K;
          const uint32_t fo = 2u * (uint32_t)((i + kOff) & 15), ro = 30u - 2u * (uint32_t)i;
          uint32_t A[5], B[5];
#pragma unroll
          for (int j = 0; j < 5; ++j) {
            A[j] = alignbit(fw[kBase + j + 1], fw[kBase + j], fo);  // a shift of 0 returns the low operand
            B[j] = alignbit(rw[j + 1], rw[j], ro);
          }
#pragma unroll
          for (int wi = 0; wi < 4; ++wi) {
            const uint32_t f_lo = A[wi], f_hi = A[wi + 1], r_lo = B[3 - wi], r_hi = B[4 - wi];
          const uint64_t f_lsb = u64_of(f_lo, f_hi) & kMask, r_lsb = u64_of(r_lo, r_hi) & kMask;
          // Canonical strand.  Left to itself hipcc compares into VCC and selects the two words with VCC-reading
          // v_cndmask; on gfx950 the second VCC read of such a pair is pathologically slow (36 cycles for the group
          // against 13 with the lane mask in an ordinary SGPR pair: profiles/r02_ubench_valu_gfx950.txt), so the
          // compare goes to an SGPR pair and both selects read that.  s_nop 1: two wait states between a VALU
          // write of an SGPR and a VALU read of it.
          const uint64_t fwd = __builtin_amdgcn_uicmpl(f_lsb, r_lsb, 37 /* ICMP_ULE */);
          uint32_t clo, chi;
          asm("s_nop 1\n\tv_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(clo) : "v"((uint32_t)r_lsb), "v"((uint32_t)f_lsb), "s"(fwd));
          asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(chi) : "v"((uint32_t)(r_lsb >> 32)), "v"((uint32_t)(f_lsb >> 32)), "s"(fwd), "v"(clo));
          uint64_t P[4] = {0, 0, 0, 0};
#pragma unroll
          for (int j = 0; j < kWords; ++j) {
            const uint32_t src = j < 2 ? clo : chi;
            const uint32_t glo = (src >> (16 * (j & 1))) & 0xffu, ghi = (src >> (16 * (j & 1) + 8)) & 0xffu;
            if constexpr (LUT) {
              const uint64_t lo = s_lo[j][glo];
              const uint32_t hi = (uint32_t)(lo >> 32) + s_hi[j][ghi];
              P[j] = u64_of((uint32_t)lo, hi);
            } else {
              const uint64_t word = u64_of(ascii_group(glo, K - 8 * j), ascii_group(ghi, K - 8 * j - 4));
              P[j] = word * ((j & 1) ? kC2 : kC1);
            }
          }
          (void)kMaskHi;
          // Screen on the high words: hash = (X ^ X>>33) + (Y ^ Y>>33) has high word X.hi + Y.hi + carry, so it can
          // be <= max_hash only if X.hi + Y.hi is <= max_hash.hi or is 0xffffffff (the carry wraps it to 0) -- one
          // unsigned compare of X.hi + Y.hi + 1 against max_hash.hi + 1.  The sum of the two high words is linear in
          // the inputs of the last multiply (murmur_dev.h), so the 999 in 1000 windows that are dropped never form
          // X and Y: 2 mul_hi + 2 mul_lo + 4 adds + 1 compare instead of two 64-bit multiplies, an add and 2 compares.
          uint64_t U, V;
          murmur3_pre_last_mul<K>(P, U, V);
          if (take_all || last_mul_high_sum_plus1(U, V) <= screen_hi) {
            const uint64_t X = U * kF2, Y = V * kF2;
            const uint64_t h = (X ^ (X >> 33)) + (Y ^ (Y >> 33));
            if (h > max_hash || ((bad[wi >> 1] >> (16 * (wi & 1) + i)) & 1u)) continue;  // validity is only looked at for the 1 in 1000
            const uint32_t slot = atomicAdd(&s_n, 1u);
            if (slot < kStageCap) {
              s_hash[slot] = h;
              s_blk[slot] = t;
            } else if (region_off) {  // staging full (tiny `scaled`): append straight to the genome's region
              region_append(cand_hash, region_off, cursor, overflow, find_genome(genome_blk, n_genomes, t), h);
            } else {  // ... or to the global candidate list
              const unsigned long long g = atomicAdd(count, 1ULL);
              if (g < cap) {
                cand_hash[g] = h;
                cand_genome[g] = find_genome(genome_blk, n_genomes, t);
              }
            }
          }
          }
        }
      };
      constexpr int kOff0 = 33 - K, kSplit = (kOff0 & 15) ? 16 - (kOff0 & 15) : 16;
      columns(std::integral_constant<int, (kOff0 >> 4)>{}, 0, kSplit);
      if constexpr (kSplit < 16) columns(std::integral_constant<int, (kOff0 >> 4) + 1>{}, kSplit, 16);
    }
  }
  __syncthreads();
  const uint32_t n = min(s_n, kStageCap);
  if (n == 0) return;
  if (region_off) {
    // per-genome regions (see sketch_lds.hip): a workgroup that lies inside one genome -- all but the
    // few that straddle a boundary -- reserves its slots with one atomic on that genome's cursor
    const uint32_t b0 = blk0 + blockIdx.x * kThreads;
    const uint32_t b1 = min(b0 + (uint32_t)kThreads, n_blocks64) - 1u;
    const uint32_t g0 = find_genome(genome_blk, n_genomes, b0);
    if (genome_blk[g0 + 1] > b1) {
      if (tid == 0) s_base = atomicAdd(&cursor[g0], n);
      __syncthreads();
      const uint64_t room = region_off[g0 + 1] - region_off[g0];
      uint64_t *__restrict__ dst = cand_hash + region_off[g0];
      for (uint32_t i = tid; i < n; i += kThreads) {
        const uint64_t slot = s_base + i;
        if (slot < room) dst[slot] = s_hash[i]; else *overflow = 1u;
      }
    } else {
      for (uint32_t i = tid; i < n; i += kThreads)
        region_append(cand_hash, region_off, cursor, overflow, find_genome(genome_blk, n_genomes, s_blk[i]), s_hash[i]);
    }
    return;
  }
  if (tid == 0) s_base = atomicAdd(count, (unsigned long long)n);
  __syncthreads();
  const unsigned long long base = s_base;
  for (uint32_t i = tid; i < n; i += kThreads) {
    const unsigned long long g = base + i;
    if (g < cap) {
      cand_hash[g] = s_hash[i];
      cand_genome[g] = find_genome(genome_blk, n_genomes, s_blk[i]);
    }
  }
}

template <int K, bool LUT>
int launch_variant(pa_ctx *c, const uint32_t *d_packed, const uint32_t *d_mask, const uint64_t *d_dirty, uint64_t n_blocks64,
           const uint32_t *d_genome_blk, uint32_t n_genomes, uint64_t max_hash, uint64_t *d_cand_hash,
           uint32_t *d_cand_genome, uint64_t cap, uint64_t *d_count, const uint64_t *d_region_off, uint32_t *d_cursor,
           uint32_t *d_overflow, uint64_t blk0, hipStream_t stream) {
  const uint32_t grid = ceil_div_u64(n_blocks64 - blk0, kThreads);
  hipLaunchKernelGGL((kmer_hash_kernel<K, LUT>), dim3(grid), dim3(kThreads), 0, stream ? stream : c->stream,
                     reinterpret_cast<const uint4 *>(d_packed), reinterpret_cast<const uint2 *>(d_mask), d_dirty,
                     (uint32_t)n_blocks64, d_genome_blk, n_genomes, max_hash, d_cand_hash, d_cand_genome, cap,
                     reinterpret_cast<unsigned long long *>(d_count), d_region_off, d_cursor, d_overflow, (uint32_t)blk0);
  PA_HIP(hipGetLastError());
  return PA_OK;
}

template <int K>
int launch(pa_ctx *c, const uint32_t *d_packed, const uint32_t *d_mask, const uint64_t *d_dirty, uint64_t n_blocks64,
           const uint32_t *d_genome_blk, uint32_t n_genomes, uint64_t max_hash, uint64_t *d_cand_hash,
           uint32_t *d_cand_genome, uint64_t cap, uint64_t *d_count, const uint64_t *d_region_off, uint32_t *d_cursor,
           uint32_t *d_overflow, uint64_t blk0, hipStream_t stream) {
  static const bool arithmetic = [] {
    const char *v = PA_TOOL_ENV("PA_KMER_VARIANT");
    return v && v[0] == '0';
  }();
  if constexpr (K == 31)  // the ablation build exists for the benchmarked k only
    if (arithmetic)
      return launch_variant<K, false>(c, d_packed, d_mask, d_dirty, n_blocks64, d_genome_blk, n_genomes, max_hash, d_cand_hash,
                                    d_cand_genome, cap, d_count, d_region_off, d_cursor, d_overflow, blk0, stream);
  return launch_variant<K, true>(c, d_packed, d_mask, d_dirty, n_blocks64, d_genome_blk, n_genomes, max_hash, d_cand_hash,
                                 d_cand_genome, cap, d_count, d_region_off, d_cursor, d_overflow, blk0, stream);
}

// ---- k from 33 to 64, sixty-four windows per thread --------------------------------------------------------------
// The structure of kmer_hash_kernel carried over to k-mers of up to 128 bits: one thread per 64-position block with a
// 64-position look-back (the whole previous block), the forward stream and the reverse-complement stream as ten dwords
// each, windows taken column by column (seven funnel shifts per stream serve the four windows e = i, i + 16, i + 32,
// i + 48), the canonical strand from a 128-bit compare whose three lane masks are combined on the scalar unit, up to
// eight first products from the LDS tables, the high-word screen.  Validity: the invalid-position bitset of the 128
// positions is dilated by K; a block is skipped as clean only if neither its own dirty bit nor the previous block's is
// set (the bitmap's "or the 32 positions before" reaches back 96 positions that way, the windows 63).
template <int K>
__global__ __launch_bounds__(kThreads) void kmer_hash_wide_kernel(
    const uint4 *__restrict__ packed, const uint2 *__restrict__ mask, const uint64_t *__restrict__ dirty, uint32_t n_blocks64,
    const uint32_t *__restrict__ genome_blk, uint32_t n_genomes, uint64_t max_hash,
    uint64_t *__restrict__ cand_hash, uint32_t *__restrict__ cand_genome, uint64_t cap,
    unsigned long long *__restrict__ count, const uint64_t *__restrict__ region_off, uint32_t *__restrict__ cursor,
    uint32_t *__restrict__ overflow, uint32_t blk0) {
  static_assert(K >= 33 && K <= 64, "k-mers of 65 to 128 bits");
  __shared__ uint64_t s_hash[kStageCap];
  __shared__ uint32_t s_blk[kStageCap];
  __shared__ uint32_t s_n;
  __shared__ unsigned long long s_base;
  constexpr int kWords = (K + 7) / 8;
  __shared__ uint64_t s_lo[kWords][256];
  __shared__ uint32_t s_hi[kWords][256];
  const uint32_t tid = threadIdx.x;
  const uint32_t max_hi = (uint32_t)(max_hash >> 32);
  const bool take_all = max_hi == 0xffffffffu;
  const uint32_t screen_hi = max_hi + 1u;
  if (tid == 0) s_n = 0;
#pragma unroll
  for (int j = 0; j < kWords; ++j) {
    const uint64_t cj = (j & 1) ? kC2 : kC1;
    s_lo[j][tid] = (uint64_t)ascii_group(tid, K - 8 * j) * cj;
    s_hi[j][tid] = (uint32_t)((uint64_t)ascii_group(tid, K - 8 * j - 4) * cj);
  }
  __syncthreads();

  const uint32_t t = blk0 + blockIdx.x * kThreads + tid;
  if (t < n_blocks64) {
    const uint4 cur = packed[t];
    const uint4 prev = t > 0 ? packed[t - 1] : make_uint4(0u, 0u, 0u, 0u);
    // ---- which windows are usable (bit i of d0..d3 <-> position i - 64)
    uint32_t d2 = 0u, d3 = 0u;
    const uint32_t wave_blk = __builtin_amdgcn_readfirstlane(t & ~63u);
    const uint64_t dirty_word = dirty[wave_blk >> 6];
    const uint64_t before = wave_blk ? dirty[(wave_blk >> 6) - 1] >> 63 : 1ull;  // the block before this wave's first
    const uint64_t dirty_or_prev = dirty_word | (dirty_word << 1) | before;
    if ((dirty_or_prev >> (t & 63u)) & 1ull) {
      const uint2 m = mask[t];
      uint32_t d0 = 0xffffffffu, d1 = 0xffffffffu;
      if (t > 0) { const uint2 mp = mask[t - 1]; d0 = mp.x; d1 = mp.y; }
      d2 = m.x;
      d3 = m.y;
#pragma unroll
      for (int c = 1; c <= 16; c *= 2) {  // OR of the bitset with itself moved up by 1, 2, 4, 8, 16: width 32
        d3 |= alignbit(d3, d2, 32 - c);
        d2 |= alignbit(d2, d1, 32 - c);
        d1 |= alignbit(d1, d0, 32 - c);
        d0 |= d0 << c;
      }
      if constexpr (K == 64) {  // width 64: one whole word further
        d3 |= d2; d2 |= d1; d1 |= d0;
      } else {
        constexpr int r = K - 32;  // 1 .. 31: the remaining width
        d3 |= alignbit(d3, d2, 32 - r);
        d2 |= alignbit(d2, d1, 32 - r);
        d1 |= alignbit(d1, d0, 32 - r);
      }
    }
    const uint32_t bad[2] = {d2, d3};
    if ((d2 & d3) != 0xffffffffu) {
      constexpr uint32_t kMask2 = (2 * K - 64 >= 32) ? 0xffffffffu : ((1u << (2 * K - 64)) - 1u);
      constexpr uint32_t kMask3 = (2 * K - 96 >= 32) ? 0xffffffffu : (2 * K - 96 <= 0 ? 0u : ((1u << ((2 * K - 96) & 31)) - 1u));
      const uint32_t fw[10] = {prev.x, prev.y, prev.z, prev.w, cur.x, cur.y, cur.z, cur.w, 0u, 0u};
      uint32_t rw[10];
#pragma unroll
      for (int j = 0; j < 10; ++j) {
        const uint32_t x = __builtin_bitreverse32(fw[9 - j]);
        rw[j] = ~(((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1));
      }
      auto columns = [&](auto base_tag, int i_begin, int i_end) {
        constexpr int kBase = decltype(base_tag)::value;
#pragma unroll 1
        for (int i = i_begin; i < i_end; ++i) {
          constexpr int kOff = 65 - K;
          const uint32_t fo = 2u * (uint32_t)((i + kOff) & 15), ro = 30u - 2u * (uint32_t)i;
          uint32_t A[7], B[7];
#pragma unroll
          for (int j = 0; j < 7; ++j) {
            A[j] = alignbit(fw[kBase + j + 1], fw[kBase + j], fo);  // kBase <= 2: index <= 9
            B[j] = alignbit(rw[j + 3], rw[j + 2], ro);
          }
#pragma unroll
          for (int wi = 0; wi < 4; ++wi) {
            const uint32_t f0 = A[wi], f1 = A[wi + 1], f2 = A[wi + 2] & kMask2, f3 = A[wi + 3] & kMask3;
            const uint32_t r0 = B[3 - wi], r1 = B[4 - wi], r2 = B[5 - wi] & kMask2, r3 = B[6 - wi] & kMask3;
            // forward <= reverse complement as 128-bit numbers: three 64-bit compares into lane masks, combined on the
            // scalar unit, then four selects reading the SGPR pair (see the kernel above for why not VCC)
            const uint64_t lt_hi = __builtin_amdgcn_uicmpl(u64_of(f2, f3), u64_of(r2, r3), 36 /* ICMP_ULT */);
            const uint64_t eq_hi = __builtin_amdgcn_uicmpl(u64_of(f2, f3), u64_of(r2, r3), 32 /* ICMP_EQ */);
            const uint64_t le_lo = __builtin_amdgcn_uicmpl(u64_of(f0, f1), u64_of(r0, r1), 37 /* ICMP_ULE */);
            const uint64_t fwd = lt_hi | (eq_hi & le_lo);
            uint32_t c0, c1, c2, c3;
            asm("s_nop 1\n\tv_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(c0) : "v"(r0), "v"(f0), "s"(fwd));
            asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(c1) : "v"(r1), "v"(f1), "s"(fwd), "v"(c0));
            asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(c2) : "v"(r2), "v"(f2), "s"(fwd), "v"(c1));
            asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(c3) : "v"(r3), "v"(f3), "s"(fwd), "v"(c2));
            const uint32_t cw[4] = {c0, c1, c2, c3};
            uint64_t P[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < kWords; ++j) {
              const uint32_t src = cw[j >> 1];
              const uint32_t glo = (src >> (16 * (j & 1))) & 0xffu, ghi = (src >> (16 * (j & 1) + 8)) & 0xffu;
              const uint64_t lo = s_lo[j][glo];
              P[j] = u64_of((uint32_t)lo, (uint32_t)(lo >> 32) + s_hi[j][ghi]);
            }
            uint64_t U, V;
            murmur3_pre_last_mul<K>(P, U, V);
            if (take_all || last_mul_high_sum_plus1(U, V) <= screen_hi) {
              const uint64_t X = U * kF2, Y = V * kF2;
              const uint64_t h = (X ^ (X >> 33)) + (Y ^ (Y >> 33));
              if (h > max_hash || ((bad[wi >> 1] >> (16 * (wi & 1) + i)) & 1u)) continue;
              const uint32_t slot = atomicAdd(&s_n, 1u);
              if (slot < kStageCap) {
                s_hash[slot] = h;
                s_blk[slot] = t;
              } else if (region_off) {
                region_append(cand_hash, region_off, cursor, overflow, find_genome(genome_blk, n_genomes, t), h);
              } else {
                const unsigned long long g = atomicAdd(count, 1ULL);
                if (g < cap) {
                  cand_hash[g] = h;
                  cand_genome[g] = find_genome(genome_blk, n_genomes, t);
                }
              }
            }
          }
        }
      };
      constexpr int kOff0 = 65 - K, kSplit = (kOff0 & 15) ? 16 - (kOff0 & 15) : 16;
      columns(std::integral_constant<int, (kOff0 >> 4)>{}, 0, kSplit);
      if constexpr (kSplit < 16) columns(std::integral_constant<int, (kOff0 >> 4) + 1>{}, kSplit, 16);
    }
  }
  __syncthreads();
  const uint32_t n = min(s_n, kStageCap);
  if (n == 0) return;
  if (region_off) {
    const uint32_t b0 = blk0 + blockIdx.x * kThreads;
    const uint32_t b1 = min(b0 + (uint32_t)kThreads, n_blocks64) - 1u;
    const uint32_t g0 = find_genome(genome_blk, n_genomes, b0);
    if (genome_blk[g0 + 1] > b1) {
      if (tid == 0) s_base = atomicAdd(&cursor[g0], n);
      __syncthreads();
      const uint64_t room = region_off[g0 + 1] - region_off[g0];
      uint64_t *__restrict__ dst = cand_hash + region_off[g0];
      for (uint32_t i = tid; i < n; i += kThreads) {
        const uint64_t slot = s_base + i;
        if (slot < room) dst[slot] = s_hash[i]; else *overflow = 1u;
      }
    } else {
      for (uint32_t i = tid; i < n; i += kThreads)
        region_append(cand_hash, region_off, cursor, overflow, find_genome(genome_blk, n_genomes, s_blk[i]), s_hash[i]);
    }
    return;
  }
  if (tid == 0) s_base = atomicAdd(count, (unsigned long long)n);
  __syncthreads();
  const unsigned long long base = s_base;
  for (uint32_t i = tid; i < n; i += kThreads) {
    const unsigned long long g = base + i;
    if (g < cap) {
      cand_hash[g] = s_hash[i];
      cand_genome[g] = find_genome(genome_blk, n_genomes, s_blk[i]);
    }
  }
}

// ---- k from 33 to 64, one window per thread-step (PA_KMER_LONG=plain: the cross-check of the kernel above) ---------
// sourmash hashes k-mers of any length (its own defaults are 21, 31 and 51) and the reference passes --kmersize
// through (pyani_plus/public_cli_args.py:229, pyani_plus/methods/sourmash.py:75-76).  Beyond 32 bases a k-mer no
// longer fits the register pair the kernel above is built around, so these sizes take a simpler decomposition of the
// same arithmetic: one thread per window END (the window reaches back up to 63 positions, into blocks that are always
// already there -- also when the arena arrives in chunks), both strands as four 32-bit words, the first multiply
// of each of the up to eight murmur words from the same kind of LDS tables as above, the high-word screen before
// the last multiply.  Same results as the oracle for every k.
__device__ __forceinline__ uint32_t rev_groups32(uint32_t x) {  // the sixteen 2-bit groups of a word in reverse order
  x = __builtin_bitreverse32(x);
  return ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
}

// everything of MurmurHash3_x64_128 between the first multiply of every word (P[j], from the tables) and the last
// multiply of the two fmix64: U and V as in murmur3_pre_last_mul, for a run-time k (uniform over the launch)
__device__ __forceinline__ void murmur3_pre_last_mul_rt(const uint64_t (&P)[8], uint32_t k, uint64_t &U, uint64_t &V) {
  uint64_t h1 = 42, h2 = 42;
  const uint32_t nblocks = k >> 4, tail = k & 15u;
#pragma unroll
  for (uint32_t i = 0; i < 4; ++i) {
    if (i >= nblocks) break;
    h1 ^= rotl64(P[2 * i], 31) * kC2;
    h1 = rotl64(h1, 27) + h2;
    h1 = times5_plus(h1, 0x52dce729ULL);
    h2 ^= rotl64(P[2 * i + 1], 33) * kC1;
    h2 = rotl64(h2, 31) + h1;
    h2 = times5_plus(h2, 0x38495ab5ULL);
  }
  uint64_t t1 = 0, t2 = 0;
#pragma unroll
  for (uint32_t i = 0; i < 4; ++i)
    if (i == nblocks) { t1 = P[2 * i]; t2 = P[2 * i + 1]; }
  if (tail > 8) h2 ^= rotl64(t2, 33) * kC1;
  if (tail > 0) h1 ^= rotl64(t1, 31) * kC2;
  h1 ^= (uint64_t)k;
  h2 ^= (uint64_t)k;
  h1 += h2;
  h2 += h1;
  h1 ^= h1 >> 33; h1 *= kF1; h1 ^= h1 >> 33;
  h2 ^= h2 >> 33; h2 *= kF1; h2 ^= h2 >> 33;
  U = h1;
  V = h2;
}

constexpr int kLongIter = 32;  // window ends per thread of kmer_hash_long_kernel
__global__ __launch_bounds__(kThreads) void kmer_hash_long_kernel(
    const uint32_t *__restrict__ packed, const uint32_t *__restrict__ mask, uint64_t pos0, uint64_t pos1, uint32_t k,
    const uint32_t *__restrict__ genome_blk, uint32_t n_genomes, uint64_t max_hash, uint64_t *__restrict__ cand_hash,
    uint32_t *__restrict__ cand_genome, uint64_t cap, unsigned long long *__restrict__ count,
    const uint64_t *__restrict__ region_off, uint32_t *__restrict__ cursor, uint32_t *__restrict__ overflow) {
  __shared__ uint64_t s_lo[8][256];
  __shared__ uint32_t s_hi[8][256];
  const uint32_t tid = threadIdx.x;
  const uint32_t n_words = (k + 7u) >> 3;
  for (uint32_t j = 0; j < n_words; ++j) {  // word j of the k-mer: bases 8j .. 8j+7, the last one partial
    const uint64_t cj = (j & 1u) ? kC2 : kC1;
    s_lo[j][tid] = (uint64_t)ascii_group(tid, (int)k - 8 * (int)j) * cj;
    s_hi[j][tid] = (uint32_t)((uint64_t)ascii_group(tid, (int)k - 8 * (int)j - 4) * cj);
  }
  __syncthreads();
  // A workgroup takes kLongIter * 256 consecutive window ends, 256 at a time (the tables above cost as much as hashing
  // a window: they have to serve many).  The grid is two-dimensional: an arena of 1 000 genomes of 5 Mb has 5 * 10^9
  // positions, more than one grid dimension is sure to hold.
  const uint64_t wg_first = pos0 + ((uint64_t)blockIdx.y * gridDim.x + blockIdx.x) * (uint64_t)(kThreads * kLongIter);
#pragma unroll 1
  for (int it = 0; it < kLongIter; ++it) {
  const uint64_t e = wg_first + (uint64_t)it * kThreads + tid;  // last position of the window
  if (e >= pos1 || e + 1 < k) continue;
  const uint64_t a = e + 1 - k;  // first position
  // ---- usable?  the k mask bits from position a on: up to three mask words
  {
    const uint64_t m0 = a >> 5, m_last = e >> 5;
    const uint32_t sm = (uint32_t)(a & 31u);
    const uint32_t w0 = mask[m0], w1 = m0 + 1 <= m_last ? mask[m0 + 1] : 0u, w2 = m0 + 2 <= m_last ? mask[m0 + 2] : 0u;
    const uint32_t b0 = alignbit(w1, w0, sm), b1 = alignbit(w2, w1, sm);  // a shift of 0 returns the low operand
    const uint32_t top = k == 64 ? 0xffffffffu : ((1u << (k - 32u)) - 1u);  // k is 33 .. 64 here
    if (b0 | (b1 & top)) continue;  // an invalid position inside the window (also: a record or genome boundary)
  }
  // ---- the window as four words, base j at bits 2j (the order murmur wants), and its reverse complement
  uint32_t F[4], R[4];
  {
    const uint64_t p0 = a >> 4, p_last = e >> 4;
    const uint32_t sp = 2u * (uint32_t)(a & 15u);
    uint32_t w[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) w[i] = p0 + i <= p_last ? packed[p0 + i] : 0u;
#pragma unroll
    for (int i = 0; i < 4; ++i) F[i] = alignbit(w[i + 1], w[i], sp);
    const uint32_t used = 2u * k - 64u;  // bits of the upper two words that belong to the k-mer (2 .. 64)
    if (used < 32u) { F[2] &= (1u << used) - 1u; F[3] = 0u; }
    else if (used < 64u) F[3] &= (1u << (used - 32u)) - 1u;
    // complement of the group-reversed window, moved down by the 128 - 2k bits the reversal left empty at the bottom
    const uint32_t C[6] = {~rev_groups32(F[3]), ~rev_groups32(F[2]), ~rev_groups32(F[1]), ~rev_groups32(F[0]), 0u, 0u};
    const uint32_t s = 128u - 2u * k, ws = s >> 5, bs = s & 31u;  // 0 .. 62: at most one whole word
#pragma unroll
    for (int i = 0; i < 4; ++i) R[i] = ws ? alignbit(C[i + 2], C[i + 1], bs) : alignbit(C[i + 1], C[i], bs);
    if (used < 32u) { R[2] &= (1u << used) - 1u; R[3] = 0u; }
    else if (used < 64u) R[3] &= (1u << (used - 32u)) - 1u;
  }
  // ---- the lexicographically smaller strand: "forward <= reverse complement" is F <= R as 128-bit numbers
  // (MSB-first order of one strand is the complement of the LSB-first form of the other, see the kernel above)
  const uint64_t f_hi = u64_of(F[2], F[3]), f_lo = u64_of(F[0], F[1]), r_hi = u64_of(R[2], R[3]), r_lo = u64_of(R[0], R[1]);
  const bool fwd = f_hi < r_hi || (f_hi == r_hi && f_lo <= r_lo);
  uint32_t c[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) c[i] = fwd ? F[i] : R[i];
  uint64_t P[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (uint32_t j = 0; j < 8; ++j) {
    if (j >= n_words) break;
    const uint32_t src = c[j >> 1], glo = (src >> (16u * (j & 1u))) & 0xffu, ghi = (src >> (16u * (j & 1u) + 8u)) & 0xffu;
    const uint64_t lo = s_lo[j][glo];
    P[j] = u64_of((uint32_t)lo, (uint32_t)(lo >> 32) + s_hi[j][ghi]);
  }
  uint64_t U, V;
  murmur3_pre_last_mul_rt(P, k, U, V);
  const uint32_t max_hi = (uint32_t)(max_hash >> 32);
  if (max_hi != 0xffffffffu && last_mul_high_sum_plus1(U, V) > max_hi + 1u) continue;  // the screen of the kernel above
  const uint64_t X = U * kF2, Y = V * kF2;
  const uint64_t h = (X ^ (X >> 33)) + (Y ^ (Y >> 33));
  if (h > max_hash) continue;
  const uint32_t blk = (uint32_t)(e >> 6);
  const uint32_t g = find_genome(genome_blk, n_genomes, blk);
  if (region_off) {
    region_append(cand_hash, region_off, cursor, overflow, g, h);
  } else {
    const unsigned long long slot = atomicAdd(count, 1ULL);
    if (slot < cap) {
      cand_hash[slot] = h;
      cand_genome[slot] = g;
    }
  }
  }  // window ends of this workgroup
}

}  // namespace

int pa_launch_kmer_hash(pa_ctx *c, const uint32_t *d_packed, const uint32_t *d_mask, const uint64_t *d_dirty, uint64_t n_blocks64,
                        const uint32_t *d_genome_blk, uint32_t n_genomes, uint32_t k, uint64_t max_hash,
                        uint64_t *d_cand_hash, uint32_t *d_cand_genome, uint64_t cap, uint64_t *d_count,
                        const uint64_t *d_region_off, uint32_t *d_cursor, uint32_t *d_overflow, uint64_t blk0,
                        hipStream_t stream) {
  PA_REQUIRE(n_blocks64 < (1ULL << 32), "arena too large: %llu blocks of 64 bases", (unsigned long long)n_blocks64);
  if (n_blocks64 <= blk0) return PA_OK;
  const char *long_form = PA_TOOL_ENV("PA_KMER_LONG");  // "plain": the one-window-per-step form (cross-check, read per launch)
  const bool long_plain = long_form && long_form[0] == 'p';
  if (k > 32 && k <= 64 && !long_plain) {
    PA_REQUIRE((blk0 & 63u) == 0, "k-mer hash launch must start at a multiple of 64 blocks, not %llu", (unsigned long long)blk0);
    const uint32_t grid = ceil_div_u64(n_blocks64 - blk0, kThreads);
#define PA_WIDE_CASE(KK)                                                                                                   \
  case KK:                                                                                                                 \
    hipLaunchKernelGGL((kmer_hash_wide_kernel<KK>), dim3(grid), dim3(kThreads), 0, stream ? stream : c->stream,            \
                       reinterpret_cast<const uint4 *>(d_packed), reinterpret_cast<const uint2 *>(d_mask), d_dirty,        \
                       (uint32_t)n_blocks64, d_genome_blk, n_genomes, max_hash, d_cand_hash, d_cand_genome, cap,           \
                       reinterpret_cast<unsigned long long *>(d_count), d_region_off, d_cursor, d_overflow, (uint32_t)blk0); \
    break;
    switch (k) {
      PA_WIDE_CASE(33) PA_WIDE_CASE(34) PA_WIDE_CASE(35) PA_WIDE_CASE(36) PA_WIDE_CASE(37) PA_WIDE_CASE(38) PA_WIDE_CASE(39)
      PA_WIDE_CASE(40) PA_WIDE_CASE(41) PA_WIDE_CASE(42) PA_WIDE_CASE(43) PA_WIDE_CASE(44) PA_WIDE_CASE(45) PA_WIDE_CASE(46)
      PA_WIDE_CASE(47) PA_WIDE_CASE(48) PA_WIDE_CASE(49) PA_WIDE_CASE(50) PA_WIDE_CASE(51) PA_WIDE_CASE(52) PA_WIDE_CASE(53)
      PA_WIDE_CASE(54) PA_WIDE_CASE(55) PA_WIDE_CASE(56) PA_WIDE_CASE(57) PA_WIDE_CASE(58) PA_WIDE_CASE(59) PA_WIDE_CASE(60)
      PA_WIDE_CASE(61) PA_WIDE_CASE(62) PA_WIDE_CASE(63) PA_WIDE_CASE(64)
      default: break;
    }
#undef PA_WIDE_CASE
    PA_HIP(hipGetLastError());
    return PA_OK;
  }
  if (k > 32 && k <= 64) {
    const uint64_t pos0 = blk0 * 64, pos1 = n_blocks64 * 64;
    const uint64_t per_wg = (uint64_t)kThreads * kLongIter;
    const uint64_t n_wg = (pos1 - pos0 + per_wg - 1) / per_wg;
    const uint32_t gx = (uint32_t)std::min<uint64_t>(n_wg, 1u << 20), gy = (uint32_t)((n_wg + gx - 1) / gx);
    PA_REQUIRE(gy <= 65535u, "arena too large for one launch of the long k-mer kernel: %llu positions", (unsigned long long)(pos1 - pos0));
    hipLaunchKernelGGL(kmer_hash_long_kernel, dim3(gx, gy), dim3(kThreads), 0,
                       stream ? stream : c->stream, d_packed, d_mask, pos0, pos1, k, d_genome_blk, n_genomes, max_hash,
                       d_cand_hash, d_cand_genome, cap, reinterpret_cast<unsigned long long *>(d_count), d_region_off,
                       d_cursor, d_overflow);
    PA_HIP(hipGetLastError());
    return PA_OK;
  }
#define PA_K_CASE(KK) \
  case KK: return launch<KK>(c, d_packed, d_mask, d_dirty, n_blocks64, d_genome_blk, n_genomes, max_hash, d_cand_hash, d_cand_genome, cap, d_count, d_region_off, d_cursor, d_overflow, blk0, stream);
  PA_REQUIRE((blk0 & 63u) == 0, "k-mer hash launch must start at a multiple of 64 blocks, not %llu", (unsigned long long)blk0);
  switch (k) {
    PA_K_CASE(1) PA_K_CASE(2) PA_K_CASE(3) PA_K_CASE(4) PA_K_CASE(5) PA_K_CASE(6) PA_K_CASE(7) PA_K_CASE(8)
    PA_K_CASE(9) PA_K_CASE(10) PA_K_CASE(11) PA_K_CASE(12) PA_K_CASE(13) PA_K_CASE(14) PA_K_CASE(15) PA_K_CASE(16)
    PA_K_CASE(17) PA_K_CASE(18) PA_K_CASE(19) PA_K_CASE(20) PA_K_CASE(21) PA_K_CASE(22) PA_K_CASE(23) PA_K_CASE(24)
    PA_K_CASE(25) PA_K_CASE(26) PA_K_CASE(27) PA_K_CASE(28) PA_K_CASE(29) PA_K_CASE(30) PA_K_CASE(31) PA_K_CASE(32)
    default:
      pa_set_error("k=%u outside [1,64]", k);
      return PA_E_INVALID;
  }
#undef PA_K_CASE
}
