// pairs_bitrow.hip -- all-vs-all sketch intersection sizes (gfx950), default algorithm.
//
// Replaces `sourmash sig collect` x2 + `sourmash scripts manysearch`
// (pyani_plus/methods/sourmash.py:162-200), whose `intersect_hashes` column is
// |Q n S| from a pairwise merge of two sorted u64 lists.  Instead of N^2
// pairwise merges this path sorts every posting once:
//
//   1. dictionary: radix-sort all P = sum |S_g| hashes (value = posting index);
//      equal hashes become adjacent, head flags + scan give each distinct hash
//      a dense id in [0, U).  Exact: ids are a bijection of the hash values.
//   2. bit rows:   row[id] = bitset over the subjects of the tile that contain
//      hash `id` (U rows of W 32-bit words, built with atomicOr).
//   3. counts:     for query q, counts[q][:] = column sums of the rows selected
//      by q's ids, accumulated 128 columns per lane in bit-sliced (vertical)
//      counters -- one 16-byte row fragment adds into 128 pair counters with
//      16 VALU ops.
//
// The per-ordered-pair cost drops from O(|Q|+|S|) merge steps to
// O(|Q|/32) word operations; the result is the exact integer |Q n S|.
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "pa_internal.h"

namespace {

constexpr int kThreads = 256;
constexpr int kPlanes = 10;                      // vertical counter planes -> flush every 1023 rows (a thread of a 16-thread row sees 312 of a 5 000-hash sketch: one flush)
constexpr uint32_t kMaxTileSubjects = 2048;      // widest bit row: 64 words = 256 bytes
constexpr uint32_t kNone = 0xffffffffu;          // "no id": the hash occurs in no subject of the tile
constexpr uint64_t kEmptyKey = ~0ULL;            // empty slot of the hash dictionary

__global__ __launch_bounds__(kThreads) void iota_kernel(uint32_t *__restrict__ vals, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i < n) vals[i] = (uint32_t)i;
}

// OR of every sketch's largest hash (its last entry): the highest set bit of that OR is the
// highest set bit of any key, which bounds the radix passes.  One thread per genome.
__global__ __launch_bounds__(kThreads) void last_or_kernel(const uint64_t *__restrict__ hashes,
                                                           const uint64_t *__restrict__ off, uint32_t n,
                                                           unsigned long long *__restrict__ or_out) {
  const uint32_t g = blockIdx.x * kThreads + threadIdx.x;
  uint64_t k = 0;
  if (g < n && off[g + 1] > off[g]) k = hashes[off[g + 1] - 1];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) k |= __shfl_xor(k, o, 64);
  if ((threadIdx.x & 63u) == 0 && k) atomicOr(or_out, (unsigned long long)k);
}

__global__ __launch_bounds__(kThreads) void key_heads_kernel(const uint64_t *__restrict__ keys, uint64_t n,
                                                             uint32_t *__restrict__ flags) {
  const uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i < n) flags[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}

// largest g with off[g] <= j
__device__ __forceinline__ uint32_t owner_of(const uint64_t *__restrict__ off, uint32_t n, uint64_t j) {
  uint32_t lo = 0, hi = n;
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (off[mid] <= j) lo = mid; else hi = mid;
  }
  return lo;
}

// sorted position i -> id; scatter it back to CSR order and remember (id, genome) in sorted order
__global__ __launch_bounds__(kThreads) void assign_ids_kernel(
    const uint32_t *__restrict__ sorted_post, const uint32_t *__restrict__ flags, const uint32_t *__restrict__ pos,
    uint64_t n, const uint64_t *__restrict__ off, uint32_t n_genomes, uint32_t *__restrict__ ids_csr,
    uint32_t *__restrict__ id_sorted, uint32_t *__restrict__ genome_sorted) {
  const uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  const uint32_t id = pos[i] + flags[i] - 1u;  // inclusive scan - 1
  const uint32_t j = sorted_post[i];
  ids_csr[j] = id;
  id_sorted[i] = id;
  genome_sorted[i] = owner_of(off, n_genomes, j);
}

__global__ __launch_bounds__(kThreads) void build_rows_kernel(const uint32_t *__restrict__ id_sorted,
                                                              const uint32_t *__restrict__ genome_sorted, uint64_t n,
                                                              uint32_t t0, uint32_t t1, uint32_t w32,
                                                              uint32_t *__restrict__ rows) {
  const uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  const uint32_t g = genome_sorted[i];
  if (g < t0 || g >= t1) return;
  const uint32_t col = g - t0;
  atomicOr(&rows[(uint64_t)id_sorted[i] * w32 + (col >> 5)], 1u << (col & 31u));
}

struct U4 { uint32_t v[4]; };

// One workgroup per query.  TPR = threads per bit row (row = TPR * 16 bytes).
template <int TPR>
__global__ __launch_bounds__(kThreads) void row_sum_kernel(const uint32_t *__restrict__ ids_csr,
                                                           const uint64_t *__restrict__ off, uint32_t q0,
                                                           const uint32_t *__restrict__ rows, uint32_t tile_cols,
                                                           uint32_t *__restrict__ counts, uint32_t ns,
                                                           uint32_t col0) {
  constexpr int kRowsPerIter = kThreads / TPR;
  constexpr int kW32 = TPR * 4;
  __shared__ uint32_t s_cnt[kW32 * 32];
  const uint32_t tid = threadIdx.x;
  for (uint32_t x = tid; x < kW32 * 32; x += kThreads) s_cnt[x] = 0;
  __syncthreads();

  const uint32_t q = q0 + blockIdx.x;
  const uint64_t beg = off[q], len = off[q + 1] - beg;
  const uint32_t *__restrict__ ids = ids_csr + beg;
  const uint32_t slot = tid / TPR, quad = tid % TPR;

  uint32_t plane[kPlanes][4];
#pragma unroll
  for (int p = 0; p < kPlanes; ++p)
#pragma unroll
    for (int c = 0; c < 4; ++c) plane[p][c] = 0;
  uint32_t pending = 0;

  // The vertical counters into the LDS counts: only the columns whose counter is not zero are visited (a hash of one
  // species sets ~50 of a tile's 2 048 columns; walking all 128 columns of a thread was more work than the row sums)
  auto flush = [&]() {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      uint32_t any = 0;
#pragma unroll
      for (int p = 0; p < kPlanes; ++p) any |= plane[p][c];
      while (any) {
        const uint32_t b = (uint32_t)__builtin_ctz(any);
        any &= any - 1u;
        uint32_t v = 0;
#pragma unroll
        for (int p = 0; p < kPlanes; ++p) v |= ((plane[p][c] >> b) & 1u) << p;
        atomicAdd(&s_cnt[(quad * 4 + c) * 32 + b], v);
      }
    }
#pragma unroll
    for (int p = 0; p < kPlanes; ++p)
#pragma unroll
      for (int c = 0; c < 4; ++c) plane[p][c] = 0;
    pending = 0;
  };

  // TPR need not divide the workgroup: the threads past the last whole row sit the loop out.  Eight rows per turn:
  // their loads are in flight together, and they enter the vertical counters through a tree of carry-save adders
  // (sum = a ^ b ^ c, carry = majority(a, b, c): two three-input bit operations on gfx950) -- ones, twos and fours
  // are kept across turns, the eights ripple into the seven upper planes -- about 3 operations per row and word
  // instead of the 30 of adding every row to all ten planes.
  constexpr int kBatch = 8;
  auto csa = [](uint32_t &carry, uint32_t &sum, uint32_t a, uint32_t b, uint32_t c) {
    const uint32_t u = a ^ b;
    carry = (a & b) | (u & c);
    sum = u ^ c;
  };
  // the ids of a turn are loaded one turn ahead: a row's address depends on its id, and the two loads in a row were
  // what a turn waited for
  const uint64_t j_begin = slot < (uint32_t)kRowsPerIter ? slot : len;
  uint32_t id_next[kBatch];
#pragma unroll
  for (int u = 0; u < kBatch; ++u) {
    const uint64_t j = j_begin + (uint64_t)u * kRowsPerIter;
    id_next[u] = j < len ? ids[j] : kNone;
  }
  for (uint64_t j0 = j_begin; j0 < len; j0 += (uint64_t)kRowsPerIter * kBatch) {
    uint4 r[kBatch];
#pragma unroll
    for (int u = 0; u < kBatch; ++u) {
      const uint32_t id = id_next[u];  // kNone: hash absent from every subject of the tile
      r[u] = id != kNone ? *reinterpret_cast<const uint4 *>(rows + (uint64_t)id * kW32 + quad * 4) : make_uint4(0u, 0u, 0u, 0u);
    }
#pragma unroll
    for (int u = 0; u < kBatch; ++u) {
      const uint64_t j = j0 + (uint64_t)(kBatch + u) * kRowsPerIter;
      id_next[u] = j < len ? ids[j] : kNone;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      auto word = [&](int u) -> uint32_t { return c == 0 ? r[u].x : c == 1 ? r[u].y : c == 2 ? r[u].z : r[u].w; };
      uint32_t twos_a, twos_b, fours_a, fours_b, eights;
      csa(twos_a, plane[0][c], plane[0][c], word(0), word(1));
      csa(twos_b, plane[0][c], plane[0][c], word(2), word(3));
      csa(fours_a, plane[1][c], plane[1][c], twos_a, twos_b);
      csa(twos_a, plane[0][c], plane[0][c], word(4), word(5));
      csa(twos_b, plane[0][c], plane[0][c], word(6), word(7));
      csa(fours_b, plane[1][c], plane[1][c], twos_a, twos_b);
      csa(eights, plane[2][c], plane[2][c], fours_a, fours_b);
#pragma unroll
      for (int p = 3; p < kPlanes; ++p) {
        const uint32_t t = plane[p][c] & eights;
        plane[p][c] ^= eights;
        eights = t;
      }
    }
    pending += kBatch;
    if (pending + kBatch > (1u << kPlanes) - 1u) flush();
  }
  if (pending) flush();
  __syncthreads();
  uint32_t *__restrict__ out = counts + (uint64_t)blockIdx.x * ns + col0;
  for (uint32_t x = tid; x < tile_cols; x += kThreads) out[x] = s_cnt[x];
}

// Zero the first U rows of the bit-row table, U read on the device (the number of distinct subject hashes the
// insert kernel has just counted): the grid is sized for the upper bound (one row per subject posting) and the
// blocks past U * w32 words leave at once, so the host never waits for U.
__global__ __launch_bounds__(kThreads) void zero_rows_kernel(uint4 *__restrict__ rows, const uint32_t *__restrict__ d_u,
                                                             uint32_t w32) {
  const uint64_t n16 = ((uint64_t)(*d_u ? *d_u : 1u) * w32) / 4u;  // 16-byte pieces (w32 is a multiple of 4)
  const uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i < n16) rows[i] = make_uint4(0u, 0u, 0u, 0u);
}

// ---- hash dictionary (PA_PAIRS_BITROW_HASH): dense ids without sorting -------------------------
// Open addressing, linear probing, load 2/3.  Only the hashes of the tile's SUBJECTS are inserted; the
// winner of a slot draws the dense id.  Key and id share one 16-byte entry, so a probe costs one HBM
// line, which is what bounds both kernels (the table is far larger than L2).  Lookups run in a later
// launch, so every id is visible.  The legal key ~0 doubles as the empty marker and is therefore kept
// in `special[0]` instead.
struct DictEntry {
  unsigned long long key;
  uint32_t id;
  uint32_t pad;
};
static_assert(sizeof(DictEntry) == 16, "one probe = one 16-byte entry");

__device__ __forceinline__ uint32_t slot_of(uint64_t h, uint32_t cap) {
  const uint32_t x = ((uint32_t)h * 0x9E3779B1u) ^ (uint32_t)(h >> 32);
  return (uint32_t)(((uint64_t)x * cap) >> 32);
}

constexpr int kInsertPerThread = 4;
__global__ __launch_bounds__(kThreads) void table_insert_kernel(const uint64_t *__restrict__ hashes, uint64_t p0,
                                                                uint64_t p1, DictEntry *__restrict__ table,
                                                                uint32_t cap, uint32_t *__restrict__ counter,
                                                                uint32_t *__restrict__ special) {
  __shared__ uint32_t s_won, s_base;
  if (threadIdx.x == 0) s_won = 0;
  __syncthreads();
  uint32_t won_slot[kInsertPerThread];
  int won = 0;
  bool won_special = false;
  const uint64_t base = p0 + (uint64_t)blockIdx.x * (kThreads * kInsertPerThread) + threadIdx.x;
#pragma unroll
  for (int i = 0; i < kInsertPerThread; ++i) {
    const uint64_t p = base + (uint64_t)i * kThreads;
    if (p >= p1) break;
    const uint64_t h = hashes[p];
    if (h == kEmptyKey) {
      won_special = atomicCAS(&special[0], kNone, kNone - 1u) == kNone;
      continue;
    }
    uint32_t slot = slot_of(h, cap);
    for (;;) {
      // most postings repeat a key that is already there: look before paying for the atomic.  A plain,
      // cached load (a non-temporal one was 1.5x slower): the popular keys are then served by L2, and a stale
      // line can only show "empty" for a slot that has been taken since, which the CAS below corrects.
      unsigned long long k = *reinterpret_cast<volatile const unsigned long long *>(&table[slot].key);
      if (k == kEmptyKey) k = atomicCAS(&table[slot].key, (unsigned long long)kEmptyKey, (unsigned long long)h);
      if (k == kEmptyKey) { won_slot[won++] = slot; break; }
      if (k == h) break;
      if (++slot == cap) slot = 0;
    }
  }
  // dense ids: one global atomic per workgroup instead of one per new key
  const uint32_t mine = (uint32_t)won + (won_special ? 1u : 0u);
  uint32_t local = mine ? atomicAdd(&s_won, mine) : 0u;
  __syncthreads();
  if (threadIdx.x == 0) s_base = s_won ? atomicAdd(counter, s_won) : 0u;
  __syncthreads();
  local += s_base;
  for (int i = 0; i < won; ++i) table[won_slot[i]].id = local++;
  if (won_special) special[0] = local;
}

// ids of the postings [p0, p1); postings of tile subjects (SET_BITS) also set their bit in the row.  With
// SET_BITS the grid is two-dimensional -- blockIdx.y is the subject, blockIdx.x a 256-posting piece of its sketch
// -- so the column comes from the block index and not from a search in the offsets.
template <bool SET_BITS>
__global__ __launch_bounds__(kThreads) void table_lookup_kernel(
    const uint64_t *__restrict__ hashes, uint64_t p0, uint64_t p1, const DictEntry *__restrict__ table, uint32_t cap,
    const uint32_t *__restrict__ special, uint32_t *__restrict__ ids_csr, const uint64_t *__restrict__ off, uint32_t n,
    uint32_t t0, uint32_t w32, uint32_t *__restrict__ rows) {
  uint64_t p;
  if constexpr (SET_BITS) {
    const uint64_t beg = off[t0 + blockIdx.y];
    p = beg + (uint64_t)blockIdx.x * kThreads + threadIdx.x;
    if (p >= off[t0 + blockIdx.y + 1]) return;
  } else {
    p = p0 + (uint64_t)blockIdx.x * kThreads + threadIdx.x;
    if (p >= p1) return;
  }
  const uint64_t h = hashes[p];
  uint32_t id = kNone;
  if (h == kEmptyKey) {
    id = special[0];
  } else {
    uint32_t slot = slot_of(h, cap);
    for (;;) {
      const uint4 e = *reinterpret_cast<const uint4 *>(&table[slot]);  // key and id in one load
      const uint64_t k = ((uint64_t)e.y << 32) | e.x;
      if (k == h) { id = e.z; break; }
      if (k == kEmptyKey) break;
      if (++slot == cap) slot = 0;
    }
  }
  ids_csr[p] = id;
  if constexpr (SET_BITS) {
    const uint32_t col = blockIdx.y;
    atomicOr(&rows[(uint64_t)id * w32 + (col >> 5)], 1u << (col & 31u));
  }
}

// Order-free fingerprint of a run of postings: (sum, sum of squares-ish mix) over the hashes, two 64-bit words.
// A prepared dictionary is only valid for the subject tile whose postings it was built from; the two runs live in
// different buffers (the rank's own sketches before the all-gather, the gathered CSR after), so they are compared
// by content.
__global__ __launch_bounds__(kThreads) void postings_fingerprint_kernel(const uint64_t *__restrict__ hashes, uint64_t n,
                                                                        unsigned long long *__restrict__ out) {
  uint64_t a = 0, b = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kThreads) {
    const uint64_t h = hashes[i];
    a += h;
    b += (h ^ (h >> 29)) * 0x9E3779B97F4A7C15ULL;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
  if ((threadIdx.x & 63u) == 0) {
    atomicAdd(&out[0], (unsigned long long)a);
    atomicAdd(&out[1], (unsigned long long)b);
  }
}

// All-vs-all over more than one subject tile: |A n B| = |B n A|, so tile j only evaluates the queries of tiles
// i <= j and the blocks below the tile diagonal are the transposes of the ones above.  32 x 32 pieces through LDS
// so that both the read and the write are coalesced; tile edges are multiples of 32, so a piece never straddles one.
__global__ __launch_bounds__(256) void mirror_lower_kernel(uint32_t *__restrict__ counts, uint32_t n, uint32_t tile) {
  __shared__ uint32_t s_t[32][33];
  const uint32_t r0 = blockIdx.y * 32u, c0 = blockIdx.x * 32u;  // destination piece: rows r0.., columns c0..
  if (r0 / tile <= c0 / tile) return;                             // on or above the tile diagonal: computed directly
  const uint32_t tx = threadIdx.x & 31u, ty = threadIdx.x >> 5;   // 32 x 8 threads
  for (uint32_t y = ty; y < 32u; y += 8u) {
    const uint32_t sr = c0 + y, sc = r0 + tx;  // source element (sr, sc) = transpose position
    s_t[y][tx] = (sr < n && sc < n) ? counts[(uint64_t)sr * n + sc] : 0u;
  }
  __syncthreads();
  for (uint32_t y = ty; y < 32u; y += 8u) {
    const uint32_t dr = r0 + y, dc = c0 + tx;
    if (dr < n && dc < n) counts[(uint64_t)dr * n + dc] = s_t[tx][y];
  }
}

template <int TPR>
void launch_row_sum(pa_ctx *c, uint32_t nq, const uint32_t *ids, const uint64_t *off, uint32_t q0,
                    const uint32_t *rows, uint32_t tile_cols, uint32_t *counts, uint32_t ns, uint32_t col0) {
  hipLaunchKernelGGL(row_sum_kernel<TPR>, dim3(nq), dim3(kThreads), 0, c->stream, ids, off, q0, rows, tile_cols,
                     counts, ns, col0);
}

int dispatch_row_sum(pa_ctx *c, int tpr, uint32_t nq, const uint32_t *ids, const uint64_t *off, uint32_t q0,
                     const uint32_t *rows, uint32_t tile_cols, uint32_t *counts, uint32_t ns, uint32_t col0) {
#define PA_ROW_CASE(T) \
  case T: launch_row_sum<T>(c, nq, ids, off, q0, rows, tile_cols, counts, ns, col0); return PA_OK;
  switch (tpr) {
    PA_ROW_CASE(1) PA_ROW_CASE(2) PA_ROW_CASE(3) PA_ROW_CASE(4) PA_ROW_CASE(5) PA_ROW_CASE(6) PA_ROW_CASE(7)
    PA_ROW_CASE(8) PA_ROW_CASE(9) PA_ROW_CASE(10) PA_ROW_CASE(11) PA_ROW_CASE(12) PA_ROW_CASE(13) PA_ROW_CASE(14)
    PA_ROW_CASE(15) PA_ROW_CASE(16)
    default:
      pa_set_error("pair phase: %d threads per bit row (tile wider than %u subjects)", tpr, kMaxTileSubjects);
      return PA_E_INVALID;
  }
#undef PA_ROW_CASE
}

}  // namespace

// Dense ids in ascending hash order for all P postings (id order == hash order): sort (hash, posting),
// flag the first posting of every distinct hash, scan.  Leaves ids in CSR order in c->ids and the
// (id, genome) pairs in sorted order in c->post_genome[0..P) / [P..2P).
int pa_dense_ids_sorted(pa_ctx *c, const uint64_t *d_hashes, const uint64_t *d_off, uint32_t n, uint64_t P,
                        uint64_t *n_distinct) {
  ProfScope prof(c, PA_PROF_PAIR_DICT);
  for (int b = 0; b < 2; ++b) {
    PA_TRY(c->dict_keys[b].reserve(P * sizeof(uint64_t)));
    PA_TRY(c->dict_vals[b].reserve(P * sizeof(uint32_t)));
  }
  PA_TRY(c->ids.reserve(P * sizeof(uint32_t)));
  PA_TRY(c->post_genome.reserve(2 * P * sizeof(uint32_t)));
  PA_TRY(c->flags.reserve(2 * P * sizeof(uint32_t)));
  uint64_t *keys[2] = {c->dict_keys[0].as<uint64_t>(), c->dict_keys[1].as<uint64_t>()};
  uint32_t *vals[2] = {c->dict_vals[0].as<uint32_t>(), c->dict_vals[1].as<uint32_t>()};
  uint32_t *d_ids = c->ids.as<uint32_t>();
  uint32_t *d_id_sorted = c->post_genome.as<uint32_t>();
  uint32_t *d_genome_sorted = d_id_sorted + P;
  uint32_t *d_flags = c->flags.as<uint32_t>(), *d_pos = d_flags + P;
  uint64_t *d_scalars = c->counters.as<uint64_t>();  // [2] = OR of keys, [3] = U

  const uint32_t grid = ceil_div_u64(P, kThreads);
  PA_HIP(hipMemcpyAsync(keys[0], d_hashes, P * sizeof(uint64_t), hipMemcpyDeviceToDevice, c->stream));
  PA_HIP(hipMemsetAsync(d_scalars + 2, 0, 2 * sizeof(uint64_t), c->stream));
  hipLaunchKernelGGL(iota_kernel, dim3(grid), dim3(kThreads), 0, c->stream, vals[0], P);
  hipLaunchKernelGGL(last_or_kernel, dim3(ceil_div_u64(n, kThreads)), dim3(kThreads), 0, c->stream, d_hashes, d_off,
                     n, reinterpret_cast<unsigned long long *>(d_scalars + 2));
  PA_HIP(hipMemcpyAsync(c->h_pinned, d_scalars + 2, sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
  PA_HIP(hipStreamSynchronize(c->stream));
  const uint64_t all_or = c->h_pinned[0];
  int bit_hi = all_or ? 64 - __builtin_clzll(all_or) : 0;
  bit_hi = (bit_hi + 7) & ~7;
  int which = 0;
  PA_TRY(pa_radix_sort_pairs(c, keys, vals, P, 0, bit_hi, false, &which));
  hipLaunchKernelGGL(key_heads_kernel, dim3(grid), dim3(kThreads), 0, c->stream, keys[which], P, d_flags);
  PA_TRY(pa_exclusive_scan_u32(c, d_flags, d_pos, P, d_scalars + 3));
  hipLaunchKernelGGL(assign_ids_kernel, dim3(grid), dim3(kThreads), 0, c->stream, vals[which], d_flags, d_pos, P,
                     d_off, n, d_ids, d_id_sorted, d_genome_sorted);
  PA_HIP(hipMemcpyAsync(c->h_pinned, d_scalars + 3, sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
  PA_HIP(hipStreamSynchronize(c->stream));
  *n_distinct = c->h_pinned[0];
  return PA_OK;
}

int pa_pairs_bitrow(pa_ctx *c, const uint64_t *d_hashes, const uint64_t *d_off, uint32_t n, uint64_t total,
                    uint32_t q0, uint32_t q1, uint32_t s0, uint32_t s1, uint32_t *d_counts) {
  const uint32_t nq = q1 - q0, ns = s1 - s0;
  if (nq == 0 || ns == 0) return PA_OK;
  if (total == 0) {
    PA_HIP(hipMemsetAsync(d_counts, 0, (uint64_t)nq * ns * sizeof(uint32_t), c->stream));
    return PA_OK;
  }
  PA_REQUIRE(total < (1ULL << 32), "pair phase: %llu postings exceed the 32-bit index space",
             (unsigned long long)total);
  const uint64_t P = total;
  uint64_t U = 0;
  PA_TRY(pa_dense_ids_sorted(c, d_hashes, d_off, n, P, &U));
  uint32_t *d_ids = c->ids.as<uint32_t>();
  uint32_t *d_id_sorted = c->post_genome.as<uint32_t>();
  uint32_t *d_genome_sorted = d_id_sorted + P;
  // subject tiles of up to kMaxTileSubjects columns
  for (uint32_t t0 = s0; t0 < s1; t0 += kMaxTileSubjects) {
    const uint32_t t1 = (s1 - t0 > kMaxTileSubjects) ? t0 + kMaxTileSubjects : s1;
    const uint32_t cols = t1 - t0;
    const int tpr = (int)((cols + 127u) / 128u);  // threads per row: row width = tpr*128 columns
    const uint32_t w32 = (uint32_t)tpr * 4u;
    const uint64_t row_bytes = U * w32 * sizeof(uint32_t);
    {
      ProfScope prof(c, PA_PROF_PAIR_DICT);
      PA_TRY(c->bitrows.reserve(row_bytes));
      PA_HIP(hipMemsetAsync(c->bitrows.p, 0, row_bytes, c->stream));
      hipLaunchKernelGGL(build_rows_kernel, dim3(ceil_div_u64(P, kThreads)), dim3(kThreads), 0, c->stream,
                         d_id_sorted, d_genome_sorted, P, t0, t1, w32, c->bitrows.as<uint32_t>());
    }
    {
      ProfScope prof(c, PA_PROF_PAIR_COUNT);
      const uint32_t *rows = c->bitrows.as<uint32_t>();
      PA_TRY(dispatch_row_sum(c, tpr, nq, d_ids, d_off, q0, rows, cols, d_counts, ns, t0 - s0));
    }
    PA_HIP(hipGetLastError());
  }
  return PA_OK;
}


// Build the tile's dictionary from a flat run of subject postings (dense ids drawn by the slot winners).
static int dict_insert(pa_ctx *c, const uint64_t *d_postings, uint64_t n_post, uint32_t *cap_out) {
  const uint64_t cap64 = n_post + n_post / 2 + 1024;  // load <= 2/3
  PA_REQUIRE(cap64 < (1ULL << 32), "pair phase: tile with %llu subject postings is too large for the hash dictionary",
             (unsigned long long)n_post);
  PA_TRY(c->dict_scalars.reserve(64));
  uint32_t *d_counter = c->dict_scalars.as<uint32_t>();  // [0] counter, [1] special id
  PA_TRY(c->dict_keys[0].reserve(cap64 * sizeof(DictEntry)));
  PA_HIP(hipMemsetAsync(c->dict_keys[0].p, 0xff, cap64 * sizeof(DictEntry), c->stream));
  PA_HIP(hipMemsetAsync(d_counter, 0, 4, c->stream));
  PA_HIP(hipMemsetAsync(d_counter + 1, 0xff, 4, c->stream));
  if (n_post)
    hipLaunchKernelGGL(table_insert_kernel, dim3(ceil_div_u64(n_post, kThreads * kInsertPerThread)), dim3(kThreads), 0,
                       c->stream, d_postings, (uint64_t)0, n_post, c->dict_keys[0].as<DictEntry>(), (uint32_t)cap64,
                       d_counter, d_counter + 1);
  PA_HIP(hipGetLastError());
  *cap_out = (uint32_t)cap64;
  return PA_OK;
}

// Multi-GPU overlap: a rank's dictionary needs only the hashes of its own subject
// columns, which it has before the sketch all-gather starts.  pa_pair_dict_prepare enqueues the insert on the
// context's stream and remembers it; the next pa_pair_counts(_ex) whose (single) subject tile holds exactly
// n_postings postings takes the prepared dictionary instead of building one.
int pa_pair_dict_prepare_impl(pa_ctx *c, const uint64_t *d_subject_hashes, uint64_t n_postings) {
  ProfScope prof(c, PA_PROF_PAIR_DICT);
  c->dict_prepared = false;
  uint32_t cap = 0;
  PA_TRY(dict_insert(c, d_subject_hashes, n_postings, &cap));
  // what the dictionary was built from, for the call that consumes it
  unsigned long long *d_fp = c->dict_scalars.as<unsigned long long>() + 1;
  PA_HIP(hipMemsetAsync(d_fp, 0, 16, c->stream));
  if (n_postings)
    hipLaunchKernelGGL(postings_fingerprint_kernel, dim3(std::min<uint32_t>(1024u, ceil_div_u64(n_postings, kThreads))),
                       dim3(kThreads), 0, c->stream, d_subject_hashes, n_postings, d_fp);
  PA_HIP(hipGetLastError());
  c->dict_prepared = true;
  c->dict_prepared_postings = n_postings;
  c->dict_prepared_cap = cap;
  return PA_OK;
}

int pa_pairs_bitrow_hash(pa_ctx *c, const uint64_t *d_hashes, const uint64_t *d_off, const uint64_t *h_off_in,
                         uint32_t n, uint64_t total, uint32_t q0, uint32_t q1, uint32_t s0, uint32_t s1,
                         uint32_t *d_counts) {
  const uint32_t nq = q1 - q0, ns = s1 - s0;
  const bool prepared = c->dict_prepared;
  c->dict_prepared = false;  // a prepared dictionary is used by the next call or not at all
  if (nq == 0 || ns == 0) return PA_OK;
  if (total == 0) {
    PA_HIP(hipMemsetAsync(d_counts, 0, (uint64_t)nq * ns * sizeof(uint32_t), c->stream));
    return PA_OK;
  }
  PA_REQUIRE(total < (1ULL << 32), "pair phase: %llu postings exceed the 32-bit index space", (unsigned long long)total);
  // CSR offsets on the host: handed in by a caller that has them (sketch sizes known from the all-gather or a
  // .sig cache), otherwise one copy of the whole array -- the only host round trip of the pair phase.
  std::vector<uint64_t> h_off_own;
  const uint64_t *h_off = h_off_in;
  if (!h_off) {
    h_off_own.resize(n + 1);
    PA_HIP(hipMemcpyAsync(h_off_own.data(), d_off, (uint64_t)(n + 1) * 8, hipMemcpyDeviceToHost, c->stream));
    PA_HIP(hipStreamSynchronize(c->stream));
    h_off = h_off_own.data();
  }
  PA_TRY(c->ids.reserve(total * sizeof(uint32_t)));
  PA_TRY(c->dict_scalars.reserve(64));
  uint32_t *d_ids = c->ids.as<uint32_t>();
  uint32_t *d_counter = c->dict_scalars.as<uint32_t>();  // [0] counter, [1] special id
  PA_REQUIRE(!prepared || ns <= kMaxTileSubjects, "pair phase: a prepared dictionary serves one tile of at most %u subjects",
             kMaxTileSubjects);
  // all-vs-all over several subject tiles: tile j takes the queries of tiles i <= j only, the rest is mirrored
  static const bool symmetry_off = [] { const char *v = PA_TOOL_ENV("PA_PAIRS_SYMMETRIC"); return v && v[0] == '0'; }();
  const bool symmetric = !symmetry_off && q0 == s0 && q1 == s1 && ns > kMaxTileSubjects;
  for (uint32_t t0 = s0; t0 < s1; t0 += kMaxTileSubjects) {
    const uint32_t t1 = (s1 - t0 > kMaxTileSubjects) ? t0 + kMaxTileSubjects : s1;
    const uint32_t cols = t1 - t0;
    const int tpr = (int)((cols + 127u) / 128u);  // threads per row: row width = tpr*128 columns
    const uint32_t w32 = (uint32_t)tpr * 4u;
    const uint32_t tq1 = symmetric ? t1 : q1;  // last query (exclusive) this tile evaluates
    const uint64_t pt0 = h_off[t0], pt1 = h_off[t1], pq0 = h_off[q0], pq1 = h_off[tq1];
    uint32_t cap = 0;
    {
      ProfScope prof(c, PA_PROF_PAIR_DICT);
      if (prepared) {
        PA_REQUIRE(c->dict_prepared_postings == pt1 - pt0,
                   "pair phase: the prepared dictionary holds %llu postings, the subject tile has %llu",
                   (unsigned long long)c->dict_prepared_postings, (unsigned long long)(pt1 - pt0));
        // same number of postings is not the same postings: compare the fingerprints (one small copy and a wait;
        // the lookups below are then enqueued a few microseconds later than they could have been)
        unsigned long long *d_fp = c->dict_scalars.as<unsigned long long>() + 1;
        PA_HIP(hipMemsetAsync(d_fp + 2, 0, 16, c->stream));
        if (pt1 > pt0)
          hipLaunchKernelGGL(postings_fingerprint_kernel, dim3(std::min<uint32_t>(1024u, ceil_div_u64(pt1 - pt0, kThreads))),
                             dim3(kThreads), 0, c->stream, d_hashes + pt0, pt1 - pt0, d_fp + 2);
        PA_HIP(hipMemcpyAsync(c->h_pinned, d_fp, 32, hipMemcpyDeviceToHost, c->stream));
        PA_HIP(hipStreamSynchronize(c->stream));
        PA_REQUIRE(c->h_pinned[0] == c->h_pinned[2] && c->h_pinned[1] == c->h_pinned[3],
                   "pair phase: the prepared dictionary was built from other postings than the subject tile's "
                   "(same count, %llu, different content)", (unsigned long long)(pt1 - pt0));
        cap = c->dict_prepared_cap;
      } else {
        PA_TRY(dict_insert(c, d_hashes + pt0, pt1 - pt0, &cap));
      }
      // rows: one per distinct subject hash, at most one per subject posting; zeroed up to the device-side count.
      // The speculative bound (no wait for the count) is only taken while it stays moderate: beyond kRowBoundBytes
      // the count comes to the host and the table is sized by it, as unrelated genomes share few hashes but related
      // ones (or a small `scaled`) would otherwise reserve tens of GB that never shrink.
      constexpr uint64_t kRowBoundBytes = 4ULL << 30;
      uint64_t row_bound = (pt1 - pt0) ? (pt1 - pt0) : 1;
      if (row_bound * w32 * sizeof(uint32_t) > kRowBoundBytes && row_bound * w32 * sizeof(uint32_t) > c->bitrows.bytes) {
        PA_HIP(hipMemcpyAsync(c->h_pinned, d_counter, 8, hipMemcpyDeviceToHost, c->stream));
        PA_HIP(hipStreamSynchronize(c->stream));
        const uint64_t distinct = (uint32_t)c->h_pinned[0];  // ids drawn so far, the one of the key ~0 included
        row_bound = std::max<uint64_t>(1, std::min<uint64_t>(row_bound, distinct));
      }
      PA_TRY(c->bitrows.reserve(row_bound * w32 * sizeof(uint32_t)));
      hipLaunchKernelGGL(zero_rows_kernel, dim3(ceil_div_u64(row_bound * w32 / 4u, kThreads)), dim3(kThreads), 0,
                         c->stream, c->bitrows.as<uint4>(), d_counter, w32);
      const DictEntry *table = c->dict_keys[0].as<DictEntry>();
      if (pt1 > pt0) {
        uint64_t longest = 0;
        for (uint32_t g = t0; g < t1; ++g) longest = std::max(longest, h_off[g + 1] - h_off[g]);
        hipLaunchKernelGGL(table_lookup_kernel<true>, dim3(ceil_div_u64(longest, kThreads), cols), dim3(kThreads), 0,
                           c->stream, d_hashes, pt0, pt1, table, cap, d_counter + 1, d_ids, d_off, n, t0, w32,
                           c->bitrows.as<uint32_t>());
      }
      // query postings outside the tile's own range
      const uint64_t a0 = pq0, a1 = pq1 < pt0 ? pq1 : pt0;  // part before the tile
      const uint64_t b0 = pq0 > pt1 ? pq0 : pt1, b1 = pq1;  // part after the tile
      if (a1 > a0)
        hipLaunchKernelGGL(table_lookup_kernel<false>, dim3(ceil_div_u64(a1 - a0, kThreads)), dim3(kThreads), 0,
                           c->stream, d_hashes, a0, a1, table, cap, d_counter + 1, d_ids, d_off, n, t0, w32,
                           (uint32_t *)nullptr);
      if (b1 > b0)
        hipLaunchKernelGGL(table_lookup_kernel<false>, dim3(ceil_div_u64(b1 - b0, kThreads)), dim3(kThreads), 0,
                           c->stream, d_hashes, b0, b1, table, cap, d_counter + 1, d_ids, d_off, n, t0, w32,
                           (uint32_t *)nullptr);
    }
    {
      ProfScope prof(c, PA_PROF_PAIR_COUNT);
      const uint32_t *rows = c->bitrows.as<uint32_t>();
      PA_TRY(dispatch_row_sum(c, tpr, tq1 - q0, d_ids, d_off, q0, rows, cols, d_counts, ns, t0 - s0));
    }
    PA_HIP(hipGetLastError());
  }
  if (symmetric) {
    ProfScope prof(c, PA_PROF_PAIR_COUNT);
    static_assert(kMaxTileSubjects % 32u == 0, "mirror pieces must not straddle a tile edge");
    const uint32_t g = (ns + 31u) / 32u;
    hipLaunchKernelGGL(mirror_lower_kernel, dim3(g, g), dim3(256), 0, c->stream, d_counts, ns, kMaxTileSubjects);
    PA_HIP(hipGetLastError());
  }
  return PA_OK;
}
