/*
 * sourmash_oracle.c -- CPU restatement of the pyani-plus "sourmash" hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under pyani_plus_amd/ may import, link or
 * call this file; it exists so that tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py can check (and time) the HIP path against an
 * independent scalar implementation.
 *
 * Where the algorithm comes from
 * ------------------------------
 * The arithmetic of this path is NOT in /root/reference: pyani-plus shells out
 * to third-party tools, pinned in requirements-thirdparty-linux.txt:8-9 as
 * sourmash-minimal>=4.8.11 and sourmash_plugin_branchwater>=0.9.11:
 *   - sketch:   pyani_plus/methods/sourmash.py:67-83
 *               (`sourmash scripts singlesketch -I DNA -p k=K,scaled=S`)
 *   - pairs:    pyani_plus/methods/sourmash.py:184-200
 *               (`sourmash scripts manysearch -m DNA -t 0`)
 *   - mapping:  pyani_plus/methods/sourmash.py:107-144 and
 *               pyani_plus/private_cli.py:1875-1887
 *               (identity = max_containment_ani, cov_query = query_containment_ani,
 *                pairs missing from the CSV -> NULL)
 * This file restates the published algorithm of those tools (FracMinHash over
 * canonical k-mers hashed with MurmurHash3_x64_128, seed 42, first 64-bit word;
 * containment ANI = containment^(1/k)) and is PINNED against the reference's own
 * fixtures: all 9 `.sig` files, all 27 manysearch.csv rows and the two constants
 * in tests/test_coverage.py:169-174 are reproduced bit for bit by
 * tests/test_oracle_golden.py (fixtures copied as data under tests/golden/).
 *
 * FASTA handling follows pyani_plus/utils.py:67-90 (fasta_bytes_iterator): text
 * before the first '>' is ignored, a record's sequence is the concatenation of
 * its lines with " \t\r\n" removed; k-mer windows never span records.
 *
 * Build: see oracle/Makefile (gcc -O2 -fopenmp -shared -fPIC).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* ---- MurmurHash3_x64_128 (Appleby, public domain algorithm), first word ---- */

static inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }

static inline uint64_t fmix64(uint64_t k) {
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdULL;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ULL;
  k ^= k >> 33;
  return k;
}

static inline uint64_t load_le64(const uint8_t *p) {
  uint64_t v = 0;
  for (int i = 7; i >= 0; --i) v = (v << 8) | p[i];
  return v;
}

ORC_API uint64_t orc_murmur3_h1(const uint8_t *data, uint32_t len, uint32_t seed) {
  const uint64_t c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;
  uint64_t h1 = seed, h2 = seed;
  const uint32_t nblocks = len / 16;
  for (uint32_t i = 0; i < nblocks; ++i) {
    uint64_t k1 = load_le64(data + 16 * i), k2 = load_le64(data + 16 * i + 8);
    k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
    h1 = rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729;
    k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2;
    h2 = rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5;
  }
  const uint8_t *tail = data + 16 * nblocks;
  uint64_t k1 = 0, k2 = 0;
  const uint32_t rem = len & 15;
  for (uint32_t i = rem; i > 8; --i) k2 |= (uint64_t)tail[i - 1] << (8 * (i - 9));
  if (rem > 8) { k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2; }
  for (uint32_t i = (rem > 8 ? 8 : rem); i > 0; --i) k1 |= (uint64_t)tail[i - 1] << (8 * (i - 1));
  if (rem > 0) { k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1; }
  h1 ^= len; h2 ^= len;
  h1 += h2; h2 += h1;
  h1 = fmix64(h1); h2 = fmix64(h2);
  h1 += h2;
  return h1;
}

/* ---- FracMinHash threshold (sourmash: max_hash for a given `scaled`) ----
 * Fixture `.sig` files pin 61489146912365176 (scaled=300) and
 * 18446744073709552 (scaled=1000): the double-rounded 2^64/scaled. */
ORC_API uint64_t orc_max_hash(uint64_t scaled) {
  if (scaled == 0) return 0;
  if (scaled == 1) return UINT64_MAX;
  return (uint64_t)(18446744073709551616.0 /* 2^64 */ / (double)scaled);
}

/* ---- k-mer hashing of one record ---- */

static inline int base_code(uint8_t c) {
  switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return -1;
  }
}

typedef struct { uint64_t *v; uint64_t n, cap; } u64vec;

static int vec_push(u64vec *a, uint64_t x) {
  if (a->n == a->cap) {
    uint64_t nc = a->cap ? a->cap * 2 : 1024;
    uint64_t *nv = (uint64_t *)realloc(a->v, nc * sizeof(uint64_t));
    if (!nv) return -1;
    a->v = nv; a->cap = nc;
  }
  a->v[a->n++] = x;
  return 0;
}

/* Every window of k residues made only of ACGT (case-insensitive) contributes
 * h = murmur3(min(kmer, revcomp(kmer)) as upper-case ASCII, seed 42).h1 if
 * h <= max_hash.  Deliberately the slow, obvious form: materialise both strands
 * as ASCII and memcmp them. */
static int hash_record(const uint8_t *seq, uint64_t len, uint32_t k, uint64_t max_hash, u64vec *out) {
  if (k == 0 || k > 64 || len < k) return 0;
  uint8_t fwd[64], rev[64];
  uint64_t run = 0; /* consecutive valid residues ending at i */
  static const char up[4] = {'A', 'C', 'G', 'T'};
  for (uint64_t i = 0; i < len; ++i) {
    run = base_code(seq[i]) >= 0 ? run + 1 : 0;
    if (run < k) continue;
    const uint8_t *w = seq + i + 1 - k;
    for (uint32_t j = 0; j < k; ++j) {
      int c = base_code(w[j]);
      fwd[j] = (uint8_t)up[c];
      rev[k - 1 - j] = (uint8_t)up[3 - c];
    }
    const uint8_t *canon = memcmp(fwd, rev, k) <= 0 ? fwd : rev;
    uint64_t h = orc_murmur3_h1(canon, k, 42);
    if (h <= max_hash && vec_push(out, h)) return -1;
  }
  return 0;
}

static int cmp_u64(const void *a, const void *b) {
  uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
  return x < y ? -1 : x > y;
}

static uint64_t sort_unique(uint64_t *v, uint64_t n) {
  if (n == 0) return 0;
  qsort(v, n, sizeof(uint64_t), cmp_u64);
  uint64_t m = 1;
  for (uint64_t i = 1; i < n; ++i)
    if (v[i] != v[m - 1]) v[m++] = v[i];
  return m;
}

/* Sketch a whole FASTA text (already decompressed).  Writes up to `cap` sorted
 * unique hashes to `out`; returns the sketch size (may exceed cap: call again),
 * or -1 on allocation failure.  n_residues_out (optional) = sum of record
 * lengths after whitespace removal (= Genome.length, db_orm.py:832-866). */
ORC_API int64_t orc_sketch_fasta(const uint8_t *text, uint64_t n, uint32_t k, uint64_t max_hash,
                                 uint64_t *out, uint64_t cap, uint64_t *n_residues_out) {
  u64vec hashes = {0, 0, 0};
  uint8_t *seq = (uint8_t *)malloc(n ? n : 1);
  if (!seq) return -1;
  uint64_t total = 0, pos = 0, slen = 0;
  int in_record = 0, rc = 0;
  while (pos < n && rc == 0) {
    uint64_t eol = pos;
    while (eol < n && text[eol] != '\n') ++eol;
    if (text[pos] == '>') { /* title line: flush the previous record */
      if (in_record) { rc = hash_record(seq, slen, k, max_hash, &hashes); total += slen; }
      in_record = 1; slen = 0;
    } else if (in_record) {
      for (uint64_t i = pos; i < eol; ++i) {
        uint8_t c = text[i];
        if (c != ' ' && c != '\t' && c != '\r' && c != '\n') seq[slen++] = c;
      }
    }
    pos = eol + 1;
  }
  if (in_record && rc == 0) { rc = hash_record(seq, slen, k, max_hash, &hashes); total += slen; }
  free(seq);
  if (rc) { free(hashes.v); return -1; }
  uint64_t m = sort_unique(hashes.v, hashes.n);
  for (uint64_t i = 0; i < m && i < cap; ++i) out[i] = hashes.v[i];
  free(hashes.v);
  if (n_residues_out) *n_residues_out = total;
  return (int64_t)m;
}

/* Sketch one bare residue string (one record, no FASTA framing). */
ORC_API int64_t orc_sketch_seq(const uint8_t *seq, uint64_t len, uint32_t k, uint64_t max_hash,
                               uint64_t *out, uint64_t cap) {
  u64vec hashes = {0, 0, 0};
  if (hash_record(seq, len, k, max_hash, &hashes)) { free(hashes.v); return -1; }
  uint64_t m = sort_unique(hashes.v, hashes.n);
  for (uint64_t i = 0; i < m && i < cap; ++i) out[i] = hashes.v[i];
  free(hashes.v);
  return (int64_t)m;
}

/* A tuned scalar CPU form of the same function, used ONLY as the timed
 * `cpu_baseline` ("port") in bench.py so that the GPU is compared with a
 * reasonable CPU implementation rather than with the deliberately naive one
 * above: rolling 2-bit forward / reverse-complement registers, integer
 * canonical compare, ASCII expansion through a 256-entry table.  k <= 32.
 * tests/test_oracle_golden.py checks it equals orc_sketch_seq on every fixture. */
static uint32_t g_ascii4[256];
static int g_ascii4_ready = 0;
static void init_ascii4(void) {
  static const char up[4] = {'A', 'C', 'G', 'T'};
  for (int v = 0; v < 256; ++v) {
    uint32_t w = 0;
    for (int j = 0; j < 4; ++j) w |= (uint32_t)(uint8_t)up[(v >> (2 * j)) & 3] << (8 * j);
    g_ascii4[v] = w;
  }
  g_ascii4_ready = 1;
}

ORC_API int64_t orc_sketch_seq_fast(const uint8_t *seq, uint64_t len, uint32_t k, uint64_t max_hash,
                                    uint64_t *out, uint64_t cap) {
  if (k == 0 || k > 32) return -2;
  if (!g_ascii4_ready) init_ascii4();
  u64vec hashes = {0, 0, 0};
  const uint64_t kmask = k == 32 ? ~0ULL : ((1ULL << (2 * k)) - 1);
  const int top = 2 * ((int)k - 1);
  uint64_t f_msb = 0, f_lsb = 0, run = 0;
  uint8_t buf[32] __attribute__((aligned(8)));
  for (uint64_t i = 0; i < len; ++i) {
    const int c = base_code(seq[i]);
    if (c < 0) { run = 0; f_msb = f_lsb = 0; continue; }
    f_msb = ((f_msb << 2) | (uint64_t)c) & kmask;   /* base 0 of the window in the top bits */
    f_lsb = (f_lsb >> 2) | ((uint64_t)c << top);    /* base j of the window at bits 2j */
    if (++run < k) continue;
    /* revcomp: MSB-first form is ~f_lsb, LSB-first form is ~f_msb */
    const uint64_t canon = f_msb <= (f_lsb ^ kmask) ? f_lsb : (f_msb ^ kmask);
    uint32_t w[8];
    for (int d = 0; d < 8; ++d) w[d] = g_ascii4[(canon >> (8 * d)) & 0xff];
    memcpy(buf, w, 32);
    const uint64_t h = orc_murmur3_h1(buf, k, 42);
    if (h <= max_hash && vec_push(&hashes, h)) { free(hashes.v); return -1; }
  }
  uint64_t m = sort_unique(hashes.v, hashes.n);
  for (uint64_t i = 0; i < m && i < cap; ++i) out[i] = hashes.v[i];
  free(hashes.v);
  return (int64_t)m;
}

/* Many bare sequences, one OpenMP task per sequence (cpu_baseline leg).
 * seqs = concatenated residues, seq_off[n+1]; out = caller buffer with
 * out_off[n+1] capacities; sizes[n] receives the sketch sizes. */
ORC_API int orc_sketch_many(const uint8_t *seqs, const uint64_t *seq_off, uint32_t n, uint32_t k,
                            uint64_t max_hash, uint64_t *out, const uint64_t *out_off, int64_t *sizes,
                            int threads, int fast) {
  int bad = 0;
  if (!g_ascii4_ready) init_ascii4();
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : 1)
  for (uint32_t g = 0; g < n; ++g) {
    sizes[g] = (fast ? orc_sketch_seq_fast : orc_sketch_seq)(seqs + seq_off[g], seq_off[g + 1] - seq_off[g], k, max_hash,
                              out + out_off[g], out_off[g + 1] - out_off[g]);
    if (sizes[g] < 0 || (uint64_t)sizes[g] > out_off[g + 1] - out_off[g]) bad = 1;
  }
  return bad ? -1 : 0;
}

/* ---- pairs: |A ∩ B| of two ascending duplicate-free lists ---- */
ORC_API uint32_t orc_intersect(const uint64_t *a, uint64_t na, const uint64_t *b, uint64_t nb) {
  uint64_t i = 0, j = 0;
  uint32_t c = 0;
  while (i < na && j < nb) {
    if (a[i] < b[j]) ++i;
    else if (a[i] > b[j]) ++j;
    else { ++c; ++i; ++j; }
  }
  return c;
}

/* counts[q*ns + s] for queries [q0,q1) x subjects [s0,s1) of a CSR sketch set. */
ORC_API void orc_pair_counts(const uint64_t *hashes, const uint64_t *off, uint32_t q0, uint32_t q1,
                             uint32_t s0, uint32_t s1, uint32_t *counts, int threads) {
  const uint32_t ns = s1 - s0;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : 1)
  for (uint32_t q = q0; q < q1; ++q)
    for (uint32_t s = s0; s < s1; ++s)
      counts[(uint64_t)(q - q0) * ns + (s - s0)] =
          orc_intersect(hashes + off[q], off[q + 1] - off[q], hashes + off[s], off[s + 1] - off[s]);
}

/* ---- containment -> ANI (manysearch columns read at sourmash.py:107-110) ----
 * query_containment_ani = (I/|Q|)^(1/k), match_containment_ani = (I/|M|)^(1/k),
 * max_containment_ani = max of the two; I == 0 -> row absent -> NULL
 * (sourmash.py:141-144).  Host libm pow reproduces all 27 fixture rows.
 * identity = max_containment_ani, cov_query = query_containment_ani
 * (private_cli.py:1879-1880). */
ORC_API void orc_ani(const uint32_t *counts, const uint64_t *q_sizes, const uint64_t *s_sizes,
                     uint32_t nq, uint32_t ns, uint32_t k, double *identity, double *cov_query,
                     uint8_t *is_null) {
  const double inv_k = 1.0 / (double)k;
  for (uint32_t q = 0; q < nq; ++q)
    for (uint32_t s = 0; s < ns; ++s) {
      const uint64_t idx = (uint64_t)q * ns + s;
      const uint32_t c = counts[idx];
      if (c == 0) {
        identity[idx] = NAN; cov_query[idx] = NAN; is_null[idx] = 1;
        continue;
      }
      const double qa = pow((double)c / (double)q_sizes[q], inv_k);
      const double ma = pow((double)c / (double)s_sizes[s], inv_k);
      identity[idx] = qa > ma ? qa : ma;
      cov_query[idx] = qa;
      is_null[idx] = 0;
    }
}

/* ---- bottom-m MinHash (BASELINE.json configs[1] names it; the REFERENCE NEVER USES IT: every fixture
 * has "num":0 and the only sketch parameter is scaled=N, pyani_plus/methods/sourmash.py:75-76).
 * PARITY UNPINNED: these three functions restate the published Mash estimator (Ondov et al. 2016) and
 * are checked only against each other and the HIP path. ---- */

/* the m smallest distinct canonical k-mer hashes of one bare sequence */
ORC_API int64_t orc_sketch_bottom_seq(const uint8_t *seq, uint64_t len, uint32_t k, uint64_t m, uint64_t *out) {
  u64vec hashes = {0, 0, 0};
  if (hash_record(seq, len, k, UINT64_MAX, &hashes)) { free(hashes.v); return -1; }
  uint64_t n = sort_unique(hashes.v, hashes.n);
  if (n > m) n = m;
  for (uint64_t i = 0; i < n; ++i) out[i] = hashes.v[i];
  free(hashes.v);
  return (int64_t)n;
}

/* Mash Jaccard of two bottom-m sketches: among the (up to) m smallest elements of A u B, how many
 * are in both.  *denom = elements of the union examined = min(m, |A u B|). */
ORC_API void orc_mash_pair(const uint64_t *a, uint64_t na, const uint64_t *b, uint64_t nb, uint64_t m,
                           uint32_t *common, uint32_t *denom) {
  uint64_t i = 0, j = 0, taken = 0;
  uint32_t c = 0;
  while (taken < m && (i < na || j < nb)) {
    if (j >= nb || (i < na && a[i] < b[j])) ++i;
    else if (i >= na || b[j] < a[i]) ++j;
    else { ++c; ++i; ++j; }
    ++taken;
  }
  *common = c;
  *denom = (uint32_t)taken;
}

/* Mash distance -> ANI: j = common/denom, d = -ln(2j/(1+j))/k, ANI = 1 - d; common == 0 -> NaN (NULL) */
ORC_API double orc_mash_ani(uint32_t common, uint32_t denom, uint32_t k) {
  if (common == 0 || denom == 0) return NAN;
  if (common == denom) return 1.0;
  const double j = (double)common / (double)denom;
  return 1.0 + log(2.0 * j / (1.0 + j)) / (double)k;
}
