"""ctypes binding of oracle/_build/liboracle.so (test infrastructure only)."""

from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB_PATH = _HERE / "_build" / "liboracle.so"
_lib = None

_u8p = C.POINTER(C.c_uint8)
_u32p = C.POINTER(C.c_uint32)
_u64p = C.POINTER(C.c_uint64)
_i64p = C.POINTER(C.c_int64)
_f64p = C.POINTER(C.c_double)


def build(force: bool = False) -> Path:
    """Compile the C restatement with gcc (a few hundred ms)."""
    newest = max((_HERE / name).stat().st_mtime for name in ("sourmash_oracle.c", "fragani_oracle.c"))
    if force or not _LIB_PATH.is_file() or _LIB_PATH.stat().st_mtime < newest:
        subprocess.run(["make", "-C", str(_HERE), "-s", "-B"], check=True)
    return _LIB_PATH


def _load() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        lib = C.CDLL(str(_LIB_PATH))
        lib.orc_murmur3_h1.restype = C.c_uint64
        lib.orc_murmur3_h1.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32]
        lib.orc_max_hash.restype = C.c_uint64
        lib.orc_max_hash.argtypes = [C.c_uint64]
        lib.orc_sketch_fasta.restype = C.c_int64
        lib.orc_sketch_fasta.argtypes = [C.c_char_p, C.c_uint64, C.c_uint32, C.c_uint64, _u64p, C.c_uint64, _u64p]
        lib.orc_sketch_seq.restype = C.c_int64
        lib.orc_sketch_seq.argtypes = [C.c_char_p, C.c_uint64, C.c_uint32, C.c_uint64, _u64p, C.c_uint64]
        lib.orc_sketch_many.restype = C.c_int
        lib.orc_sketch_many.argtypes = [_u8p, _u64p, C.c_uint32, C.c_uint32, C.c_uint64, _u64p, _u64p, _i64p, C.c_int, C.c_int]
        lib.orc_intersect.restype = C.c_uint32
        lib.orc_intersect.argtypes = [_u64p, C.c_uint64, _u64p, C.c_uint64]
        lib.orc_pair_counts.restype = None
        lib.orc_pair_counts.argtypes = [_u64p, _u64p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _u32p, C.c_int]
        lib.orc_ani.restype = None
        lib.orc_ani.argtypes = [_u32p, _u64p, _u64p, C.c_uint32, C.c_uint32, C.c_uint32, _f64p, _f64p, _u8p]
        _lib = lib
    return _lib


def _p(a: np.ndarray, t):
    return a.ctypes.data_as(t)


def murmur3_h1(data: bytes, seed: int = 42) -> int:
    return int(_load().orc_murmur3_h1(data, len(data), seed))


def max_hash(scaled: int) -> int:
    return int(_load().orc_max_hash(scaled))


def sketch_fasta_text(text: bytes, k: int, scaled: int) -> tuple[np.ndarray, int]:
    """(sorted unique hashes, total residues) of a decompressed FASTA text."""
    lib = _load()
    cap = max(1024, len(text) // max(1, scaled) * 2 + 1024)
    total = C.c_uint64(0)
    while True:
        out = np.empty(cap, dtype=np.uint64)
        n = lib.orc_sketch_fasta(text, len(text), k, max_hash(scaled), _p(out, _u64p), cap, C.byref(total))
        if n < 0:
            raise MemoryError("oracle allocation failed")
        if n <= cap:
            return out[:n].copy(), int(total.value)
        cap = int(n)


def sketch_seq(seq: bytes, k: int, scaled: int) -> np.ndarray:
    lib = _load()
    cap = max(1024, len(seq) // max(1, scaled) * 2 + 1024)
    while True:
        out = np.empty(cap, dtype=np.uint64)
        n = lib.orc_sketch_seq(seq, len(seq), k, max_hash(scaled), _p(out, _u64p), cap)
        if n < 0:
            raise MemoryError("oracle allocation failed")
        if n <= cap:
            return out[:n].copy()
        cap = int(n)


def sketch_many(seqs: list[bytes] | list[np.ndarray], k: int, scaled: int, threads: int = 1, fast: bool = False) -> list[np.ndarray]:
    """Sketch bare residue strings, one OpenMP task each (fast=True: the tuned scalar form, which holds a k-mer in one
    64-bit word: k above 32 takes the plain form whatever ``fast`` says)."""
    lib = _load()
    fast = fast and k <= 32
    lens = np.array([len(s) for s in seqs], dtype=np.uint64)
    off = np.zeros(len(seqs) + 1, dtype=np.uint64)
    np.cumsum(lens, out=off[1:])
    flat = np.empty(int(off[-1]), dtype=np.uint8)
    for i, s in enumerate(seqs):
        flat[int(off[i]) : int(off[i + 1])] = np.frombuffer(s, dtype=np.uint8) if isinstance(s, (bytes, bytearray)) else s
    caps = np.maximum(1024, lens // max(1, scaled) * 2 + 1024).astype(np.uint64)
    while True:
        ooff = np.zeros(len(seqs) + 1, dtype=np.uint64)
        np.cumsum(caps, out=ooff[1:])
        out = np.empty(int(ooff[-1]), dtype=np.uint64)
        sizes = np.zeros(len(seqs), dtype=np.int64)
        rc = lib.orc_sketch_many(_p(flat, _u8p), _p(off, _u64p), len(seqs), k, max_hash(scaled), _p(out, _u64p), _p(ooff, _u64p), _p(sizes, _i64p), threads, int(fast))
        if (sizes < 0).any():
            raise MemoryError("oracle allocation failed")
        if rc == 0:
            return [out[int(ooff[i]) : int(ooff[i]) + int(sizes[i])].copy() for i in range(len(seqs))]
        caps = np.maximum(caps, sizes.astype(np.uint64))


def intersect(a: np.ndarray, b: np.ndarray) -> int:
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = np.ascontiguousarray(b, dtype=np.uint64)
    return int(_load().orc_intersect(_p(a, _u64p), len(a), _p(b, _u64p), len(b)))


def _csr(sketches: list[np.ndarray]) -> tuple[np.ndarray, np.ndarray]:
    off = np.zeros(len(sketches) + 1, dtype=np.uint64)
    np.cumsum([len(s) for s in sketches], out=off[1:])
    flat = np.concatenate([np.asarray(s, dtype=np.uint64) for s in sketches]) if sketches else np.empty(0, np.uint64)
    return np.ascontiguousarray(flat, dtype=np.uint64), off


def pair_counts(sketches: list[np.ndarray], q_range=None, s_range=None, threads: int = 1) -> np.ndarray:
    """counts[q, s] = |sketch q ∩ sketch s| for the given index ranges (default all)."""
    flat, off = _csr(sketches)
    n = len(sketches)
    q0, q1 = q_range or (0, n)
    s0, s1 = s_range or (0, n)
    counts = np.zeros((q1 - q0, s1 - s0), dtype=np.uint32)
    if flat.size == 0:
        flat = np.zeros(1, dtype=np.uint64)
    _load().orc_pair_counts(_p(flat, _u64p), _p(off, _u64p), q0, q1, s0, s1, _p(counts, _u32p), threads)
    return counts


def ani(counts: np.ndarray, q_sizes, s_sizes, k: int) -> tuple[np.ndarray, np.ndarray, np.ndarray]:
    """(identity, cov_query, is_null) as the reference maps manysearch rows."""
    counts = np.ascontiguousarray(counts, dtype=np.uint32)
    nq, ns = counts.shape
    q_sizes = np.ascontiguousarray(q_sizes, dtype=np.uint64)
    s_sizes = np.ascontiguousarray(s_sizes, dtype=np.uint64)
    ident = np.empty((nq, ns), dtype=np.float64)
    cov = np.empty((nq, ns), dtype=np.float64)
    null = np.empty((nq, ns), dtype=np.uint8)
    _load().orc_ani(_p(counts, _u32p), _p(q_sizes, _u64p), _p(s_sizes, _u64p), nq, ns, k, _p(ident, _f64p), _p(cov, _f64p), _p(null, _u8p))
    return ident, cov, null.astype(bool)


# ---------------------------------------------------------------- fastANI-style fragment ANI
_frag_typed = False


def _load_frag() -> C.CDLL:
    global _frag_typed
    lib = _load()
    if not _frag_typed:
        i32p = C.POINTER(C.c_int32)
        lib.orc_fragani_window_size.restype = C.c_int
        lib.orc_fragani_window_size.argtypes = [C.c_int, C.c_int]
        lib.orc_fragani_min_hits.restype = C.c_int
        lib.orc_fragani_min_hits.argtypes = [C.c_int, C.c_int]
        lib.orc_fragani_min_shared.restype = C.c_int
        lib.orc_fragani_min_shared.argtypes = [C.c_int, C.c_int]
        lib.orc_fragani_identity.restype = C.c_double
        lib.orc_fragani_identity.argtypes = [C.c_int, C.c_int, C.c_int]
        lib.orc_fragani_kmer_hash.restype = C.c_uint32
        lib.orc_fragani_kmer_hash.argtypes = [C.c_char_p, C.c_int]
        lib.orc_fragani_minimizers.restype = C.c_int64
        lib.orc_fragani_minimizers.argtypes = [C.c_char_p, C.c_uint64, C.c_int, C.c_int, _u32p, i32p, C.c_uint64]
        lib.orc_fragani_map.restype = C.c_int
        lib.orc_fragani_map.argtypes = [C.c_char_p, _u64p, C.c_uint32, C.c_char_p, _u64p, C.c_uint32, C.c_int, C.c_int, C.c_int, i32p, i32p, i32p, i32p, i32p, C.POINTER(C.c_int)]
        lib.orc_fragani_pair.restype = C.c_int
        lib.orc_fragani_pair.argtypes = [C.c_char_p, _u64p, C.c_uint32, C.c_char_p, _u64p, C.c_uint32, C.c_int, C.c_int, C.c_double, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        lib.orc_fragani_set_option.restype = None
        lib.orc_fragani_set_option.argtypes = [C.c_int, C.c_double]
        lib.orc_fragani_get_option.restype = C.c_double
        lib.orc_fragani_get_option.argtypes = [C.c_int]
        _frag_typed = True
    return lib


FRAGANI_OPTIONS = {"window_rule": 0, "bin_rule": 1, "l2_rule": 2, "conf": 3, "l2_pos": 4, "l2_stop": 5, "tie": 6, "freq": 7, "float": 8}


def fragani_set_option(name: str, value: float) -> None:
    """Switch one of the oracle's restatement choices (tests/tools/fragani_bisect.py); process-wide."""
    _load_frag().orc_fragani_set_option(FRAGANI_OPTIONS[name], float(value))


def fragani_set_fast(on: bool) -> None:
    """The tuned form of the L2 evaluation (the window kept as the slide moves) instead of the checking form; process-wide.
    bench.py's CPU-baseline leg times it; parity checks use the checking form (tests/test_fragani_oracle.py holds the two equal)."""
    lib = _load_frag()
    lib.orc_fragani_set_fast.restype = None
    lib.orc_fragani_set_fast.argtypes = [C.c_int]
    lib.orc_fragani_set_fast(1 if on else 0)


def fragani_get_option(name: str) -> float:
    return float(_load_frag().orc_fragani_get_option(FRAGANI_OPTIONS[name]))


def fragani_window_size(k: int, frag_len: int) -> int:
    return int(_load_frag().orc_fragani_window_size(k, frag_len))


def fragani_tables(k: int, s_max: int) -> tuple[np.ndarray, np.ndarray]:
    """(min_hits[s], min_shared[s]) for s = 0..s_max (entry 0 unused)."""
    lib = _load_frag()
    mh = np.array([0] + [lib.orc_fragani_min_hits(s, k) for s in range(1, s_max + 1)], dtype=np.int32)
    ms = np.array([0] + [lib.orc_fragani_min_shared(s, k) for s in range(1, s_max + 1)], dtype=np.int32)
    return mh, ms


def fragani_identity(shared: int, s: int, k: int) -> float:
    return float(_load_frag().orc_fragani_identity(shared, s, k))


def fragani_kmer_hash(kmer: bytes) -> int:
    return int(_load_frag().orc_fragani_kmer_hash(kmer, len(kmer)))


def fragani_minimizers(seq: bytes, k: int, w: int) -> tuple[np.ndarray, np.ndarray]:
    lib = _load_frag()
    cap = max(16, len(seq))
    h = np.empty(cap, dtype=np.uint32)
    p = np.empty(cap, dtype=np.int32)
    n = lib.orc_fragani_minimizers(seq, len(seq), k, w, _p(h, _u32p), p.ctypes.data_as(C.POINTER(C.c_int32)), cap)
    if n < 0:
        raise MemoryError("oracle allocation failed")
    return h[:n].copy(), p[:n].copy()


def _contig_blob(contigs: list[bytes]) -> tuple[bytes, np.ndarray]:
    off = np.zeros(len(contigs) + 1, dtype=np.uint64)
    np.cumsum([len(c) for c in contigs], out=off[1:])
    return b"".join(contigs), off


def fragani_map(query: list[bytes], ref: list[bytes], k: int = 16, frag_len: int = 3000, window: int = 0):
    """Per-fragment mappings of a genome pair: dict of int32 arrays (frag, ref_seq, ref_pos, shared, s) and total."""
    lib = _load_frag()
    qs, qo = _contig_blob(query)
    rs, ro = _contig_blob(ref)
    total_cap = int(sum(len(c) // frag_len for c in query)) + 1
    arrs = [np.zeros(total_cap, dtype=np.int32) for _ in range(5)]
    total = C.c_int(0)
    i32p = C.POINTER(C.c_int32)
    n = lib.orc_fragani_map(qs, _p(qo, _u64p), len(query), rs, _p(ro, _u64p), len(ref), k, frag_len, window, *[a.ctypes.data_as(i32p) for a in arrs], C.byref(total))
    if n < 0:
        raise MemoryError("oracle allocation failed")
    names = ("frag", "ref_seq", "ref_pos", "shared", "s")
    return {name: a[:n].copy() for name, a in zip(names, arrs)}, int(total.value)


def fragani_pair(query: list[bytes], ref: list[bytes], k: int = 16, frag_len: int = 3000, min_fraction: float = 0.2, window: int = 0):
    """(ANI percent or NaN, kept fragments, total fragments) of one ordered genome pair."""
    lib = _load_frag()
    qs, qo = _contig_blob(query)
    rs, ro = _contig_blob(ref)
    ani_v, m, t = C.c_double(0), C.c_int(0), C.c_int(0)
    rc = lib.orc_fragani_pair(qs, _p(qo, _u64p), len(query), rs, _p(ro, _u64p), len(ref), k, frag_len, min_fraction, window, C.byref(ani_v), C.byref(m), C.byref(t))
    if rc:
        raise MemoryError("oracle allocation failed")
    return float(ani_v.value), int(m.value), int(t.value)


def fragani_many(queries: list[list[bytes]], ref: list[bytes], k: int = 16, frag_len: int = 3000, min_fraction: float = 0.2,
                 window: int = 0, threads: int = 0):
    """``fragani_pair`` of every query against ONE reference whose index is built once, queries on ``threads`` OpenMP
    threads (0 = all): the shape of the reference's ``fastANI --ql queries -r subject`` call.  Returns three arrays
    (ANI percent or NaN, kept fragments, total fragments)."""
    lib = _load_frag()
    lib.orc_fragani_many.restype = C.c_int
    lib.orc_fragani_many.argtypes = [C.c_char_p, _u64p, C.c_uint32, C.c_uint32, C.POINTER(C.c_char_p), C.POINTER(_u64p), _u32p, C.c_int,
                                     C.c_int, C.c_double, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    rs, ro = _contig_blob(ref)
    n = len(queries)
    blobs = [_contig_blob(q) for q in queries]
    q_seqs = (C.c_char_p * max(n, 1))(*[b[0] for b in blobs])
    q_offs = (_u64p * max(n, 1))(*[_p(b[1], _u64p) for b in blobs])
    q_contigs = np.array([len(q) for q in queries], dtype=np.uint32)
    ani = np.zeros(max(n, 1), dtype=np.float64)
    matched = np.zeros(max(n, 1), dtype=np.int32)
    total = np.zeros(max(n, 1), dtype=np.int32)
    rc = lib.orc_fragani_many(rs, _p(ro, _u64p), len(ref), n, q_seqs, q_offs, _p(q_contigs, _u32p), k, frag_len, min_fraction, window, threads,
                              ani.ctypes.data_as(C.POINTER(C.c_double)), matched.ctypes.data_as(C.POINTER(C.c_int)),
                              total.ctypes.data_as(C.POINTER(C.c_int)))
    if rc:
        raise MemoryError("oracle allocation failed")
    return ani[:n], matched[:n], total[:n]


# ---------------------------------------------------------------- bottom-m MinHash (parity unpinned)
def sketch_bottom_seq(seq: bytes, k: int, m: int) -> np.ndarray:
    lib = _load()
    lib.orc_sketch_bottom_seq.restype = C.c_int64
    lib.orc_sketch_bottom_seq.argtypes = [C.c_char_p, C.c_uint64, C.c_uint32, C.c_uint64, _u64p]
    out = np.empty(max(m, 1), dtype=np.uint64)
    n = lib.orc_sketch_bottom_seq(seq, len(seq), k, m, _p(out, _u64p))
    if n < 0:
        raise MemoryError("oracle allocation failed")
    return out[:n].copy()


def mash_pairs(sketches: list[np.ndarray], m: int) -> tuple[np.ndarray, np.ndarray]:
    """(common, denom) uint32 matrices of the Mash Jaccard estimator over all ordered pairs."""
    lib = _load()
    lib.orc_mash_pair.restype = None
    lib.orc_mash_pair.argtypes = [_u64p, C.c_uint64, _u64p, C.c_uint64, C.c_uint64, _u32p, _u32p]
    n = len(sketches)
    common = np.zeros((n, n), dtype=np.uint32)
    denom = np.zeros((n, n), dtype=np.uint32)
    sk = [np.ascontiguousarray(s, dtype=np.uint64) for s in sketches]
    c, d = C.c_uint32(0), C.c_uint32(0)
    for q in range(n):
        for s in range(n):
            lib.orc_mash_pair(_p(sk[q], _u64p), len(sk[q]), _p(sk[s], _u64p), len(sk[s]), m, C.byref(c), C.byref(d))
            common[q, s], denom[q, s] = c.value, d.value
    return common, denom


def mash_ani(common: np.ndarray, denom: np.ndarray, k: int) -> np.ndarray:
    lib = _load()
    lib.orc_mash_ani.restype = C.c_double
    lib.orc_mash_ani.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
    out = np.empty(common.shape, dtype=np.float64)
    for idx in np.ndindex(common.shape):
        out[idx] = lib.orc_mash_ani(int(common[idx]), int(denom[idx]), k)
    return out
