"""GPU parity on seeded random inputs and the reference-tested edge cases (through the C ABI).

Integer work (hashes, sketches, intersection counts) must be bit-exact against the oracle;
the device ANI transform is f64 within 1 ulp (rtol 2.3e-16) of host libm, which itself
reproduces the reference fixtures bit for bit (tests/test_gpu_parity_fixtures.py).
"""

from __future__ import annotations

import numpy as np
import pytest

import oracle
from pyani_plus_amd.synth import arena_to_ascii, synth_arena_numpy

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    from pyani_plus_amd.engine import HipEngine

    eng = HipEngine(0)
    yield eng
    eng.close()


@pytest.fixture(scope="module")
def tools_engine():
    """The -DPA_TOOLS build of the library: the environment switches that force a rare path exist there only."""
    from pyani_plus_amd.engine import HipEngine

    eng = HipEngine(0, tools=True)
    yield eng
    eng.close()


def _random_fasta(rng, n_records, max_len, *, lower=False, n_runs=0) -> bytes:
    out = []
    for r in range(n_records):
        length = int(rng.integers(0, max_len))
        seq = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=length)
        for _ in range(n_runs):
            if length > 50:
                p = int(rng.integers(0, length - 5))
                seq[p : p + int(rng.integers(1, 40))] = ord("N")
        if lower:
            mask = rng.random(length) < 0.3
            seq[mask] = seq[mask] + 32
        body = seq.tobytes()
        lines = [body[i : i + 70] for i in range(0, len(body), 70)]
        out.append(b">rec%d some description\n" % r + b"\n".join(lines) + b"\n")
    return b"".join(out)


@pytest.mark.parametrize("k", [15, 16, 21, 31, 32, 7, 20, 24, 30, 33, 51, 64])
@pytest.mark.parametrize("scaled", [1, 7, 1000])
def test_sketch_random_fasta_matches_oracle(engine, k, scaled):
    from pyani_plus_amd.engine import pack_genomes

    rng = np.random.default_rng(1000 * k + scaled)
    texts = [
        _random_fasta(rng, 1, 30000),
        _random_fasta(rng, 3, 8000, n_runs=4),
        _random_fasta(rng, 2, 5000, lower=True, n_runs=2),
        b"",  # empty file
        b">short\nACGTACGTAC\n",  # shorter than k
        b">exact\n" + rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=64 * 50).tobytes() + b"\n",  # multiple of 64
        b">allN\n" + b"N" * 500 + b"\n",
        b"no header at all\nACGTACGT\n",
    ]
    arena = pack_genomes(texts)
    got = engine.sketch(engine.upload(arena), k, scaled).to_host()
    for g, text in enumerate(texts):
        want, total = oracle.sketch_fasta_text(text, k, scaled)
        assert arena.residues[g] == total
        assert np.array_equal(got[g], want), f"genome {g}: {len(got[g])} vs {len(want)} (k={k}, scaled={scaled})"


def test_sketch_every_kmer_size_matches_oracle(engine):
    """The reference hands any --kmersize to sourmash (pyani_plus/public_cli_args.py:229,
    pyani_plus/methods/sourmash.py:75-76): every k from 1 to 32 is compiled into the tuned kernel, 33 to 64 (sourmash's
    own third default is 51) take the plain 128-bit form; all of them equal the oracle, and beyond 64 is refused."""
    from pyani_plus_amd.engine import pack_genomes

    rng = np.random.default_rng(99)
    texts = [_random_fasta(rng, 2, 6000, n_runs=3), _random_fasta(rng, 1, 9000, lower=True), b">tiny\nACGTTGCA\n"]
    dev = engine.upload(pack_genomes(texts))
    for k in range(1, 65):
        got = engine.sketch(dev, k, 3).to_host()
        for g, text in enumerate(texts):
            want, _total = oracle.sketch_fasta_text(text, k, 3)
            assert np.array_equal(got[g], want), f"k={k} genome {g}: {len(got[g])} vs {len(want)}"
    from pyani_plus_amd._capi import HipBackendError

    with pytest.raises(HipBackendError, match="outside"):
        engine.sketch(dev, 65, 3)


def test_sketch_staging_overflow_and_capacity_retry(engine):
    """scaled=1 floods the LDS staging buffer; a low-complexity genome overruns the candidate estimate."""
    from pyani_plus_amd.engine import pack_genomes

    rng = np.random.default_rng(7)
    k = 31
    # find a 31-mer whose canonical hash passes scaled=1000
    thresh = oracle.max_hash(1000)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    while True:
        kmer = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=k).tobytes()
        canon = min(kmer, kmer.translate(comp)[::-1])
        if oracle.murmur3_h1(canon, 42) <= thresh:
            break
    repeat = kmer * (4_200_000 // k)  # every 31st window is that k-mer: ~135 000 candidates, estimate ~70 000
    rand = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=300_000).tobytes()
    arena = pack_genomes([repeat, rand], fasta=False)
    got = engine.sketch(engine.upload(arena), k, 1000).to_host()
    assert np.array_equal(got[0], oracle.sketch_seq(repeat, k, 1000)) and got[0].size >= 1
    assert np.array_equal(got[1], oracle.sketch_seq(rand, k, 1000))
    # a repeat small enough for the genome's candidate region: the LDS sort sees one value range with far more
    # keys than its per-thread insertion sort takes and switches to the bitonic network
    few = kmer * 150 + rand[:200_000]
    arena = pack_genomes([few, rand[:100_000]], fasta=False)
    got = engine.sketch(engine.upload(arena), k, 1000).to_host()
    assert np.array_equal(got[0], oracle.sketch_seq(few, k, 1000)) and np.array_equal(got[1], oracle.sketch_seq(rand[:100_000], k, 1000))
    # scaled=1: every window survives -> staging overflow path, duplicates collapse
    small = pack_genomes([rand[:50_000], (b"ACGT" * 3000)], fasta=False)
    got = engine.sketch(engine.upload(small), k, 1).to_host()
    assert np.array_equal(got[0], oracle.sketch_seq(rand[:50_000], k, 1))
    assert np.array_equal(got[1], oracle.sketch_seq(b"ACGT" * 3000, k, 1)) and got[1].size <= 4


def test_sketch_synthetic_arena_and_idempotence(engine):
    arena = synth_arena_numpy(12, [200_000, 64, 0 + 100, 150_000, 64 * 1000, 99_999, 31, 30, 250_000, 5, 128, 77_777], n_species=3)
    dev = engine.upload(arena)
    a = engine.sketch(dev, 31, 100).to_host()
    b = engine.sketch(dev, 31, 100).to_host()
    for g in range(arena.n_genomes):
        want = oracle.sketch_seq(arena_to_ascii(arena, g), 31, 100)
        assert np.array_equal(a[g], want) and np.array_equal(a[g], b[g])


def _random_sketches(rng, sizes, universe):
    pool = np.unique(rng.integers(0, 2**63, size=universe, dtype=np.uint64))
    return [np.sort(rng.choice(pool, size=min(s, pool.size), replace=False)) for s in sizes]


@pytest.mark.parametrize(
    "sizes,universe",
    [
        ([0, 1, 5, 64, 65, 300, 0, 1000], 1500),  # empty sketches, tiny, dense overlap
        (list(range(0, 140)), 400),  # 140 subjects -> two-thread rows
        ([50] * 300, 2000),  # 300 subjects -> three-thread rows (384 columns)
        ([20] * 1250, 3000),  # 1250 subjects -> ten-thread rows: the per-rank tile of the 8-GPU configuration
        ([12] * 1700, 2500),  # 14-thread rows, 18 whole rows per iteration and 4 idle threads
        ([8] * 2100, 3000),  # > 2048 subjects -> two subject tiles
        ([70_000, 500, 66_000], 90_000),  # > 255 rows per lane -> vertical-counter flush
    ],
)
def test_pair_counts_match_oracle(engine, sizes, universe):
    rng = np.random.default_rng(len(sizes) + universe)
    sketches = _random_sketches(rng, sizes, universe)
    sk = engine.sketches_from_host(sketches)
    want = oracle.pair_counts(sketches, threads=8)
    for algo in (1, 2, 3):
        got = engine.pair_counts(sk, algo=algo).cpu().numpy().view(np.uint32)
        assert np.array_equal(got, want), f"algo {algo}"
    n = len(sketches)
    # rectangular tiles (a subject column, a query band)
    for q_range, s_range in (((0, n), (n - 1, n)), ((1, min(n, 5)), (0, n)), ((2, 3), (1, 2))):
        for algo in (1, 2, 3):
            got = engine.pair_counts(sk, q_range, s_range, algo=algo).cpu().numpy().view(np.uint32)
            assert np.array_equal(got, want[q_range[0] : q_range[1], s_range[0] : s_range[1]])


def test_pair_counts_with_extreme_hash_values(engine):
    """0 and 2^64-1 are legal hashes; the hash dictionary keeps the latter (its empty marker) aside."""
    top = np.uint64(2**64 - 1)
    sketches = [
        np.array([0, 5, top], dtype=np.uint64),
        np.array([top], dtype=np.uint64),
        np.array([0, 7], dtype=np.uint64),
        np.array([], dtype=np.uint64),
        np.array([5, 7, 2**63, top], dtype=np.uint64),
    ]
    sk = engine.sketches_from_host(sketches)
    want = oracle.pair_counts(sketches)
    for algo in (0, 1, 2, 3):
        assert np.array_equal(engine.pair_counts(sk, algo=algo).cpu().numpy().view(np.uint32), want), f"algo {algo}"
    got = engine.pair_counts(sk, (0, 5), (2, 4), algo=3).cpu().numpy().view(np.uint32)  # tile without the top hash
    assert np.array_equal(got, want[:, 2:4])


def test_device_ani_within_one_ulp_of_libm(engine):
    from pyani_plus_amd.engine import ani_host

    rng = np.random.default_rng(11)
    sketches = _random_sketches(rng, rng.integers(1, 3000, size=200).tolist(), 6000)
    sk = engine.sketches_from_host(sketches)
    counts_t = engine.pair_counts(sk)
    counts = counts_t.cpu().numpy().view(np.uint32)
    sizes = [len(s) for s in sketches]
    for k in (21, 31):
        ident, cov, null = ani_host(counts, sizes, sizes, k)
        d_ident, d_cov = (x.cpu().numpy() for x in engine.ani(counts_t, sk, k))
        assert np.array_equal(np.isnan(d_ident), null) and np.array_equal(np.isnan(d_cov), null)
        for got, want in ((d_ident, ident), (d_cov, cov)):
            rel = np.abs(got[~null] - want[~null]) / want[~null]
            assert rel.max() <= 2.3e-16, rel.max()
        assert np.all(np.diag(d_ident) == 1.0) and np.all(np.diag(d_cov) == 1.0)
        assert np.array_equal(d_ident, d_ident.T, equal_nan=True)  # max-containment is symmetric


def test_full_size_properties_baseline_config(engine):
    """BASELINE configs[1] at full size: 1000 x 5 Mb, k=31, scaled=1000 (size-independent properties)."""
    from pyani_plus_amd.synth import device_arena_to_host, synth_arena_torch

    n, length, k, scaled = 1000, 5_000_000, 31, 1000
    arena = synth_arena_torch(engine, n, length)
    sk = engine.sketch(arena, k, scaled)
    sk2 = engine.sketch(arena, k, scaled)
    t = engine.torch
    assert sk.total == sk2.total and t.equal(sk.hashes[: sk.total], sk2.hashes[: sk2.total]) and t.equal(sk.off, sk2.off)
    off = sk.off.cpu().numpy()
    sizes = np.diff(off)
    assert sizes.min() > 4500 and sizes.max() < 5500  # ~L/scaled
    flat = sk.hashes[: sk.total].cpu().numpy().view(np.uint64)
    assert flat.max() <= oracle.max_hash(scaled)
    for g in range(0, n, 97):  # ascending and duplicate-free inside every sampled sketch
        assert np.all(np.diff(flat[off[g] : off[g + 1]].astype(np.float64)) > 0)
    counts = engine.pair_counts(sk)
    c = counts.cpu().numpy().view(np.uint32)
    assert np.array_equal(c, c.T) and np.array_equal(np.diag(c), sizes.astype(np.uint32))
    merged = engine.pair_counts(sk, algo=2).cpu().numpy().view(np.uint32)  # independent second kernel, full 10^6 pairs
    assert np.array_equal(c, merged)
    # genomes of the same species overlap, different species do not
    assert c[0, 40] > 3000 and c[0, 1] == 0
    # oracle spot checks at full genome size
    sample = [0, 41, 999]
    host = device_arena_to_host(arena, sample, length)
    for i, g in enumerate(sample):
        want = oracle.sketch_seq(arena_to_ascii(host, i), k, scaled)
        assert np.array_equal(flat[off[g] : off[g + 1]], want)
    block = [flat[off[g] : off[g + 1]] for g in range(0, 1000, 25)]
    assert np.array_equal(c[0:1000:25, 0:1000:25], oracle.pair_counts(block, threads=8))


@pytest.mark.parametrize("k", [33, 34, 47, 48, 49, 51, 63, 64])
def test_two_forms_of_the_long_kmer_kernel_agree(tools_engine, k, monkeypatch):
    """k above 32 has two independent kernels: 64 windows per thread (streams, column order, 128-bit compare on lane
    masks) and one window per thread-step (PA_KMER_LONG=plain).  Same sketches, Ns, lower case, records and all; both
    equal the oracle."""
    engine = tools_engine  # the -DPA_TOOLS build: the switch below exists there only
    from pyani_plus_amd.engine import pack_genomes

    rng = np.random.default_rng(3 * k)
    texts = [_random_fasta(rng, 3, 20_000, n_runs=5), _random_fasta(rng, 1, 70_000, n_runs=2), _random_fasta(rng, 2, 9_000, lower=True),
             b">short\n" + b"ACGT" * 12 + b"\n", b">exact\n" + rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=64 * 40).tobytes() + b"\n"]
    dev = engine.upload(pack_genomes(texts))
    for scaled in (1, 50):
        monkeypatch.delenv("PA_KMER_LONG", raising=False)
        wide = engine.sketch(dev, k, scaled).to_host()
        monkeypatch.setenv("PA_KMER_LONG", "plain")
        plain = engine.sketch(dev, k, scaled).to_host()
        for g, text in enumerate(texts):
            want, _total = oracle.sketch_fasta_text(text, k, scaled)
            assert np.array_equal(wide[g], want) and np.array_equal(plain[g], want), (k, scaled, g, len(wide[g]), len(plain[g]), len(want))


def test_full_size_long_kmer(engine):
    """k = 51 over the full 5 * 10^9 positions of BASELINE configs[1] (more threads than one grid dimension holds):
    sketch sizes ~ L / scaled everywhere, sampled genomes equal the oracle, the last genome included."""
    from pyani_plus_amd.synth import device_arena_to_host, synth_arena_torch

    n, length, k, scaled = 1000, 5_000_000, 51, 1000
    arena = synth_arena_torch(engine, n, length)
    sk = engine.sketch(arena, k, scaled)
    off = sk.off.cpu().numpy()
    sizes = np.diff(off)
    assert sizes.min() > 4500 and sizes.max() < 5500
    flat = sk.hashes[: sk.total].cpu().numpy().view(np.uint64)
    sample = [0, 500, 999]
    host = device_arena_to_host(arena, sample, length)
    for i, g in enumerate(sample):
        assert np.array_equal(flat[off[g] : off[g + 1]], oracle.sketch_seq(arena_to_ascii(host, i), k, scaled))


def test_mixed_length_set_config5_style(engine):
    """BASELINE configs[4] shape at reduced count: log-uniform lengths 100 kb - 10 Mb, cost-balanced shards."""
    from pyani_plus_amd.distributed import shard_bounds_by_cost
    from pyani_plus_amd.engine import pack_genomes

    rng = np.random.default_rng(2026)
    n = 40
    lengths = np.exp(rng.uniform(np.log(1e5), np.log(1e7), size=n)).astype(int)
    roots = [rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=int(lengths.max())) for _ in range(3)]
    seqs = []
    for g, length in enumerate(lengths):
        seq = roots[g % 3][:length].copy()
        hit = rng.random(length) < (0.002, 0.02, 0.08)[(g // 3) % 3]
        seq[hit] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=int(hit.sum()))]
        seqs.append(seq.tobytes())
    arena = pack_genomes(seqs, fasta=False)
    sk = engine.sketch(engine.upload(arena), 31, 1000)
    got = sk.to_host()
    want = oracle.sketch_many(seqs, 31, 1000, threads=8, fast=True)
    for a, b in zip(got, want):
        assert np.array_equal(a, b)
    sizes = np.array([len(s) for s in got])
    assert sizes.min() >= 60 and sizes.max() > 20 * sizes.min()  # ragged sketches
    counts = engine.pair_counts(sk).cpu().numpy().view(np.uint32)
    assert np.array_equal(counts, oracle.pair_counts(want, threads=8))
    # cost-balanced contiguous column tiles reproduce the full matrix when stitched
    bounds = shard_bounds_by_cost(sizes, 4)
    stitched = np.concatenate([engine.pair_counts(sk, (0, n), b).cpu().numpy().view(np.uint32) for b in bounds if b[1] > b[0]], axis=1)
    assert np.array_equal(stitched, counts)


def test_degenerate_inputs_and_error_codes(engine):
    """Empty sets, empty genomes, unsupported k and bad ranges through the C ABI."""
    import ctypes as C

    from pyani_plus_amd import _capi
    from pyani_plus_amd.engine import DeviceSketches, pack_genomes

    # no genome has a usable window -> every sketch is empty, all counts 0, all ANI NULL
    arena = pack_genomes([b"", b"ACGT", b"NNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNN"], fasta=False)
    sk = engine.sketch(engine.upload(arena), 31, 1)
    assert sk.total == 0 and [len(x) for x in sk.to_host()] == [0, 0, 0]
    counts = engine.pair_counts(sk)
    assert int(counts.abs().sum().item()) == 0
    ident, cov = engine.ani(counts, sk, 31)
    assert bool(ident.isnan().all()) and bool(cov.isnan().all())
    # zero genomes
    t = engine.torch
    empty = DeviceSketches(t.zeros(1, dtype=t.int64, device=engine.device), t.zeros(1, dtype=t.int64, device=engine.device), 0, 0)
    assert tuple(engine.pair_counts(empty).shape) == (0, 0)
    # unsupported k, bad ranges, bad algo -> error codes with messages, no crash
    one = pack_genomes([b"ACGT" * 100], fasta=False)
    with pytest.raises(_capi.HipBackendError, match=r"outside \[1,64\]"):
        engine.sketch(engine.upload(one), 65, 10)
    with pytest.raises(_capi.HipBackendError, match=r"outside \[1,64\]"):
        engine.sketch(engine.upload(one), 0, 10)
    good = engine.sketch(engine.upload(one), 31, 10)
    with pytest.raises(_capi.HipBackendError, match="ranges"):
        engine.pair_counts(good, (0, 2), (0, 1))
    with pytest.raises(_capi.HipBackendError, match="unknown algo"):
        engine.pair_counts(good, algo=7)
    lib = _capi.load_library()
    assert lib.pa_ctx_sync(None) == -1 and b"null context" in lib.pa_last_error()
    # arena size not a multiple of 64 is rejected before any launch
    total = C.c_uint64(0)
    gs = (C.c_uint64 * 2)(0, 100)
    st = lib.pa_sketch(engine.ctx, one_ptr := engine.upload(one).packed.data_ptr(), one_ptr, None, 100, gs, 1, 31, 1, None, 0, good.off.data_ptr(), C.byref(total))
    assert st == -1 and b"multiple of 64" in lib.pa_last_error()


def test_streamed_sketch_equals_resident_sketch(engine):
    """pa_sketch_streamed: run-length mask + chunked upload behind the hash kernel give the same device arena
    and the same sketches as upload + pa_sketch, on the LDS-sort path and on the fallbacks."""
    from pyani_plus_amd.engine import pack_genomes

    rng = np.random.default_rng(11)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    texts = []
    for g in range(9):
        seq = bytearray(rng.choice(acgt, size=int(rng.integers(1000, 400_000))).tobytes())
        for _ in range(int(rng.integers(0, 4))):  # N runs of various lengths, some crossing word boundaries
            a, n_len = int(rng.integers(0, len(seq) - 500)), int(rng.integers(1, 400))
            seq[a : a + n_len] = b"N" * n_len
        texts.append(b">g%d\n" % g + bytes(seq[: len(seq) // 2]) + b"\n>second record\n" + bytes(seq[len(seq) // 2 :]) + b"\n")
    texts.append(b">empty\n")
    ladder = synth_arena_numpy(40, [5_000 + 37_000 * i for i in range(40)], n_species=5)
    # FASTA with N runs on the LDS-sort path; a length ladder; scaled=1 (regions too long for LDS: plain upload)
    for arena, scaled in ((pack_genomes(texts), 200), (ladder, 1000), (pack_genomes(texts), 1)):
        dev = engine.upload(arena)
        want = engine.sketch(dev, 31, scaled)
        pinned = engine.pin_arena(arena)
        dev2, got = engine.sketch_streamed(pinned, 31, scaled)
        assert got.total == want.total
        assert engine.torch.equal(got.off, want.off)
        assert engine.torch.equal(got.hashes[: got.total], want.hashes[: want.total])
        assert engine.torch.equal(dev2.mask[: arena.mask.size], dev.mask)
        assert engine.torch.equal(dev2.packed[: arena.packed.size], dev.packed)


@pytest.mark.parametrize("k", [31, 51, 64])
def test_streamed_sketch_windows_across_chunk_boundaries(tools_engine, k, monkeypatch):
    """The hash kernel of chunk c looks back into chunk c - 1 (up to k - 1 positions, 63 for the long-k form): with
    chunks of 64 arena blocks every 4 096th position is such a boundary."""
    engine = tools_engine  # the -DPA_TOOLS build: the switch below exists there only
    from pyani_plus_amd.engine import pack_genomes

    rng = np.random.default_rng(k)
    texts = [_random_fasta(rng, 2, 30_000, n_runs=3), _random_fasta(rng, 1, 50_000), _random_fasta(rng, 3, 9_000, lower=True)]
    arena = pack_genomes(texts)
    want = engine.sketch(engine.upload(arena), k, 20).to_host()
    monkeypatch.setenv("PA_STREAM_CHUNK_BLOCKS", "64")
    _dev, got = engine.sketch_streamed(engine.pin_arena(arena), k, 20)
    got = got.to_host()
    for g, text in enumerate(texts):
        oracle_mins, _total = oracle.sketch_fasta_text(text, k, 20)
        assert np.array_equal(got[g], want[g]) and np.array_equal(got[g], oracle_mins), (k, g)
