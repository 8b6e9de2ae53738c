"""GPU: the fastANI-style fragment-ANI path against its oracle (exact integers) and, through the
oracle, against the reference's fastANI fixtures (every row exactly as fastANI wrote it, tests/test_fragani_oracle.py)."""

from __future__ import annotations

import math

import numpy as np
import pytest

import oracle
from tests.helpers import GOLDEN, read_fasta_bytes
from tests.test_fragani_oracle import bacterial_row_bounds, contigs_of, fixture_rows, printed

pytestmark = pytest.mark.gpu
K, FRAG = 16, 3000


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


def _random_genomes(seed: int):
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    root = rng.choice(acgt, size=40_000)
    texts, contig_lists = [], []
    for g, (rate, cuts) in enumerate([(0.0, [40_000]), (0.03, [9_500, 12_345, 18_155]), (0.12, [40_000]), (0.0, [7_000, 3_100, 2_999, 26_901])]):
        seq = root.copy()
        hit = rng.random(seq.size) < rate
        seq[hit] = acgt[rng.integers(0, 4, size=int(hit.sum()))]
        if g == 2:
            seq[5_000:5_040] = ord("N")  # a run of N inside a fragment
        if g == 3:  # single unknown residues: inside a contig, as a contig's first and as a contig's last residue
            seq[[1_234, 7_000, 7_000 + 3_100 - 1, 13_099 + 777, 13_099 + 778, 30_000]] = ord("N")
        contigs, pos = [], 0
        for n in cuts:
            contigs.append(seq[pos : pos + n].tobytes())
            pos += n
        contig_lists.append(contigs)
        texts.append(b"".join(b">c%d\n" % i + c + b"\n" for i, c in enumerate(contigs)))
    unrelated = rng.choice(acgt, size=20_000).tobytes()
    contig_lists.append([unrelated])
    texts.append(b">u\n" + unrelated + b"\n")
    return texts, contig_lists


def mean_f(ident_sum, matched) -> float:
    """the pair's ANI as fastANI computes it: float sum / float count (pyani_plus_amd.methods.fastani_hip.fastani_mean)"""
    from pyani_plus_amd.methods.fastani_hip import fastani_mean

    return float(fastani_mean(ident_sum, matched))


def test_parameters_match_oracle(engine):
    from pyani_plus_amd import _capi

    lib = _capi.load_library()
    for k, frag in ((16, 3000), (15, 2000), (16, 1000), (14, 3000)):
        assert lib.pa_fragani_window(k, frag) == oracle.fragani_window_size(k, frag)
    mh = np.zeros(513, dtype=np.uint32)
    ms = np.zeros(513, dtype=np.uint32)
    assert lib.pa_fragani_tables(16, 512, mh.ctypes.data, ms.ctypes.data) == 0
    omh, oms = oracle.fragani_tables(16, 512)
    assert np.array_equal(mh, omh.astype(np.uint32)) and np.array_equal(ms, oms.astype(np.uint32))
    for shared, s in ((0, 10), (3, 200), (150, 214), (214, 214)):
        assert lib.pa_fragani_identity(shared, s, 16) == oracle.fragani_identity(shared, s, 16)


@pytest.mark.parametrize("k,w", [(16, 23), (15, 19), (16, 5), (12, 64), (13, 24), (8, 31), (11, 9)])
def test_minimizers_equal_oracle(engine, k, w):
    from pyani_plus_amd.engine import pack_genomes

    texts, contig_lists = _random_genomes(k * 100 + w)
    texts.append(read_fasta_bytes(GOLDEN / "viral_example" / "OP073605.fasta"))
    contig_lists.append(contigs_of(GOLDEN / "viral_example" / "OP073605.fasta"))
    arena = pack_genomes(texts)
    h, wp, ct = engine.fragani_sketch(engine.upload(arena), arena.contig_start, arena.contig_len, arena.contig_genome, k, w)
    ci = 0
    for contigs in contig_lists:
        for contig in contigs:
            want_h, want_p = oracle.fragani_minimizers(contig, k, w)
            sel = ct == ci
            assert np.array_equal(h[sel], want_h), f"contig {ci}: {sel.sum()} vs {len(want_h)} minimizers"
            assert np.array_equal(wp[sel].astype(np.int32), want_p)
            ci += 1
    assert ci == len(arena.contig_start)


def test_minimizer_run_is_repeated_when_the_estimate_is_too_small(tools_engine, monkeypatch):
    """The single-pass minimizer kernel writes into arrays sized from the expected density 2 / (w + 1); low-complexity
    sequence (every window of a homopolymer records a new, rightmost, position) needs more, and the run is repeated
    with the exact size.  Forced here by giving the first run room for 100 minimizers."""
    engine = tools_engine  # the -DPA_TOOLS build: the switch below exists there only
    from pyani_plus_amd.engine import pack_genomes

    rng = np.random.default_rng(3)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    contigs = [b"A" * 5_000 + rng.choice(acgt, size=6_000).tobytes() + b"AC" * 1_500, rng.choice(acgt, size=30_000).tobytes()]
    arena = pack_genomes([b"".join(b">c%d\n" % i + c + b"\n" for i, c in enumerate(contigs))])
    for room in ("100", None):
        if room is None:
            monkeypatch.delenv("PA_FRAGANI_MINIMIZER_ROOM", raising=False)
        else:
            monkeypatch.setenv("PA_FRAGANI_MINIMIZER_ROOM", room)
        h, wp, ct = engine.fragani_sketch(engine.upload(arena), arena.contig_start, arena.contig_len, arena.contig_genome, 16, 24)
        for ci, contig in enumerate(contigs):
            want_h, want_p = oracle.fragani_minimizers(contig, 16, 24)
            assert np.array_equal(h[ct == ci], want_h) and np.array_equal(wp[ct == ci].astype(np.int32), want_p)
        assert len(h) > 5_000  # the homopolymer alone gives one minimizer per window


def test_minimizers_with_the_tiles_drawn_per_xcd_and_from_one_counter(tools_engine, monkeypatch):
    """minimizer_kernel hands its tiles out through one ticket counter per XCD and falls back to a single counter should
    a wait of its chained scan ever run out; which counter a tile came from decides nothing: both forms (the second forced
    through the tools build) give the oracle's minimizers, in arena order, over a few hundred tiles of many short contigs,
    runs of N and a window size that takes the plain path (w = 5) as well as fastANI's 24."""
    engine = tools_engine
    from pyani_plus_amd.engine import pack_genomes

    rng = np.random.default_rng(17)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    contigs = []
    for i in range(40):
        seq = rng.choice(acgt, size=int(rng.integers(30, 40_000))).copy()
        if i % 3 == 0 and seq.size > 400:
            a = int(rng.integers(0, seq.size - 300))
            seq[a : a + int(rng.integers(1, 300))] = ord("N")
        contigs.append(seq.tobytes())
    arena = pack_genomes([b"".join(b">c%d\n" % i + c + b"\n" for i, c in enumerate(contigs[:25])), b"".join(b">d%d\n" % i + c + b"\n" for i, c in enumerate(contigs[25:]))])
    dev = engine.upload(arena)
    for k, w in ((16, 24), (12, 5), (16, 64)):
        got = {}
        for one in ("0", "1"):
            monkeypatch.setenv("PA_FRAGANI_ONE_TICKET", one)
            got[one] = engine.fragani_sketch(dev, arena.contig_start, arena.contig_len, arena.contig_genome, k, w)
        monkeypatch.delenv("PA_FRAGANI_ONE_TICKET")
        # the fallback itself: the run with per-XCD counters is taken to have timed out and is repeated with the single counter
        monkeypatch.setenv("PA_FRAGANI_TICKET_TIMEOUT", "1")
        got["timeout"] = engine.fragani_sketch(dev, arena.contig_start, arena.contig_len, arena.contig_genome, k, w)
        monkeypatch.delenv("PA_FRAGANI_TICKET_TIMEOUT")
        for a, b, c in zip(got["0"], got["1"], got["timeout"]):
            assert np.array_equal(a, b) and np.array_equal(a, c)
        h, wp, ct = got["0"]
        for ci, contig in enumerate(contigs):
            want_h, want_p = oracle.fragani_minimizers(contig, k, w)
            assert np.array_equal(h[ct == ci], want_h) and np.array_equal(wp[ct == ci].astype(np.int32), want_p), (k, w, ci)


def _check_against_oracle(engine, texts, contig_lists, frag=FRAG, k=K):
    from pyani_plus_amd.engine import pack_genomes

    arena = pack_genomes(texts)
    total, matched, ident_sum = engine.fragani(engine.upload(arena), arena.contig_start, arena.contig_len, arena.contig_genome, k, frag)
    n = len(texts)
    for q in range(n):
        for r in range(n):
            ani, m, t = oracle.fragani_pair(contig_lists[q], contig_lists[r], k, frag, 0.0)
            assert total[q] == t
            assert matched[q, r] == m, (q, r, matched[q, r], m)
            if m:
                assert mean_f(ident_sum[q, r], m) == ani  # the same float sum in the same order, the same float division
            else:
                assert math.isnan(ani) and ident_sum[q, r] == 0.0
    return total, matched, ident_sum


def test_pairs_equal_oracle_on_random_genomes(engine):
    texts, contig_lists = _random_genomes(7)
    total, matched, _ = _check_against_oracle(engine, texts, contig_lists)
    assert matched[0, 0] == total[0] and matched[0, 4] == 0 and matched[4, 0] == 0
    _check_against_oracle(engine, texts, contig_lists, frag=1000, k=15)
    # fragments of 5 000: window 40, stretches of up to 384 minimizers -- twelve words per row of the mapping kernel's bit tables
    _check_against_oracle(engine, texts, contig_lists, frag=5000, k=16)
    # 9-mers: unrelated genomes share thousands of minimizers by chance, the same hash several times in a window
    _check_against_oracle(engine, texts, contig_lists, frag=600, k=9)


def test_fragments_that_are_mostly_n(engine):
    """Fragments with a handful of usable k-mers: sketches of a few minimizers (fewer than one coarse step of the mapping
    kernel's rank tables), windows with hardly anything in them.  Same integers as the oracle."""
    rng = np.random.default_rng(77)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    root = rng.choice(acgt, size=30_000)
    genomes = []
    for rate, gaps in [(0.0, [(3_050, 5_950), (9_100, 11_990)]), (0.02, [(3_200, 5_800)]), (0.0, []), (0.05, [(15_010, 17_995), (18_020, 20_900)])]:
        seq = root.copy()
        hit = rng.random(seq.size) < rate
        seq[hit] = acgt[rng.integers(0, 4, size=int(hit.sum()))]
        for a, b in gaps:
            seq[a:b] = ord("N")
        genomes.append(seq.tobytes())
    texts = [b">g%d\n" % i + g + b"\n" for i, g in enumerate(genomes)]
    total, matched, _ = _check_against_oracle(engine, texts, [[g] for g in genomes])
    assert total[0] == 10 and matched[2, 0] > 0
    _check_against_oracle(engine, texts, [[g] for g in genomes], frag=1000, k=15)


def test_reference_of_hundreds_of_contigs(engine):
    """A draft assembly of 320 contigs: hits on contigs past the 255th of their genome do not fit the mapping kernel's
    32-bit sort keys and take the 64-bit ones; candidates on many contigs per segment.  Same integers as the oracle."""
    rng = np.random.default_rng(5)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    root = rng.choice(acgt, size=320 * 1_150)
    whole = root.tobytes()
    pieces = [root[i * 1_150 : (i + 1) * 1_150].tobytes() for i in range(320)]
    mutated = root.copy()
    hit = rng.random(mutated.size) < 0.02
    mutated[hit] = acgt[rng.integers(0, 4, size=int(hit.sum()))]
    contig_lists = [[whole], pieces, [mutated.tobytes()]]
    texts = [b"".join(b">c%d\n" % i + c + b"\n" for i, c in enumerate(contigs)) for contigs in contig_lists]
    total, matched, _ = _check_against_oracle(engine, texts, contig_lists, frag=1000, k=15)
    assert matched[0, 1] > 300 and matched[2, 1] > 200  # mappings onto contigs of every index
    assert total[1] == 320  # one fragment per contig of the draft


def test_draft_assemblies_of_ten_kilobase_contigs(engine):
    """Genomes handed over as contigs of 10 000 residues at fastANI's defaults (three fragments and a remainder per contig,
    every contig's last window short of its end): what `tools/bench_fragani.py 1000 0 interleaved 1000 1000 500` times.  A
    genome keeps fewer of its own fragments than a one-contig genome does -- and exactly as many as the oracle says."""
    rng = np.random.default_rng(21)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    root = rng.choice(acgt, size=24 * 10_000)
    contig_lists = []
    for rate in (0.0, 0.01, 0.06):
        seq = root.copy()
        hit = rng.random(seq.size) < rate
        seq[hit] = acgt[rng.integers(0, 4, size=int(hit.sum()))]
        contig_lists.append([seq[i * 10_000 : (i + 1) * 10_000].tobytes() for i in range(24)])
    contig_lists.append([root.tobytes()])  # the same sequence in one piece
    texts = [b"".join(b">c%d\n" % i + c + b"\n" for i, c in enumerate(contigs)) for contigs in contig_lists]
    total, matched, _ = _check_against_oracle(engine, texts, contig_lists)
    assert total[0] == 24 * 3 and total[3] == 80
    assert matched[0, 0] <= total[0] and matched[3, 3] >= 79


def test_batches_are_halved_when_the_seed_hits_outgrow_their_indices(tools_engine, monkeypatch):
    """A batch of query genomes whose seed hits pass 2^31 is halved and started again; forced here with a limit of a few
    thousand hits, down to one query genome per batch.  Same integers as the one-batch run and as the oracle."""
    engine = tools_engine  # the -DPA_TOOLS build: the switch below exists there only
    texts, contig_lists = _random_genomes(31)
    one_batch = _check_against_oracle(engine, texts, contig_lists)
    monkeypatch.setenv("PA_FRAGANI_BATCH_HITS", "3000")
    halved = _check_against_oracle(engine, texts, contig_lists)
    for a, b in zip(one_batch, halved):
        assert np.array_equal(a, b)


def test_repeat_families_take_the_long_segment_paths(engine):
    """Tandem repeats give (fragment, genome) segments of thousands of seed hits: more than the mapping wave
    stages in LDS (512: sorted by frag_sort_kernel, read in place) and, for the 60-copy genome, more than one
    LDS sort takes (8 192: the whole batch is radix-sorted instead)."""
    rng = np.random.default_rng(21)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    unit = rng.choice(acgt, size=3_000)
    flank = rng.choice(acgt, size=9_000)

    def with_copies(copies: int, rate: float) -> bytes:
        parts = [flank[:4_500]]
        for _ in range(copies):
            u = unit.copy()
            hit = rng.random(u.size) < rate
            u[hit] = acgt[rng.integers(0, 4, size=int(hit.sum()))]
            parts.append(u)
        parts.append(flank[4_500:])
        return np.concatenate(parts).tobytes()

    genomes = [with_copies(8, 0.0), with_copies(8, 0.02), with_copies(60, 0.001)]
    texts = [b">g%d\n" % i + g + b"\n" for i, g in enumerate(genomes)]
    total, matched, _ = _check_against_oracle(engine, texts, [[g] for g in genomes])
    # copies of one unit compete for the same reference bins, so not every fragment is kept even against itself
    assert 0 < matched[2, 2] <= total[2] and matched[0, 1] > 0
    # without the 60-copy genome the batch stays on the bucketed path and its 2 000-hit segments go through
    # frag_sort_kernel
    _check_against_oracle(engine, texts[:2], [[g] for g in genomes[:2]])


def test_whole_batch_sort_with_unlisted_pairs(tools_engine, monkeypatch):
    """The whole-batch radix sort of a repeat family (more hits in one segment than an LDS sort takes) must leave every
    (fragment, genome) slice in place although the bucketing does not write the hits of pairs nobody maps: chance hits of
    an unrelated genome and everything outside the reference range.  A 60-copy repeat genome, a relative, an unrelated
    genome sharing a short stretch (a few seed hits per fragment: below what an L1 run needs) and a reference range;
    once with the real limit (8 192 hits) and once with the limit lowered so that the 8-copy segments take the path too."""
    engine = tools_engine  # the -DPA_TOOLS build: the switch below exists there only
    from pyani_plus_amd.engine import pack_genomes

    rng = np.random.default_rng(212)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    unit = rng.choice(acgt, size=3_000)
    flank = rng.choice(acgt, size=9_000)

    def with_copies(copies: int, rate: float) -> bytes:
        parts = [flank[:4_500]]
        for _ in range(copies):
            u = unit.copy()
            hit = rng.random(u.size) < rate
            u[hit] = acgt[rng.integers(0, 4, size=int(hit.sum()))]
            parts.append(u)
        parts.append(flank[4_500:])
        return np.concatenate(parts).tobytes()

    other = rng.choice(acgt, size=30_000)
    other[10_000:10_060] = unit[500:560]  # a few shared minimizers with every copy of the unit: seed hits, never a run
    genomes = [other.tobytes(), with_copies(60, 0.001), with_copies(8, 0.01), rng.choice(acgt, size=12_000).tobytes()]
    texts = [b">g%d\n" % i + g + b"\n" for i, g in enumerate(genomes)]
    arena = pack_genomes(texts)
    n = len(genomes)
    want = {(q, r): oracle.fragani_pair([genomes[q]], [genomes[r]], K, FRAG, 0.0) for q in range(n) for r in range(n)}
    for sort_max in (None, "600"):
        if sort_max is None:
            monkeypatch.delenv("PA_FRAGANI_SORT_MAX", raising=False)
        else:
            monkeypatch.setenv("PA_FRAGANI_SORT_MAX", sort_max)
        for ref_range in (None, (1, 3), (1, 2), (2, 4)):
            total, matched, ident_sum = engine.fragani(engine.upload(arena), arena.contig_start, arena.contig_len, arena.contig_genome,
                                                       K, FRAG, ref_range=ref_range)
            r0, r1 = ref_range or (0, n)
            for q in range(n):
                for r in range(n):
                    ani, m, t = want[q, r]
                    assert total[q] == t
                    if not r0 <= r < r1:
                        assert matched[q, r] == 0 and ident_sum[q, r] == 0.0
                        continue
                    assert matched[q, r] == m, (sort_max, ref_range, q, r, matched[q, r], m)
                    if m:
                        assert mean_f(ident_sum[q, r], m) == ani
    assert want[1, 1][1] > 0 and want[2, 1][1] > 0 and want[0, 1][1] == 0


def test_two_copy_repeats_take_the_wide_register_sort(engine):
    """Two tandem copies give segments of 257 .. 512 seed hits: staged in LDS by the launch for long segments, ordered
    eight keys per lane in registers on the way (the widest form of the bitonic network); duplicates of every hash inside
    the windows.  Same integers as the oracle."""
    rng = np.random.default_rng(33)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    unit = rng.choice(acgt, size=3_000)
    flank = rng.choice(acgt, size=6_000)

    def with_copies(copies: int, rate: float) -> bytes:
        parts = [flank[:3_000]]
        for _ in range(copies):
            u = unit.copy()
            hit = rng.random(u.size) < rate
            u[hit] = acgt[rng.integers(0, 4, size=int(hit.sum()))]
            parts.append(u)
        parts.append(flank[3_000:])
        return np.concatenate(parts).tobytes()

    genomes = [with_copies(2, 0.0), with_copies(2, 0.01), with_copies(3, 0.005), with_copies(1, 0.0)]
    texts = [b">g%d\n" % i + g + b"\n" for i, g in enumerate(genomes)]
    total, matched, _ = _check_against_oracle(engine, texts, [[g] for g in genomes])
    assert matched[0, 1] > 0 and matched[3, 2] > 0


def test_viral_fixture_rows(engine):
    files = sorted((GOLDEN / "viral_example").glob("*.f*"))
    texts = [read_fasta_bytes(p) for p in files]
    contig_lists = [contigs_of(p) for p in files]
    total, matched, ident_sum = _check_against_oracle(engine, texts, contig_lists)
    names = [p.name for p in files]
    for q, r, ani, m, t in fixture_rows("viral_example"):
        qi, ri = names.index(q), names.index(r)
        assert total[qi] == t and int(matched[qi, ri]) == m
        assert printed(mean_f(ident_sum[qi, ri], matched[qi, ri])) == ani, (q, r)  # as fastANI prints it: six significant digits


def test_bacterial_fixture_rows(engine):
    """All 16 bacterial fastANI rows on the GPU, each exactly as fastANI wrote it (the CPU oracle is only asked for two pairs here)."""
    from pyani_plus_amd.engine import load_fasta_files

    files = sorted((GOLDEN / "bacterial_example").glob("*.gz"))
    infos, arena = load_fasta_files(files)
    total, matched, ident_sum = engine.fragani(engine.upload(arena), arena.contig_start, arena.contig_len, arena.contig_genome, K, FRAG)
    names = [p.name for p in files]
    for q, r, ani, m, t in fixture_rows("bacterial_example"):
        qi, ri = names.index(q), names.index(r)
        bacterial_row_bounds(q, r, ani, m, t, mean_f(ident_sum[qi, ri], matched[qi, ri]), int(matched[qi, ri]), int(total[qi]))
    for qi, ri in ((1, 0), (0, 2)):
        ani, m, t = oracle.fragani_pair(contigs_of(files[qi]), contigs_of(files[ri]), K, FRAG, 0.0)
        assert matched[qi, ri] == m and mean_f(ident_sum[qi, ri], m) == ani


def test_plugin_column_matches_reference_matrices(engine, tmp_path):
    """compute_fastani_hip -> JSON column vs the reference's fastANI matrices for the viral set
    (identity to the digits the matrix file holds, aln_length / sim_errors / cov_query from kept fragments)."""
    import json
    import logging

    from pyani_plus_amd import rundb
    from pyani_plus_amd.methods import fastani_hip
    from tests.helpers import load_matrix_tsv, md5_hex

    files = sorted((GOLDEN / "viral_example").glob("*.f*"))
    hash_to_filename = {md5_hex(read_fasta_bytes(p)): p.name for p in files}
    tool = fastani_hip.get_fastani_hip()
    cfg = rundb.Configuration(5, fastani_hip.METHOD, tool.exe_path.stem, tool.version, fragsize=3000, kmersize=16, minmatch=0.2)
    run = rundb.Run(1, cfg, str(GOLDEN / "viral_example"), [], "Testing")

    class _S:
        def commit(self):
            pass

    out = tmp_path / "col.json"
    lengths = {h: 1 for h in hash_to_filename}
    assert fastani_hip.compute_fastani_hip(logging.getLogger("t"), tmp_path, _S(), run, out, GOLDEN / "viral_example", hash_to_filename, {}, lengths, "", engine=engine) == 0
    data = json.loads(out.read_text())
    assert data["configuration"]["method"] == "fastANI-hip" and data["configuration"]["fragsize"] == 3000
    rows = {(e["query_hash"], e["subject_hash"]): e for e in data["comparisons"]}
    assert len(rows) == 9
    stem_of = {h: name.split(".")[0] for h, name in hash_to_filename.items()}
    # the viral matrices are reproduced as fastANI printed them: identity to its six significant digits, coverage exactly
    for fname, key, tol in (("fastANI_identity.tsv", "identity", 1e-12), ("fastANI_coverage.tsv", "cov_query", 1e-15)):
        labels, want = load_matrix_tsv(GOLDEN / "viral_example" / "matrices" / fname)
        for (q, s), e in rows.items():
            w = want[labels.index(stem_of[q]), labels.index(stem_of[s])]
            assert abs(e[key] - w) <= tol, (key, stem_of[q], stem_of[s], e[key], w)
    for e in rows.values():
        assert e["aln_length"] == round(3000 * e["cov_query"] * (e["aln_length"] + 3000 * e["sim_errors"]) / 3000) and e["sim_errors"] >= 0
    # one subject column and a minFraction nothing reaches
    subject = sorted(hash_to_filename)[0]
    assert fastani_hip.compute_fastani_hip(logging.getLogger("t"), tmp_path, _S(), run, out, GOLDEN / "viral_example", hash_to_filename, {}, lengths, subject, engine=engine) == 0
    col = json.loads(out.read_text())["comparisons"]
    assert [(e["query_hash"], e["subject_hash"]) for e in col] == [(q, subject) for q in sorted(hash_to_filename)]
    assert all(e["identity"] == rows[(e["query_hash"], subject)]["identity"] for e in col)
    # a size-asymmetric pair: minFraction is taken of the SHORTER genome (fastANI's rule), not of the query's fragments
    rng = np.random.default_rng(21)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    short = rng.choice(acgt, size=45_000).tobytes()
    long_ = short + rng.choice(acgt, size=255_000).tobytes()
    asym = tmp_path / "asym"
    asym.mkdir()
    (asym / "short.fasta").write_bytes(b">short\n" + short + b"\n")
    (asym / "long.fasta").write_bytes(b">long\n" + long_ + b"\n")
    h2f = {md5_hex((asym / n).read_bytes()): n for n in ("short.fasta", "long.fasta")}
    f2h = {n: h for h, n in h2f.items()}
    run2 = rundb.Run(2, cfg, str(asym), [], "Testing")
    assert fastani_hip.compute_fastani_hip(logging.getLogger("t"), tmp_path, _S(), run2, out, asym, h2f, {}, {h: 1 for h in h2f}, "", engine=engine) == 0
    got = {(e["query_hash"], e["subject_hash"]): e for e in json.loads(out.read_text())["comparisons"]}
    lq = got[(f2h["long.fasta"], f2h["short.fasta"])]
    assert lq["identity"] is not None and lq["identity"] > 0.999 and abs(lq["cov_query"] - 0.15) < 0.011 and lq["sim_errors"] >= 84
    assert got[(f2h["short.fasta"], f2h["long.fasta"])]["cov_query"] == 1.0
    cfg.minmatch = 1.5
    assert fastani_hip.compute_fastani_hip(logging.getLogger("t"), tmp_path, _S(), run, out, GOLDEN / "viral_example", hash_to_filename, {}, lengths, subject, engine=engine) == 0
    assert all(e["identity"] is None and e["cov_query"] is None for e in json.loads(out.read_text())["comparisons"])


def test_reuse_index_when_no_contig_holds_a_fragment(engine):
    """Every contig shorter than a fragment: nothing to map, and a follow-up call that asks for the previous call's index
    (what the column worker does for every query batch after the first) gets zeros as well instead of an error."""
    from pyani_plus_amd.engine import pack_genomes

    rng = np.random.default_rng(8)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    texts = [b">g%d\n" % i + rng.choice(acgt, size=2_000 + 100 * i).tobytes() + b"\n" for i in range(3)]
    arena = pack_genomes(texts)
    dev = engine.upload(arena)
    first = engine.fragani(dev, arena.contig_start, arena.contig_len, arena.contig_genome, K, FRAG, query_range=(0, 2))
    again = engine.fragani(dev, arena.contig_start, arena.contig_len, arena.contig_genome, K, FRAG, query_range=(2, 3), reuse_index=True)
    for total, matched, ident_sum in (first, again):
        assert not total.any() and not matched.any() and not ident_sum.any()


def test_columns_only_results_equal_the_square_ones(engine):
    """``PA_FRAGANI_COLUMNS_ONLY``: a range of subject columns comes back as [n, r1 - r0] arrays (what a one-column worker
    keeps in host memory), equal to those columns of the n x n result; query batches fill their own rows."""
    from pyani_plus_amd.engine import pack_genomes

    texts, _contigs = _random_genomes(19)
    arena = pack_genomes(texts)
    dev = engine.upload(arena)
    n = len(texts)
    total, matched, ident_sum = engine.fragani(dev, arena.contig_start, arena.contig_len, arena.contig_genome, K, FRAG)
    for r0, r1 in ((0, 1), (1, 4), (4, 5), (0, n)):
        out = (np.zeros(n, dtype=np.uint32), np.full((n, r1 - r0), 7, dtype=np.uint32), np.full((n, r1 - r0), 7.0))
        for b, (q0, q1) in enumerate(((0, 2), (2, n))):
            t, m, s = engine.fragani(dev, arena.contig_start, arena.contig_len, arena.contig_genome, K, FRAG, ref_range=(r0, r1),
                                     query_range=(q0, q1), reuse_index=b > 0, out=out, columns_only=True)
        assert np.array_equal(t, total) and np.array_equal(m, matched[:, r0:r1]) and np.array_equal(s, ident_sum[:, r0:r1])


def test_contigs_that_end_right_after_their_last_fragment(engine):
    """The slide ends when the window's end reaches the first minimizer at or past rangeEnd + fragLen -- or the contig's
    end: a fragment that ends within a few residues of its contig's end cannot be mapped at its own position (the
    reference's MIBY01000011 pin).  Contigs of f fragments plus 0 .. 120 residues, against themselves and against
    mutated copies whose contigs end elsewhere: same integers as the oracle."""
    rng = np.random.default_rng(404)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    root = rng.choice(acgt, size=200_000)
    tails = [0, 1, 2, 5, 17, 23, 38, 39, 40, 41, 64, 120]
    contigs_a, contigs_b, pos = [], [], 0
    for i, r in enumerate(tails):
        n = 3_000 * (1 + i % 3) + r
        seq = root[pos : pos + n + 200]
        contigs_a.append(seq[:n].tobytes())
        mutated = seq.copy()
        hit = rng.random(mutated.size) < 0.01
        mutated[hit] = acgt[rng.integers(0, 4, size=int(hit.sum()))]
        contigs_b.append(mutated[: n + (7 * i) % 200].tobytes())  # the copy's contig ends somewhere else
        pos += n + 200
    contig_lists = [contigs_a, contigs_b, [b"".join(contigs_a)]]
    texts = [b"".join(b">c%d\n" % i + c + b"\n" for i, c in enumerate(cs)) for cs in contig_lists]
    total, matched, _ = _check_against_oracle(engine, texts, contig_lists)
    # nearly every fragment maps onto its own genome; the last fragment of a contig that ends with it (tails 0 and 1 here)
    # may find no window at all before the slide ends
    assert total[0] - 3 <= matched[0, 0] <= total[0]
    _check_against_oracle(engine, texts, contig_lists, frag=1000, k=15)


def test_a_long_run_of_n_inside_a_contig(engine):
    """70 kb of N in the middle of a contig: the window ids of neighbouring minimizers are more than 65 535 apart (the
    mapping kernel keeps them as 16-bit offsets inside a stretch), windows near the gap hold next to nothing."""
    rng = np.random.default_rng(505)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    left, right = rng.choice(acgt, size=12_000), rng.choice(acgt, size=15_000)
    gap = np.full(70_000, ord("N"), dtype=np.uint8)
    a = np.concatenate([left, gap, right])
    b = a.copy()
    hit = (rng.random(b.size) < 0.02) & (b != ord("N"))
    b[hit] = acgt[rng.integers(0, 4, size=int(hit.sum()))]
    c = np.concatenate([left, right])  # the same sequence without the gap
    genomes = [a.tobytes(), b.tobytes(), c.tobytes()]
    texts = [b">g%d\n" % i + g + b"\n" for i, g in enumerate(genomes)]
    total, matched, _ = _check_against_oracle(engine, texts, [[g] for g in genomes])
    assert matched[2, 0] > 0 and matched[0, 2] > 0


def _expected_seed_hits(genomes: list[bytes], k: int, frag: int, cut: bool) -> int:
    """Seed hits of an all-vs-all run, counted independently of the library: per fragment the distinct minimizer hashes of its
    slice of the genome's minimizers, per reference genome the occurrences of each -- none where Mashmap's frequency cut
    applies (occurrences >= the genome's threshold; the oracle's ref_index_build restated in numpy)."""
    w = oracle.fragani_window_size(k, frag)
    cw = frag - (w - 1) - (k - 1)
    minis = [oracle.fragani_minimizers(g, k, w) for g in genomes]
    tables = []
    for h, _p in minis:
        u, cnt = np.unique(h, return_counts=True)
        thr = np.iinfo(np.int64).max
        if cut and len(u):
            to_ignore = int(np.float32(len(u)) * np.float32(0.001) / np.float32(100))
            bars, sizes = np.unique(cnt, return_counts=True)
            total = 0
            for c, n in zip(bars[::-1].tolist(), sizes[::-1].tolist()):
                total += n
                if total < to_ignore:
                    thr = c
                else:
                    if total == to_ignore:
                        thr = c
                    break
        tables.append(dict(zip(u[cnt < thr].tolist(), cnt[cnt < thr].tolist())))
    hits = 0
    for (h, p), g in zip(minis, genomes):
        for f in range(len(g) // frag):
            start = f * frag
            b, e = int(np.searchsorted(p, start)), int(np.searchsorted(p, start + cw))
            b0 = b - 1 if b > 0 and not (b < len(p) and p[b] == start) else b
            for x in np.unique(h[b0:e]).tolist():
                hits += sum(t.get(x, 0) for t in tables)
    return hits


def test_frequency_cut_of_the_seeds(tools_engine, monkeypatch, capfd):
    """Mashmap's cut of the most frequent reference minimizers from the seed look-up (fastANI logs "ignore minimizers
    occurring >= N times during lookup"): per reference genome, as many bars of the histogram of occurrence counts, from the
    top, as stay within 0.001 % of its distinct minimizers.  A 1.4 Mb genome holds an (ACC)n array whose one minimizer occurs
    ~300 times: above the threshold, it gives no seed hits -- the number of seed hits the library reports equals the count
    made here from the oracle's minimizers, with the cut and (PA_FRAGANI_NO_FREQ_CUT=1) without; the results equal the
    oracle's, which applies the same cut."""
    engine = tools_engine  # the -DPA_TOOLS build: the switch below exists there only
    from pyani_plus_amd.engine import pack_genomes

    rng = np.random.default_rng(77)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)

    def rnd(n: int) -> bytes:
        return rng.choice(acgt, size=n).tobytes()

    big = rnd(700_000) + b"ACC" * 300 + rnd(2_000) + b"AGG" * 200 + rnd(700_000)
    part = np.frombuffer(big[680_000:722_000], dtype=np.uint8).copy()
    hit = rng.random(part.size) < 0.02
    part[hit] = acgt[rng.integers(0, 4, size=int(hit.sum()))]
    genomes = [big, part.tobytes(), rnd(20_000)]
    arena = pack_genomes([b">g%d\n" % i + g + b"\n" for i, g in enumerate(genomes)])
    monkeypatch.setenv("PA_FRAGANI_TRACE", "1")
    seen = {}
    for no_cut in ("0", "1"):
        monkeypatch.setenv("PA_FRAGANI_NO_FREQ_CUT", no_cut)
        capfd.readouterr()
        total, matched, ident_sum = engine.fragani(engine.upload(arena), arena.contig_start, arena.contig_len, arena.contig_genome, K, FRAG)
        err = capfd.readouterr().err
        seen[no_cut] = sum(int(line.split(" fragments, ")[1].split(" seed hits")[0]) for line in err.splitlines() if "batch of genomes" in line)
        if no_cut == "0":
            for q in range(3):
                for r in range(3):
                    ani, m, t = oracle.fragani_pair([genomes[q]], [genomes[r]], K, FRAG, 0.0)
                    assert (int(total[q]), int(matched[q, r])) == (t, m), (q, r)
                    assert m == 0 or mean_f(ident_sum[q, r], m) == ani
            assert matched[1, 0] >= 12 and matched[0, 0] >= total[0] - 2
    assert seen["0"] == _expected_seed_hits(genomes, K, FRAG, cut=True)
    assert seen["1"] == _expected_seed_hits(genomes, K, FRAG, cut=False)
    assert seen["1"] - seen["0"] > 500  # the array's minimizer, ~290 occurrences in the big genome, in the sketches of two fragments


def test_fragment_sketches_where_winnowing_restarted_at_the_fragment_could_differ(engine):
    """fastANI sketches every fragment on its own; the device takes a slice of the genome's minimizers.  Same mappings as the
    oracle (which sketches the fragments on their own) where the two could differ: runs of N across a fragment's start,
    inside its first window and over whole fragments, reverse-palindromic k-mers at a fragment's w-th position."""
    rng = np.random.default_rng(5)
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    w = oracle.fragani_window_size(K, FRAG)
    base = rng.choice(letters, size=14 * FRAG + 700)
    pal = np.frombuffer(b"ACGTACGTACGTACGT", dtype=np.uint8)

    def variant(rate: float, with_gaps: bool) -> bytes:
        g = base.copy()
        hit = rng.random(g.size) < rate
        g[hit] = letters[rng.integers(0, 4, size=int(hit.sum()))]
        g = bytearray(g.tobytes())
        if with_gaps:
            g[1 * FRAG + w - 1 : 1 * FRAG + w - 1 + K] = pal.tobytes()
            g[2 * FRAG + w - 1 : 2 * FRAG + w + 1 + K] = b"AC" + pal.tobytes()
            g[3 * FRAG - 40 : 3 * FRAG + 90] = b"N" * 130
            g[4 * FRAG + 10 : 4 * FRAG + 30] = b"N" * 20
            g[5 * FRAG - 5 : 7 * FRAG + 3] = b"N" * (2 * FRAG + 8)
            g[8 * FRAG + 2900 : 9 * FRAG] = b"N" * 100
            g[10 * FRAG : 10 * FRAG + w + K + 40] = b"N" * (w + K + 40)
        return bytes(g)

    genomes = [variant(0.0, True), variant(0.01, False), variant(0.03, True)]
    texts = [b">g%d\n" % i + g + b"\n" for i, g in enumerate(genomes)]
    total, matched, _ = _check_against_oracle(engine, texts, [[g] for g in genomes])
    assert total.tolist() == [14, 14, 14] and matched[1, 1] == 14 and 10 <= matched[0, 1] <= 12 and matched[0, 0] <= 12


def test_the_path_for_more_than_8192_genomes(tools_engine, monkeypatch):
    """Beyond 8 192 genomes there is no LDS counter per reference genome: every fragment's hits are written out, sorted as a
    whole and cut into segments by head flags (`PA_FRAGANI_HITS=sorted` takes that path for any number of genomes).  Same
    results as the oracle and as the bucketed path: random genomes, the viral fixture, a repeat family, runs of N, and the
    bacterial fixture, whose posting lists lose their most frequent minimizers (the frequency cut moves the postings and
    the minimizer indices this path reads)."""
    engine = tools_engine  # the -DPA_TOOLS build: the switch below exists there only
    from pyani_plus_amd.engine import load_fasta_files, pack_genomes

    texts, contig_lists = _random_genomes(7)
    bucketed = _check_against_oracle(engine, texts, contig_lists)
    files = sorted((GOLDEN / "bacterial_example").glob("*.gz"))
    infos, arena = load_fasta_files(files)
    dev = engine.upload(arena)
    want = engine.fragani(dev, arena.contig_start, arena.contig_len, arena.contig_genome, K, FRAG)
    monkeypatch.setenv("PA_FRAGANI_HITS", "sorted")
    got = _check_against_oracle(engine, texts, contig_lists)
    for a, b in zip(bucketed, got):
        assert np.array_equal(a, b)
    got = engine.fragani(dev, arena.contig_start, arena.contig_len, arena.contig_genome, K, FRAG)
    for a, b in zip(want, got):
        assert np.array_equal(a, b)
    names = [p.name for p in files]
    for q, r, ani, m, t in fixture_rows("bacterial_example"):
        qi, ri = names.index(q), names.index(r)
        bacterial_row_bounds(q, r, ani, m, t, mean_f(got[2][qi, ri], got[1][qi, ri]), int(got[1][qi, ri]), int(got[0][qi]))
    rng = np.random.default_rng(41)
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    unit = rng.choice(letters, size=3_000).tobytes()
    g0 = rng.choice(letters, size=6_000).tobytes() + unit * 5 + b"N" * 3_500 + rng.choice(letters, size=6_000).tobytes()
    g1 = bytearray(g0)
    for i in rng.integers(0, len(g1), size=300):
        g1[i] = ord("ACGT"[int(rng.integers(0, 4))])
    genomes = [g0, bytes(g1)]
    _check_against_oracle(engine, [b">g%d\n" % i + g + b"\n" for i, g in enumerate(genomes)], [[g] for g in genomes])


def test_long_fragments_with_the_widest_window_the_kernels_take(engine):
    """fragLen 8000 at k = 16 is a winnowing window of 64 positions -- the widest ``minimizer_kernel`` takes (its look-back) --,
    windows of ~7 900 window ids and candidate ranges of 24 000: the far end of what the 16-bit offsets inside a stretch and the
    counting sort's span are sized for.  With runs of N (one nearly a fragment long, right before a contig's last kilobases)
    and contigs that end mid-fragment: same integers as the oracle."""
    rng = np.random.default_rng(99)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    frag = 8_000
    root = rng.choice(acgt, size=230_000)
    gap = np.full(7_000, ord("N"), dtype=np.uint8)

    def mutated(seq, rate):
        out = seq.copy()
        hit = rng.random(out.size) < rate
        out[hit] = acgt[rng.integers(0, 4, size=int(hit.sum()))]
        return out

    ref = [np.concatenate((root[:180_000], gap, root[180_000:186_000])).tobytes(), root[186_000:230_000].tobytes()]
    q1 = [mutated(root[:186_000], 0.01).tobytes(), mutated(root[186_000:230_000], 0.02).tobytes()]
    q2 = [np.concatenate((mutated(root[100_000:180_000], 0.03), gap[:6_600], mutated(root[180_000:186_000], 0.03))).tobytes()]
    contig_lists = [ref, q1, q2]
    texts = [b"".join(b">c%d\n" % i + c + b"\n" for i, c in enumerate(cs)) for cs in contig_lists]
    total, matched, _ = _check_against_oracle(engine, texts, contig_lists, frag=frag)
    assert total.tolist() == [24 + 5, 23 + 5, 11] and matched[1, 0] >= 24 and matched[0, 0] >= 25 and matched[2, 0] >= 8


def test_low_complexity_fragments(engine):
    """Inside a homopolymer run or an array of a short unit every window records its minimum anew: a fragment's slice of the
    genome's minimizers then holds one entry per position (thousands, all of one hash) where its sketch holds one hash.  The
    sketch kernel takes runs of equal hashes as one entry on the way in; same mappings as the oracle."""
    rng = np.random.default_rng(8)
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)

    def rnd(n: int) -> bytes:
        return rng.choice(letters, size=n).tobytes()

    g0 = rnd(6_500) + b"A" * 900 + rnd(4_000) + b"AC" * 700 + rnd(3_100) + b"AAC" * 1_100 + rnd(6_000) + b"T" * 3_200 + rnd(5_000)
    g1 = bytearray(g0)
    for i in rng.integers(0, len(g1), size=400):
        g1[i] = ord("ACGT"[int(rng.integers(0, 4))])
    genomes = [g0, bytes(g1), rnd(9_000)]
    total, matched, _ = _check_against_oracle(engine, [b">g%d\n" % i + g + b"\n" for i, g in enumerate(genomes)], [[g] for g in genomes])
    assert matched[0, 1] >= total[0] - 3 and matched[0, 2] == 0


def test_random_genome_sets(engine):
    """Forty random genome sets of tests/tools/fragani_stress.py (repeats, runs of N, short and exactly-ending contigs, lower case,
    four k and four fragment lengths): every ordered pair, device against oracle."""
    import importlib.util
    from pathlib import Path

    from pyani_plus_amd.engine import pack_genomes

    spec = importlib.util.spec_from_file_location("fragani_stress", Path(__file__).resolve().parent / "tools" / "fragani_stress.py")
    stress = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(stress)
    rng = np.random.default_rng(2024)
    pairs = 0
    for _case in range(40):
        k, frag, genomes = stress.make_case(rng)
        if oracle.fragani_window_size(k, frag) > 64:
            continue
        arena = pack_genomes([b"".join(b">c%d\n" % i + c + b"\n" for i, c in enumerate(contigs)) for contigs in genomes])
        total, matched, ident_sum = engine.fragani(engine.upload(arena), arena.contig_start, arena.contig_len, arena.contig_genome, k, frag)
        for q in range(len(genomes)):
            for r in range(len(genomes)):
                ani, m, t = oracle.fragani_pair(genomes[q], genomes[r], k, frag, 0.0)
                assert (int(total[q]), int(matched[q, r])) == (t, m), (_case, k, frag, q, r)
                assert m == 0 or mean_f(ident_sum[q, r], m) == ani, (_case, k, frag, q, r)
                pairs += 1
    assert pairs > 300


def test_one_subject_column_at_a_time(engine):
    """A worker asked for one subject column builds the dictionary of that genome's minimizers only (the query genomes' find
    their hashes in it by value) and keeps that genome's bins only: every column of the bacterial fixture -- the frequency cut
    is active in each -- equals the column of the all-against-all run; so does a range of two.  An index built for a range
    may be taken over for a range inside it, not for one outside."""
    from pyani_plus_amd._capi import HipBackendError
    from pyani_plus_amd.engine import load_fasta_files

    files = sorted((GOLDEN / "bacterial_example").glob("*.gz"))
    infos, arena = load_fasta_files(files)
    dev = engine.upload(arena)
    args = (dev, arena.contig_start, arena.contig_len, arena.contig_genome, K, FRAG)
    total, matched, ident_sum = engine.fragani(*args)
    for r0, r1 in ((0, 1), (1, 2), (2, 3), (3, 4), (1, 3)):
        t, m, s = engine.fragani(*args, ref_range=(r0, r1), columns_only=True)
        assert np.array_equal(t, total) and np.array_equal(m, matched[:, r0:r1]) and np.array_equal(s, ident_sum[:, r0:r1]), (r0, r1)
    # (the last index holds genomes 1 and 2)
    t, m, s = engine.fragani(*args, ref_range=(2, 3), columns_only=True, reuse_index=True)
    assert np.array_equal(m, matched[:, 2:3]) and np.array_equal(s, ident_sum[:, 2:3])
    with pytest.raises(HipBackendError, match="REUSE_INDEX"):
        engine.fragani(*args, ref_range=(0, 2), reuse_index=True)
    t, m, s = engine.fragani(*args)  # and back to every genome a reference
    assert np.array_equal(m, matched) and np.array_equal(s, ident_sum)


def test_query_batches_of_one_subject_column_in_a_fresh_context(engine):
    """The reference's worker maps its queries in batches against one subject (private_cli.py:1029-1063); here the second
    batch takes over the index of the first.  In a context of its own -- a worker process's: every buffer exactly as large
    as the first call made it -- the index must stay where it is (asking for the room of a whole-set dictionary again once
    moved, and lost, the one-genome dictionary)."""
    from pyani_plus_amd.engine import HipEngine, load_fasta_files

    files = sorted((GOLDEN / "bacterial_example").glob("*.gz"))
    infos, arena = load_fasta_files(files)
    args = (arena.contig_start, arena.contig_len, arena.contig_genome, K, FRAG)
    total, matched, ident_sum = engine.fragani(engine.upload(arena), *args)
    fresh = HipEngine(0)
    try:
        dev = fresh.upload(arena)
        out = fresh.fragani(dev, *args, ref_range=(1, 2), query_range=(0, 2), columns_only=True)
        out = fresh.fragani(dev, *args, ref_range=(1, 2), query_range=(2, 4), columns_only=True, reuse_index=True, out=out)
        assert np.array_equal(out[0], total) and np.array_equal(out[1], matched[:, 1:2]) and np.array_equal(out[2], ident_sum[:, 1:2])
    finally:
        fresh.close()


def test_workspace_is_reported_and_a_cap_ends_a_call_with_the_sizes_named(engine):
    """The reference bounds a fastANI worker's memory by batches of 500 queries (pyani_plus/private_cli.py:1029-1033); this
    build's workspace is device memory that stays in the context and only grows.  pa_fragani_workspace reports it; with
    pa_fragani_set_workspace_cap a call that would pass the cap ends with PA_E_NOMEM and a message that names the call and
    the sizes -- not a HIP abort --, the workspace given back, and a call that fits (one subject column of the same genomes)
    goes on working in the same context and gives the column of the uncapped run."""
    from pyani_plus_amd import _capi
    from pyani_plus_amd.engine import HipEngine, load_fasta_files

    files = sorted((GOLDEN / "bacterial_example").glob("*.gz"))
    infos, arena = load_fasta_files(files)
    args = (arena.contig_start, arena.contig_len, arena.contig_genome, K, FRAG)
    total, matched, ident_sum = engine.fragani(engine.upload(arena), *args)
    needs = {}
    for name, kwargs in (("all", {}), ("one", {"ref_range": (1, 2), "columns_only": True})):
        fresh = HipEngine(0)
        try:
            assert fresh.fragani_workspace() == {"held_bytes": 0, "peak_bytes": 0, "cap_bytes": 0}
            fresh.fragani(fresh.upload(arena), *args, **kwargs)
            ws = fresh.fragani_workspace()
            assert ws["held_bytes"] == ws["peak_bytes"] > 0 and ws["cap_bytes"] == 0
            needs[name] = ws["held_bytes"]
        finally:
            fresh.close()
    # 17 Mb of arena: the minimizers and their links alone are ~9.6 bytes per residue; the dictionary of ONE genome is a quarter of all four's
    assert 9 * arena.genome_start[-1] < needs["one"] < needs["all"]
    capped = HipEngine(0)
    try:
        cap = (needs["one"] + needs["all"]) // 2
        capped.fragani_set_workspace_cap(cap)
        dev = capped.upload(arena)
        with pytest.raises(_capi.HipBackendError) as err:
            capped.fragani(dev, *args)
        text = str(err.value)
        assert err.value.status == _capi.PA_E_NOMEM
        assert "pa_fragani: 4 genomes" in text and "reference range [0,4)" in text and f"above the cap of {cap} bytes" in text and "held" in text
        ws = capped.fragani_workspace()  # the call that ended for want of memory gave the workspace back
        assert ws["held_bytes"] <= (1 << 20) and 0 < ws["peak_bytes"] <= cap == ws["cap_bytes"]
        t, m, s = capped.fragani(dev, *args, ref_range=(1, 2), columns_only=True)  # the same context still serves what fits
        assert np.array_equal(t, total) and np.array_equal(m, matched[:, 1:2]) and np.array_equal(s, ident_sum[:, 1:2])
        assert capped.fragani_workspace()["held_bytes"] <= cap
        capped.fragani_set_workspace_cap(0)  # no cap: the whole set
        t, m, s = capped.fragani(dev, *args)
        assert np.array_equal(m, matched) and np.array_equal(s, ident_sum)
    finally:
        capped.close()


def test_product_library_ignores_the_tool_switches(engine, tools_engine, monkeypatch):
    """The switches of tools/ and of the rare-path tests live in the -DPA_TOOLS build only: with every one of them set, the
    product library returns what it returns without them (and what the oracle says); the tools build, with its mapping
    kernel cut short after L1 by the same environment, maps nothing -- the switches are live there."""
    from pyani_plus_amd.engine import load_fasta_files

    texts, contig_lists = _random_genomes(11)
    want = _check_against_oracle(engine, texts, contig_lists)
    files = sorted((GOLDEN / "bacterial_example").glob("*.gz"))[:2]
    _infos, arena = load_fasta_files(files)
    dev = engine.upload(arena)
    want_b = engine.fragani(dev, arena.contig_start, arena.contig_len, arena.contig_genome, K, FRAG)
    for name, value in (("PA_MAP_CUT", "2"), ("PA_FRAGANI_NO_FREQ_CUT", "1"), ("PA_FRAGANI_HITS", "sorted"), ("PA_FRAGANI_SORT_MAX", "1"),
                        ("PA_FRAGANI_BATCH_HITS", "1"), ("PA_FRAGANI_MINIMIZER_ROOM", "1"), ("PA_KMER_VARIANT", "0"), ("PA_PAIRS_SYMMETRIC", "0")):
        monkeypatch.setenv(name, value)
    got = _check_against_oracle(engine, texts, contig_lists)
    for a, b in zip(want, got):
        assert np.array_equal(a, b)
    got_b = engine.fragani(dev, arena.contig_start, arena.contig_len, arena.contig_genome, K, FRAG)
    for a, b in zip(want_b, got_b):
        assert np.array_equal(a, b)
    monkeypatch.delenv("PA_FRAGANI_BATCH_HITS")
    monkeypatch.delenv("PA_FRAGANI_MINIMIZER_ROOM")
    monkeypatch.delenv("PA_FRAGANI_SORT_MAX")
    dev_t = tools_engine.upload(arena)
    cut_short = tools_engine.fragani(dev_t, arena.contig_start, arena.contig_len, arena.contig_genome, K, FRAG)
    assert not np.array_equal(cut_short[1], want_b[1])  # PA_MAP_CUT=2 ends the mapping kernel after L1: nothing is mapped


def test_iupac_letters_are_hashed_as_fastani_hashes_them(engine):
    """fastANI hashes every upper-cased character as it is and leaves what it does not know in place in the reverse
    complement (the reference hands it the FASTA text: /root/reference/pyani_plus/private_cli.py:1044-1063): a k-mer over R,
    Y, K, M, S, W, B, D, H, V -- or any other byte -- is a k-mer with its own hash, not the k-mer over N.  The arena keeps one
    "not ACGT" bit per residue; the packers list the other letters and the kernels look them up.  Letters at fragment
    starts and ends, at contig ends, inside minimizer windows, in runs, mirrored (a k-mer equal to its own reverse
    complement over non-ACGT letters), in lower case, next to N: the minimizers and every integer equal the oracle's."""
    from pyani_plus_amd.engine import pack_genomes

    rng = np.random.default_rng(2024)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    letters = np.frombuffer(b"RYKMSWBDHVryksX*-", dtype=np.uint8)
    root = rng.choice(acgt, size=36_000)
    genomes = []
    for g, rate in enumerate([0.0, 0.01, 0.04, 0.0]):
        seq = root.copy()
        hit = rng.random(seq.size) < rate
        seq[hit] = acgt[rng.integers(0, 4, size=int(hit.sum()))]
        # scattered single letters (about 1 in 150 residues), different ones in different genomes
        where = rng.choice(seq.size, size=240, replace=False)
        seq[where] = letters[rng.integers(0, len(letters), size=where.size)]
        for f in range(0, 12):  # first and last residue of fragments, and the residues around the first window's end
            seq[3000 * f] = letters[(g + f) % 10]
            seq[3000 * f + 2999] = letters[(g + 2 * f) % 10]
            seq[3000 * f + 23 + (f % 5)] = letters[(g + 3 * f) % 10]
        seq[5_000:5_030] = ord("R") if g % 2 else ord("N")  # a run: of R in one genome, of N in its relative
        seq[7_000:7_016] = np.frombuffer(b"ACGTRRYYYYRRACGT", dtype=np.uint8)  # its own reverse complement but for R/Y: a USED k-mer
        seq[9_000:9_016] = np.frombuffer(b"ACGTRYKMMKYRACGT", dtype=np.uint8)  # mirrored letters over a reverse-palindromic frame: passed over
        seq[12_000] = ord("n")
        seq[12_001] = ord("R")
        genomes.append(seq.tobytes())
    cuts = [[36_000], [9_000, 27_000], [36_000], [2_999, 3_001, 30_000]]
    contig_lists, texts = [], []
    for g, seq in enumerate(genomes):
        contigs, pos = [], 0
        for n in cuts[g]:
            contigs.append(seq[pos : pos + n])
            pos += n
        contig_lists.append(contigs)
        texts.append(b"".join(b">c%d\n" % i + c + b"\n" for i, c in enumerate(contigs)))
    arena = pack_genomes(texts)
    assert len(arena.ambig_pos) > 900 and ord("N") not in set(arena.ambig_byte.tolist())
    dev = engine.upload(arena)
    for k, w in ((16, 24), (15, 19), (12, 9)):
        h, wp, ct = engine.fragani_sketch(dev, arena.contig_start, arena.contig_len, arena.contig_genome, k, w)
        ci = 0
        differs_from_n = False
        for contigs in contig_lists:
            for contig in contigs:
                want_h, want_p = oracle.fragani_minimizers(contig, k, w)
                sel = ct == ci
                assert np.array_equal(h[sel], want_h), f"k={k} contig {ci}: minimizers differ from the oracle's"
                assert np.array_equal(wp[sel].astype(np.int32), want_p)
                as_n = bytes(c if c in b"ACGTacgt" else ord("N") for c in contig)
                differs_from_n = differs_from_n or not np.array_equal(oracle.fragani_minimizers(as_n, k, w)[0], want_h)
                ci += 1
        assert differs_from_n  # the letters matter: read as N the same contigs give other minimizers
    _check_against_oracle(engine, texts, contig_lists)
    _check_against_oracle(engine, texts, contig_lists, frag=1000, k=15)
    # the same arena without its list: every such residue is an N again (and the library forgets a list it was given)
    plain = pack_genomes(texts)
    plain.ambig_pos, plain.ambig_byte = None, None
    h_n, _wp, _ct = engine.fragani_sketch(engine.upload(plain), plain.contig_start, plain.contig_len, plain.contig_genome, 16, 24)
    all_n = [bytes(c if c in b"ACGTacgt" else ord("N") for c in contig) for contigs in contig_lists for contig in contigs]
    assert np.array_equal(h_n, np.concatenate([oracle.fragani_minimizers(c, 16, 24)[0] for c in all_n]))


def test_sparse_segments_equal_the_general_kernel(tools_engine, monkeypatch):
    """Segments of at most eight seed hits -- diverged pairs -- take ``map_sparse_kernel`` (the hit-by-hit form of the windowed
    MinHash); with ``PA_FRAGANI_SPARSE=0`` (tools build) they go through ``map_segments_kernel`` like the rest.  Same
    integers and float sums either way, and the oracle's: pairs at 15 to 25 % divergence (a handful of hits per fragment),
    with repeats (a hash twice in a stretch: handed on), runs of N, short contigs and contig ends."""
    engine = tools_engine  # the -DPA_TOOLS build: the switch below exists there only
    from pyani_plus_amd.engine import pack_genomes

    rng = np.random.default_rng(515)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    root = rng.choice(acgt, size=60_000)
    root[20_000:23_000] = root[5_000:8_000]  # a dispersed repeat
    genomes = []
    for rate in (0.0, 0.08, 0.12, 0.16, 0.2):
        seq = root.copy()
        hit = rng.random(seq.size) < rate
        seq[hit] = acgt[rng.integers(0, 4, size=int(hit.sum()))]
        genomes.append(seq)
    genomes[2][30_000:30_200] = ord("N")
    cuts = [[60_000], [25_000, 35_000], [60_000], [3_100, 56_900], [60_000]]
    texts, contig_lists = [], []
    for g, seq in enumerate(genomes):
        contigs, pos = [], 0
        for n in cuts[g]:
            contigs.append(seq[pos : pos + n].tobytes())
            pos += n
        contig_lists.append(contigs)
        texts.append(b"".join(b">c%d\n" % i + c + b"\n" for i, c in enumerate(contigs)))
    arena = pack_genomes(texts)
    dev = engine.upload(arena)
    results = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("PA_FRAGANI_SPARSE", mode)
        monkeypatch.setenv("PA_FRAGANI_TRACE", "1")
        results[mode] = engine.fragani(dev, arena.contig_start, arena.contig_len, arena.contig_genome, K, FRAG)
    for a, b in zip(results["1"], results["0"]):
        assert np.array_equal(a, b)
    monkeypatch.setenv("PA_FRAGANI_SPARSE", "1")
    total, matched, _ = _check_against_oracle(engine, texts, contig_lists)
    assert matched[0, 4] > 0 and matched[4, 0] > 0  # the most diverged pair still maps fragments: through the sparse kernel
    _check_against_oracle(engine, texts, contig_lists, frag=1000, k=15)


@pytest.mark.gpu
def test_sparse_segments_next_to_low_complexity_sequence(tools_engine, monkeypatch):
    """The sparse kernel starts a candidate at the group of begins at its first seed hit, which it finds among the 512 window
    ids it loads from the start of the candidate's range -- or, when the range holds more minimizers than that before the
    hit, through the bucket index.  Random sequence never does (a fragment holds ~240 minimizers whatever its length); a
    homopolymer run or an array of a short unit in the REFERENCE does, one minimizer per position.  Diverged pairs (a handful
    of hits per fragment) with such runs a few hundred to two thousand residues before the matching sequence: the same
    integers and float sums with and without the sparse kernel, and the oracle's."""
    engine = tools_engine
    from pyani_plus_amd.engine import pack_genomes

    rng = np.random.default_rng(626)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)

    def rnd(n: int) -> np.ndarray:
        return rng.choice(acgt, size=n)

    def run_of(unit: bytes, n: int) -> np.ndarray:
        return np.frombuffer((unit * (n // len(unit) + 1))[:n], dtype=np.uint8)

    parts, marks = [], []
    for unit, n, gap in ((b"A", 1_500, 300), (b"AC", 1_400, 1_200), (b"T", 2_600, 2_000), (b"AAC", 900, 100), (b"G", 700, 2_600)):
        parts += [rnd(int(rng.integers(4_000, 9_000))), run_of(unit, n)]
        marks.append(gap)
    parts.append(rnd(8_000))
    root = np.concatenate(parts)
    genomes = [root]
    for rate in (0.14, 0.18, 0.22):
        seq = root.copy()
        hit = rng.random(seq.size) < rate
        seq[hit] = acgt[rng.integers(0, 4, size=int(hit.sum()))]
        genomes.append(seq)
    texts = [b">g%d\n" % i + g.tobytes() + b"\n" for i, g in enumerate(genomes)]
    contig_lists = [[g.tobytes()] for g in genomes]
    arena = pack_genomes(texts)
    dev = engine.upload(arena)
    results = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("PA_FRAGANI_SPARSE", mode)
        results[mode] = engine.fragani(dev, arena.contig_start, arena.contig_len, arena.contig_genome, K, FRAG)
    for a, b in zip(results["1"], results["0"]):
        assert np.array_equal(a, b)
    monkeypatch.setenv("PA_FRAGANI_SPARSE", "1")
    total, matched, _ = _check_against_oracle(engine, texts, contig_lists)
    assert matched[3, 0] > 0 and matched[0, 3] > 0
    _check_against_oracle(engine, texts, contig_lists, frag=1000, k=14)

