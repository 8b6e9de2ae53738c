"""The fastANI-style fragment-ANI oracle against the reference's fastANI fixtures.

fastANI's internals are not in the reference tree; the oracle restates the published method and its choices were
bisected against the values the reference holds (tests/tools/fragani_bisect.py, profiles/r04_fragani_bisect.md).  With
the defaults of round 4 -- the exact slide (the window at every reference position holds the minimizers of the windows
[i, i + count_windows), the slide ends when the window's end reaches the candidate's last end), a window's position = the
window id of its first minimizer, the LAST of a fragment's equally good candidates, float identities summed in float in
(contig, bin) order -- every one of the 25 rows comes out exactly as fastANI wrote it: the identity as printed (six
significant digits), the kept fragments, the total fragments
(tests/fixtures/{viral,bacterial}_example/intermediates/fastANI/all_vs_*.fastani, byte-compared by the reference itself
at tests/snakemake/test_fastani_workflow.py:67-86).
"""

from __future__ import annotations

import math
from pathlib import Path

import numpy as np
import pytest

import oracle
from tests.helpers import GOLDEN, read_fasta_bytes

ANI_TOL = 0.0  # percentage points: every row prints as fastANI's
MATCHED_TOL = 0.0  # kept fragments: every row exact


def printed(ani: float) -> float:
    """fastANI prints the identity with six significant digits (``82.9124``, ``100``)"""
    return float(f"{ani:.6g}")
K, FRAG = 16, 3000


def contigs_of(path: Path) -> list[bytes]:
    text = read_fasta_bytes(path)
    return [b"".join(rec.split(b"\n")[1:]).translate(None, b" \t\r") for rec in text.split(b">")[1:]]


def fixture_rows(name: str) -> list[tuple[str, str, float, int, int]]:
    rows = []
    for f in sorted((GOLDEN / name / "fastANI").glob("*.fastani")):
        for line in f.read_text().splitlines():
            q, r, ani, matched, total = line.split()
            rows.append((Path(q).name, Path(r).name, float(ani), int(matched), int(total)))
    return rows


def test_parameters():
    assert oracle.fragani_window_size(16, 3000) == 24  # the window fastANI logs for its defaults
    assert oracle.fragani_window_size(15, 2000) == 20
    min_hits, min_shared = oracle.fragani_tables(16, 300)
    assert np.all(np.diff(min_hits[1:]) >= 0) and min_hits[1] == 1
    assert np.all(min_shared[1:] >= 0) and min_shared[260] >= min_hits[260] - 1
    assert oracle.fragani_identity(10, 10, 16) == 100.0
    # fastANI holds Jaccard, Mash distance and identity as floats: the value is a float, widened
    want = np.float32(100) * (np.float32(1) - np.float32((-1.0 / 16) * math.log(2.0 * 0.5 / float(np.float32(1) + np.float32(0.5)))))
    assert oracle.fragani_identity(100, 200, 16) == float(want)
    assert oracle.fragani_identity(7, 240, 16) == float(np.float32(oracle.fragani_identity(7, 240, 16)))
    # fastANI hashes the characters as they are: a k-mer holding an N is a k-mer (its reverse complement keeps the N in
    # place); only k-mers whose two strands hash alike are passed over -- reverse palindromes, runs of N
    assert oracle.fragani_kmer_hash(b"ACGTACGTACGTACGN") not in (0xFFFFFFFF, oracle.fragani_kmer_hash(b"ACGTACGTACGTACGA"))
    assert oracle.fragani_kmer_hash(b"ACGTACGTACGTACGN") == oracle.fragani_kmer_hash(b"NCGTACGTACGTACGT") == oracle.fragani_kmer_hash(b"acgtacgtacgtacgn")
    assert oracle.fragani_kmer_hash(b"N" * 16) == 0xFFFFFFFF and oracle.fragani_kmer_hash(b"ACGTACGTACGTACGT") == 0xFFFFFFFF
    assert oracle.fragani_kmer_hash(b"ACGTNNNNNNNNACGT") == 0xFFFFFFFF  # its own reverse complement, the N in place
    assert oracle.fragani_kmer_hash(b"ACGTTGCATGCATGCA") == oracle.fragani_kmer_hash(b"TGCATGCATGCAACGT")  # strand-symmetric


def fragment_slice(gh: np.ndarray, gp: np.ndarray, seq: bytes, f: int, k: int, w: int, frag: int) -> np.ndarray:
    """The HIP path's sketch of fragment f (query_sketch_kernel): the genome's minimizers recorded at the fragment's window ids,
    plus the one recorded last before them unless a new one is recorded by the first window of the fragment at which any is
    selected -- the window of the first USED k-mer at or after the fragment's w-th (d windows in: 0 unless that k-mer holds
    an N or is its own reverse complement).  No used k-mer from there to the fragment's end: no sketch."""
    cw = frag - (w - 1) - (k - 1)
    p = f * frag
    d = 0
    while d < cw and oracle.fragani_kmer_hash(seq[p + w - 1 + d : p + w - 1 + d + k]) == 0xFFFFFFFF:
        d += 1
    if d == cw:
        return gh[:0]
    b, e = int(np.searchsorted(gp, p)), int(np.searchsorted(gp, p + cw))
    fresh = b < e and gp[b] <= p + d
    return gh[b - 1 if not fresh and b > 0 else b : e]


def test_minimizers_of_a_fragment_are_a_slice_of_the_genome_minimizers():
    """The HIP path never re-sketches fragments (fastANI does, and so does the oracle): a fragment's sketch is a slice of its
    genome's minimizers.  Same hashes in the same order -- on a fixture, and where winnowing restarted at the fragment's first
    residue could differ: runs of N across and inside fragment starts, reverse-palindromic k-mers at a fragment's w-th
    position (no minimizer is selected at that window), a fragment of N only."""
    w = oracle.fragani_window_size(K, FRAG)
    rng = np.random.default_rng(5)
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    synth = bytearray(rng.choice(letters, size=12 * FRAG + 700).tobytes())
    pal = b"ACGTACGTACGTACGT"  # its own reverse complement: both strands hash alike, the k-mer is not used
    synth[1 * FRAG + w - 1 : 1 * FRAG + w - 1 + K] = pal
    synth[2 * FRAG + w - 1 : 2 * FRAG + w + 1 + K] = b"AC" + pal  # ... and the two k-mers after it (period 2 of the palindrome's own shift)
    synth[3 * FRAG - 40 : 3 * FRAG + 90] = b"N" * 130             # a run of N across a fragment's start
    synth[4 * FRAG + 10 : 4 * FRAG + 30] = b"N" * 20              # ... inside its first window
    synth[5 * FRAG - 5 : 7 * FRAG + 3] = b"N" * (2 * FRAG + 8)     # fragments 5 and 6 hold nothing usable (6: N only)
    synth[8 * FRAG + 2900 : 9 * FRAG] = b"N" * 100                # N at a fragment's end
    for seq in (contigs_of(GOLDEN / "viral_example" / "OP073605.fasta")[0], bytes(synth)):
        gh, gp = oracle.fragani_minimizers(seq, K, w)
        for f in range(len(seq) // FRAG):
            fh, _fp = oracle.fragani_minimizers(seq[f * FRAG : (f + 1) * FRAG], K, w)
            assert np.array_equal(fh, fragment_slice(gh, gp, seq, f, K, w, FRAG)), f
    assert len(oracle.fragani_minimizers(bytes(synth[6 * FRAG : 7 * FRAG]), K, w)[0]) == 0


def test_viral_rows_print_as_fastani_does():
    """All nine rows of the viral fixture: identity equal after fastANI's six-digit print, kept and total fragments exact."""
    genomes = {p.name: contigs_of(p) for p in (GOLDEN / "viral_example").glob("*.f*")}
    for q, r, ani, matched, total in fixture_rows("viral_example"):
        got_ani, got_m, got_t = oracle.fragani_pair(genomes[q], genomes[r], K, FRAG, 0.2)
        assert (got_m, got_t) == (matched, total), (q, r)
        assert printed(got_ani) == ani, (q, r, got_ani, ani)


def test_self_hits_of_the_small_contigs():
    """/root/reference/tests/test_self_vs_self.py:90-91 and 121-122: MIBY01000005 against itself is exactly 100 %, MIBY01000011
    prints 99.9953 -- its last fragment ends one residue before the contig does, the slide stops before the window that
    would take in the contig's last minimizer, and the best window left shares 223 of 225."""
    small, large = contigs_of(GOLDEN / "MIBY01000005.fasta"), contigs_of(GOLDEN / "MIBY01000011.fasta")
    assert oracle.fragani_pair(small, small, K, FRAG, 0.2) == (100.0, 2, 2)
    ani, m, t = oracle.fragani_pair(large, large, K, FRAG, 0.2)
    assert (m, t) == (6, 6) and printed(ani) == 99.9953 and ani == float(np.float32(ani))  # a float mean, as fastANI's
    maps, _ = oracle.fragani_map(large, large, K, FRAG)
    assert [(int(a), int(b)) for a, b in zip(maps["shared"], maps["s"])][-1] == (223, 225)
    assert all(int(a) == int(b) for a, b in list(zip(maps["shared"], maps["s"]))[:-1])


def bacterial_row_bounds(q: str, r: str, ani: float, matched: int, total: int, got_ani: float, got_m: int, got_t: int) -> None:
    """What a row has to satisfy (shared with the GPU test, which checks all 25 on the device): everything fastANI wrote."""
    assert (got_m, got_t) == (matched, total), (q, r, got_m, got_t)
    assert printed(got_ani) == ani, (q, r, got_ani)


BACTERIAL_SAMPLE = [("NC_010338.fna.gz", "NC_002696.fasta.gz"), ("NC_014100.fna.gz", "NC_011916.fas.gz"), ("NC_002696.fasta.gz", "NC_011916.fas.gz"),
                    ("NC_011916.fas.gz", "NC_002696.fasta.gz"), ("NC_014100.fna.gz", "NC_014100.fna.gz"), ("NC_011916.fas.gz", "NC_011916.fas.gz"),
                    ("NC_010338.fna.gz", "NC_010338.fna.gz"), ("NC_002696.fasta.gz", "NC_014100.fna.gz")]  # fmt: skip


def _bacterial_pair(pair):
    q, r = pair
    return oracle.fragani_pair(contigs_of(GOLDEN / "bacterial_example" / q), contigs_of(GOLDEN / "bacterial_example" / r), K, FRAG, 0.2)


def test_bacterial_rows():
    """Eight of the 16 bacterial rows (two 83 % pairs, two 86 % pairs, the 99.99 % pair both ways, three self rows), on a
    process pool; all 16 are exact (tests/tools/fragani_bisect.py runs them all; the GPU test checks all 16 on the device).
    The self rows are where the choice among equally good candidates shows: fragment 156 of NC_011916 lies in a repeat, maps
    onto both copies with all of its 231 minimizers, and fastANI keeps the second -- in the bin fragment 197 maps to, so
    1346 of 1347 are kept; NC_010338 loses five fragments and NC_014100 two the same way."""
    from concurrent.futures import ProcessPoolExecutor

    rows = {(a, b): (ani, m, t) for a, b, ani, m, t in fixture_rows("bacterial_example")}
    with ProcessPoolExecutor(max_workers=min(8, len(BACTERIAL_SAMPLE))) as pool:
        for (q, r), got in zip(BACTERIAL_SAMPLE, pool.map(_bacterial_pair, BACTERIAL_SAMPLE)):
            bacterial_row_bounds(q, r, *rows[(q, r)], *got)


def test_the_last_of_equally_good_candidates_is_kept():
    """A fragment that lies in an exact repeat maps onto both copies with all of its minimizers; fastANI keeps the later one."""
    rng = np.random.default_rng(11)
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    unit = rng.choice(letters, size=6_000).tobytes()
    genome = rng.choice(letters, size=9_000).tobytes() + unit + rng.choice(letters, size=12_000).tobytes() + unit + rng.choice(letters, size=3_000).tobytes()
    maps, total = oracle.fragani_map([genome], [genome])
    assert total == 12 and list(maps["frag"]) == list(range(12))
    pos = {int(f): int(p) for f, p in zip(maps["frag"], maps["ref_pos"])}
    for f in (3, 4):  # the fragments inside the first copy map onto the second: 27 000 window ids further on
        assert abs(pos[f] - (f * 3000 + 18_000)) <= 24 and int(maps["shared"][f]) == int(maps["s"][f])
    for f in (9, 10):  # those inside the second copy onto themselves
        assert abs(pos[f] - f * 3000) <= 24
    assert oracle.fragani_pair([genome], [genome])[1] == 10  # two bins hold two fragments each


def test_min_fraction_and_unrelated_genomes():
    rng = np.random.default_rng(4)
    a = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=30_000).tobytes()
    b = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=30_000).tobytes()
    ani, m, t = oracle.fragani_pair([a], [b])
    assert math.isnan(ani) and m == 0 and t == 10  # nothing maps -> no output line -> NULL
    half = a[:15_000] + b[15_000:]
    ani, m, t = oracle.fragani_pair([half], [a], min_fraction=0.2)
    assert m == 5 and t == 10 and ani > 99.9
    ani, m, t = oracle.fragani_pair([half], [a], min_fraction=0.6)
    assert math.isnan(ani) and m == 5  # below minFraction: fastANI prints nothing
    # minFraction is relative to the SHORTER genome: a long query against a short reference is still reported
    c = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=270_000).tobytes()
    ani, m, t = oracle.fragani_pair([a + c], [a], min_fraction=0.2)  # 10 of 100 fragments, but all of the 30 kb reference
    assert t == 100 and m == 10 and ani > 99.9
    assert oracle.fragani_pair([a[:2999]], [a])[2] == 0  # shorter than one fragment


def test_fastani_mean_is_a_float_quotient():
    """Where the product forms a pair's ANI (methods/fastani_hip.fastani_mean): the library's float sum by the float count,
    in float -- the oracle's own mean, so that the six printed digits are fastANI's."""
    from pyani_plus_amd.methods.fastani_hip import fastani_mean, fastani_print_round

    small = contigs_of(GOLDEN / "MIBY01000011.fasta")
    ani, m, _t = oracle.fragani_pair(small, small, K, FRAG, 0.2)
    maps, _ = oracle.fragani_map(small, small, K, FRAG)
    total = np.float32(0)
    for shared, s_ in zip(maps["shared"], maps["s"]):  # (one contig: fragment order is bin order)
        total = total + np.float32(oracle.fragani_identity(int(shared), int(s_), K))
    assert float(fastani_mean(float(total), m)) == ani and fastani_print_round(ani) == 99.9953
    got = fastani_mean(np.array([[float(total), 0.0]]), np.array([[m, 0]], dtype=np.uint32))
    assert got.shape == (1, 2) and got[0, 0] == ani and np.isnan(got[0, 1])


# ---------------------------------------------------------------- the tuned form of the L2 evaluation (bench.py's CPU baseline)
def _pair_tuned(pair):
    from oracle import pyoracle

    q, r = pair
    pyoracle.fragani_set_fast(True)
    try:
        return oracle.fragani_pair(contigs_of(GOLDEN / "bacterial_example" / q), contigs_of(GOLDEN / "bacterial_example" / r), K, FRAG, 0.2)
    finally:
        pyoracle.fragani_set_fast(False)


def test_tuned_form_reproduces_every_fastani_row():
    """The oracle's tuned L2 (the window kept as the slide moves: what bench.py times as the CPU baseline) against everything
    the checking form is pinned to: all 16 bacterial and all 9 viral rows, the self hits of the two small contigs."""
    from concurrent.futures import ProcessPoolExecutor

    from oracle import pyoracle

    rows = {(a, b): (ani, m, t) for a, b, ani, m, t in fixture_rows("bacterial_example")}
    assert len(rows) == 16
    with ProcessPoolExecutor(max_workers=8) as pool:
        for (q, r), got in zip(rows, pool.map(_pair_tuned, list(rows))):
            bacterial_row_bounds(q, r, *rows[(q, r)], *got)
    pyoracle.fragani_set_fast(True)
    try:
        genomes = {p.name: contigs_of(p) for p in (GOLDEN / "viral_example").glob("*.f*")}
        for q, r, ani, matched, total in fixture_rows("viral_example"):
            got_ani, got_m, got_t = oracle.fragani_pair(genomes[q], genomes[r], K, FRAG, 0.2)
            assert (got_m, got_t) == (matched, total) and printed(got_ani) == ani, (q, r)
        small, large = contigs_of(GOLDEN / "MIBY01000005.fasta"), contigs_of(GOLDEN / "MIBY01000011.fasta")
        assert oracle.fragani_pair(small, small, K, FRAG, 0.2) == (100.0, 2, 2)
        ani, m, t = oracle.fragani_pair(large, large, K, FRAG, 0.2)
        assert (m, t) == (6, 6) and printed(ani) == 99.9953
    finally:
        pyoracle.fragani_set_fast(False)


def test_tuned_form_equals_the_checking_form_fragment_by_fragment():
    """Every mapping (fragment, contig, position, shared, sketch size) of the two forms on sequences made to stress the window
    bookkeeping: repeats inside and across contigs (a hash several times in one window), low-complexity stretches (few
    distinct hashes: the union shorter than the sketch), runs of N, mutated copies at several rates, short contigs, k = 15 /
    fragLen 2000 as the coverage fixtures use."""
    from oracle import pyoracle

    rng = np.random.default_rng(11)
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)

    def mutate(seq: np.ndarray, rate: float) -> np.ndarray:
        out = seq.copy()
        at = rng.random(len(out)) < rate
        out[at] = rng.choice(letters, size=int(at.sum()))
        return out

    base = rng.choice(letters, size=60_000)
    unit = rng.choice(letters, size=700)
    ref = base.copy()
    ref[5_000:12_000] = np.tile(unit, 10)                    # a tandem repeat: every hash ten times over
    ref[20_000:23_000] = ref[30_000:33_000]                  # an exact repeat a fragment long
    ref[40_000:40_400] = np.frombuffer(b"ACAC" * 100, dtype=np.uint8)  # low complexity
    ref[50_000:50_120] = ord("N")
    refs = [ref[:41_000].tobytes(), ref[41_000:].tobytes(), rng.choice(letters, size=3_500).tobytes()]
    queries = [
        [mutate(ref, 0.0).tobytes()],
        [mutate(ref, 0.02)[1_234:].tobytes()],
        [mutate(ref, 0.08).tobytes(), mutate(ref[:9_000], 0.15).tobytes()],
        [np.tile(unit, 30).tobytes()],
        [rng.choice(letters, size=20_000).tobytes()],
    ]
    compared = 0
    for k, frag in ((16, 3000), (15, 2000), (16, 1000)):
        for q in queries:
            pyoracle.fragani_set_fast(False)
            slow, total = oracle.fragani_map(q, refs, k, frag)
            pyoracle.fragani_set_fast(True)
            try:
                fast, total_f = oracle.fragani_map(q, refs, k, frag)
            finally:
                pyoracle.fragani_set_fast(False)
            assert total == total_f
            for name in slow:
                assert np.array_equal(slow[name], fast[name]), (k, frag, name)
            compared += len(slow["frag"])
    assert compared > 150
