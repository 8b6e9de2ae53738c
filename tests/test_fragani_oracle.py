"""The fastANI-style fragment-ANI oracle against the reference's fastANI fixtures (tolerance only).

fastANI's internals are not in the reference tree, so parity for this method is a stated
tolerance on the 25 output rows the reference holds
(tests/fixtures/{viral,bacterial}_example/intermediates/fastANI/all_vs_*.fastani, byte-compared by the
reference itself at tests/snakemake/test_fastani_workflow.py:67-86):
    total fragments   exact  (= sum over contigs of floor(len / fragLen))
    kept fragments    within 1 % of the total (and within 1 for the phages)
    ANI               within 0.1 percentage points
tests/tools/fragani_bisect.py measures each restatement choice against the 25 rows (profiles/r02_fragani_bisect.md):
with Mashmap's sketch-size list (window 24), fastANI's fragLen-20 reference buckets and Mashmap's slide over the
reference minimizer positions the maximum deviations are 0.070 points and 0.71 %.
"""

from __future__ import annotations

import math
from pathlib import Path

import numpy as np
import pytest

import oracle
from tests.helpers import GOLDEN, read_fasta_bytes

ANI_TOL = 0.1  # percentage points (measured maximum over the 25 rows: 0.070)
MATCHED_TOL = 0.01  # kept fragments, as a fraction of the total fragments (measured maximum: 0.71 %)
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
    assert abs(oracle.fragani_identity(100, 200, 16) - 100 * (1 + math.log(2 * 0.5 / 1.5) / 16)) < 1e-12
    assert oracle.fragani_kmer_hash(b"ACGTACGTACGTACGN") == 0xFFFFFFFF  # non-ACGT -> skipped
    assert oracle.fragani_kmer_hash(b"ACGTTGCATGCATGCA") == oracle.fragani_kmer_hash(b"TGCATGCATGCAACGT")  # strand-symmetric


def test_minimizers_of_a_fragment_are_a_slice_of_the_genome_minimizers():
    """The HIP path never re-sketches fragments: a fragment's sketch is the genome's minimizers whose
    window ids fall in the fragment, plus the one still active at its first window."""
    seq = contigs_of(GOLDEN / "viral_example" / "OP073605.fasta")[0]
    w = oracle.fragani_window_size(K, FRAG)
    gh, gp = oracle.fragani_minimizers(seq, K, w)
    cw = FRAG - (w - 1) - (K - 1)
    for f in range(len(seq) // FRAG):
        fh, fp = oracle.fragani_minimizers(seq[f * FRAG : (f + 1) * FRAG], K, w)
        b, e = int(np.searchsorted(gp, f * FRAG)), int(np.searchsorted(gp, f * FRAG + cw))
        b0 = b - 1 if b > 0 and (b == len(gp) or gp[b] > f * FRAG) else b
        assert np.array_equal(fh, gh[b0:e])
        assert np.array_equal(fp, np.maximum(gp[b0:e] - f * FRAG, 0))


def test_viral_rows_within_tolerance():
    genomes = {p.name: contigs_of(p) for p in (GOLDEN / "viral_example").glob("*.f*")}
    for q, r, ani, matched, total in fixture_rows("viral_example"):
        got_ani, got_m, got_t = oracle.fragani_pair(genomes[q], genomes[r], K, FRAG, 0.2)
        assert got_t == total
        assert abs(got_m - matched) <= 1
        assert abs(got_ani - ani) <= ANI_TOL, (q, r, got_ani, ani)


@pytest.mark.parametrize(
    "q,r", [("NC_010338.fna.gz", "NC_002696.fasta.gz"), ("NC_014100.fna.gz", "NC_011916.fas.gz"), ("NC_002696.fasta.gz", "NC_011916.fas.gz")]
)
def test_bacterial_rows_within_tolerance(q, r):
    """Three of the 16 bacterial rows (an 83 %, an 86 % and a 99.99 % pair); all 16 are within the same bounds
    (tests/tools/fragani_bisect.py runs them all; the GPU test checks all 16 on the device)."""
    rows = {(a, b): (ani, m, t) for a, b, ani, m, t in fixture_rows("bacterial_example")}
    ani, matched, total = rows[(q, r)]
    got_ani, got_m, got_t = oracle.fragani_pair(contigs_of(GOLDEN / "bacterial_example" / q), contigs_of(GOLDEN / "bacterial_example" / r), K, FRAG, 0.2)
    assert got_t == total  # 1338 / 1825 / 1347 / 1551: sum over contigs of floor(len / 3000)
    assert abs(got_m - matched) <= MATCHED_TOL * total
    assert abs(got_ani - ani) <= ANI_TOL


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
