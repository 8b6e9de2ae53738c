"""Pin the CPU oracle against the reference's own sourmash fixtures.

Golden data (copied as data from /root/reference/tests/fixtures):
  * 9 `.sig` files  (asserted key-for-key by the reference at
    tests/snakemake/test_sourmash_workflow.py:43-67,103-106)
  * 3 manysearch.csv (27 rows x 15 columns)
  * sourmash_{identity,coverage}.tsv matrices (tests/snakemake/__init__.py:83-166)
  * the two constants of tests/test_coverage.py:169-174 (scaled=50, N runs)
"""

from __future__ import annotations

import hashlib

import numpy as np
import pytest

import oracle
from tests.helpers import FIXTURE_SETS, GOLDEN, load_manysearch, load_matrix_tsv, load_sig, read_fasta_bytes, md5_hex

K = 31


def test_murmur_known_answers():
    # MurmurHash3_x64_128 reference vectors (seed 0 / 42), first 64-bit word.
    assert oracle.murmur3_h1(b"", 0) == 0
    assert oracle.murmur3_h1(b"hello", 0) == 0xCBD8A7B341BD9B02
    assert oracle.murmur3_h1(b"The quick brown fox jumps over the lazy dog", 0) == 0xE34BBC7BBC071B6C
    # sourmash documents hash_murmur("ACTG")-style usage with seed 42; pin one 31-mer via the fixtures below.


def test_max_hash_matches_fixture_sigs():
    assert oracle.max_hash(300) == 61489146912365176
    assert oracle.max_hash(1000) == 18446744073709552
    for name, (scaled, genomes) in FIXTURE_SETS.items():
        for md5 in genomes:
            sig = load_sig(GOLDEN / name / "sourmash" / f"{md5}.sig")["signatures"][0]
            assert sig["max_hash"] == oracle.max_hash(scaled)
            assert sig["num"] == 0 and sig["seed"] == 42 and sig["ksize"] == K


@pytest.mark.parametrize("name", list(FIXTURE_SETS))
def test_sketch_reproduces_sig_fixtures(name):
    scaled, genomes = FIXTURE_SETS[name]
    for md5, fasta in genomes.items():
        text = read_fasta_bytes(GOLDEN / name / fasta)
        assert md5_hex(text) == md5  # genome identity = md5 of decompressed bytes
        mins, _n = oracle.sketch_fasta_text(text, K, scaled)
        sig = load_sig(GOLDEN / name / "sourmash" / f"{md5}.sig")
        want = sig["signatures"][0]
        assert sig["name"] == md5
        assert mins.tolist() == want["mins"]
        # signature md5sum = md5(str(ksize) + concatenated decimal mins)
        digest = hashlib.md5((str(K) + "".join(str(int(h)) for h in mins)).encode()).hexdigest()  # noqa: S324
        assert digest == want["md5sum"]


@pytest.mark.parametrize("name", list(FIXTURE_SETS))
def test_pairs_reproduce_manysearch_rows(name):
    scaled, genomes = FIXTURE_SETS[name]
    md5s = sorted(genomes)
    sketches = [np.array(load_sig(GOLDEN / name / "sourmash" / f"{m}.sig")["signatures"][0]["mins"], dtype=np.uint64) for m in md5s]
    counts = oracle.pair_counts(sketches)
    sizes = [len(s) for s in sketches]
    ident, cov, null = oracle.ani(counts, sizes, sizes, K)
    rows = load_manysearch(GOLDEN / name / "sourmash" / "manysearch.csv")
    seen = set()
    for row in rows:
        q, s = md5s.index(row["query_name"]), md5s.index(row["match_name"])
        seen.add((q, s))
        assert int(row["intersect_hashes"]) == counts[q, s]
        assert not null[q, s]
        # bit-for-bit: the CSV text is the shortest round-trip repr of the double
        assert float(row["query_containment_ani"]) == cov[q, s]
        assert float(row["max_containment_ani"]) == ident[q, s]
        assert float(row["containment"]) == counts[q, s] / sizes[q]
        if q == s:
            assert repr(float(ident[q, s])) == "1.0" == row["max_containment_ani"]
    # rows absent from the CSV are exactly the zero-intersection pairs -> NULL
    for q in range(len(md5s)):
        for s in range(len(md5s)):
            assert ((q, s) in seen) == (counts[q, s] > 0) == (not null[q, s])


@pytest.mark.parametrize("name", list(FIXTURE_SETS))
def test_matrices_match_reference_tsv(name):
    scaled, genomes = FIXTURE_SETS[name]
    md5s = sorted(genomes)
    sketches = []
    for m in md5s:
        mins, _ = oracle.sketch_fasta_text(read_fasta_bytes(GOLDEN / name / genomes[m]), K, scaled)
        sketches.append(mins)
    counts = oracle.pair_counts(sketches)
    sizes = [len(s) for s in sketches]
    ident, cov, null = oracle.ani(counts, sizes, sizes, K)
    # reference matrices are labelled by FASTA stem and sorted by md5 (db_orm.py:407)
    for fname, mat in (("sourmash_identity.tsv", ident), ("sourmash_coverage.tsv", cov)):
        labels, want = load_matrix_tsv(GOLDEN / name / "matrices" / fname)
        stems = [genomes[m].split(".")[0] for m in md5s]
        order = [stems.index(lab) for lab in labels]
        got = mat[np.ix_(order, order)]
        assert np.array_equal(np.isnan(want), np.isnan(got))
        # the reference compares these with atol=2e-8 (tests/snakemake/__init__.py:86)
        np.testing.assert_allclose(got[~np.isnan(got)], want[~np.isnan(want)], rtol=0, atol=2e-8)


def test_coverage_constants_with_N_runs():
    """tests/test_coverage.py:162-174: scaled=50 on the two MIBY contigs (28 N in one)."""
    texts = [read_fasta_bytes(GOLDEN / f) for f in ("MIBY01000005.fasta", "MIBY01000011.fasta")]
    md5s = [md5_hex(t) for t in texts]
    assert md5s[0].startswith("154173fb") and md5s[1].startswith("a0efc718")
    sk = [oracle.sketch_fasta_text(t, K, 50)[0] for t in texts]
    # a third input in the reference test is the concatenation of both files
    both, _ = oracle.sketch_fasta_text(texts[0] + texts[1], K, 50)
    sketches = [sk[0], sk[1], both]
    counts = oracle.pair_counts(sketches)
    sizes = [len(s) for s in sketches]
    ident, cov, null = oracle.ani(counts, sizes, sizes, K)
    vals = sorted({round(float(v), 10) for v in cov[~null].ravel()} - {1.0})
    assert 0.9622440235 in vals and 0.9884105907 in vals
    got_ident = {round(float(v), 10) for v in ident[~null].ravel()}
    assert got_ident == {1.0}
    assert null[0, 1] and null[1, 0]


def test_empty_short_and_lowercase():
    assert oracle.sketch_seq(b"", K, 1).size == 0
    assert oracle.sketch_seq(b"ACGT" * 7, K, 1).size == 0  # 28 < k
    seq = b"ACGTTGCAAGCTTGCATGCCTGCAGGTCGACTCTAGAGGATCCCCGGGTACCGAGCTCGAATTC"
    up = oracle.sketch_seq(seq, K, 1)
    lo = oracle.sketch_seq(seq.lower(), K, 1)
    assert up.size > 0 and np.array_equal(up, lo)
    # reverse complement gives the identical sketch (canonical k-mers)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    assert np.array_equal(up, oracle.sketch_seq(seq.translate(comp)[::-1], K, 1))
    # a window containing N is dropped, neighbours kept
    withn = seq[:40] + b"N" + seq[41:]
    a = oracle.sketch_seq(withn, K, 1)
    b = np.union1d(oracle.sketch_seq(seq[:40], K, 1), oracle.sketch_seq(seq[41:], K, 1))
    assert np.array_equal(a, b)


def test_fast_cpu_form_equals_naive_form():
    """The tuned scalar form timed as cpu_baseline must equal the pinned naive form."""
    for name, (scaled, genomes) in FIXTURE_SETS.items():
        if name == "bacterial_example":
            genomes = dict(list(genomes.items())[:1])
        for md5, fasta in genomes.items():
            text = read_fasta_bytes(GOLDEN / name / fasta)
            # one record per fixture file here except NC_002696 (2 records): sketch record-wise
            seqs = [b"".join(rec.split(b"\n")[1:]) for rec in text.split(b">")[1:]]
            slow = oracle.sketch_many(seqs, K, scaled, threads=2, fast=False)
            fast = oracle.sketch_many(seqs, K, scaled, threads=2, fast=True)
            for a, b in zip(slow, fast):
                assert np.array_equal(a, b)
    withn = b"ACGTTGCAAGCTTGCATGCCTGCAGGTCGACTCTAGNNAGGATCCCCGGGTACCGAGCTCGAATTCACTGGCCGTCGTTTTACAACGTCGTGACTGGGAAAACCCTGGCG"
    assert np.array_equal(oracle.sketch_many([withn], K, 1, fast=True)[0], oracle.sketch_seq(withn, K, 1))
