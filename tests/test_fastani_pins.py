"""The fastANI values the reference's own tests hold beyond the 25 ``.fastani`` rows, asserted through the product
path (``rundb.run_fastani_hip`` -> ``compute_fastani_hip`` -> JSON column -> database -> cached matrices):

* ``tests/test_self_vs_self.py:121-122`` -- MIBY01000011 against itself: identity ``0.999953`` (a non-100 % self hit);
* ``tests/test_coverage.py:143-160``     -- small / both / large contig files with ``kmersize=15, fragsize=2000,
  minmatch=0.15``: the cached identity and query-coverage matrices;
* ``tests/fixtures/bacterial_example/matrices/fastANI_{aln_lengths,sim_errors,hadamard}.tsv`` -- the proxy columns
  (``aln_length = fragsize * matched``, ``sim_errors = fragments - matched``) and identity x coverage.

fastANI itself is a third-party binary whose source is not in the reference tree; with the exact slide of round 4
 the MIBY pins are reproduced EXACTLY -- identities equal to the reference's constants, coverage
fractions, fragment totals and the NULL pattern too --; and with the last three choices settled (position of a window,
the last of equally good candidates, float sums) the bacterial proxy matrices are exact as well: every aln_length, every
sim_errors, identity x coverage to the digits the matrix file holds.

Each check runs twice: on the CPU with the oracle-backed stand-in engine, and (``-m gpu``) on the device.
"""

from __future__ import annotations

import json
import sqlite3

import numpy as np
import pytest

from pyani_plus_amd import rundb
from tests.helpers import GOLDEN, load_matrix_tsv

PIN_TOL = 0.0  # the stored identity is the reference's own double, 0.01 * float(the text fastANI printed): pyani_plus/methods/fastani.py:113
SMALL, BOTH, LARGE = "154173fb8e7415ab45532a738572f957", "7b6a6226ce00e52edca15565aa0d270d", "a0efc718e680e34d2f5c8f5d2286ca9c"


@pytest.fixture(params=["oracle", pytest.param("hip", marks=pytest.mark.gpu)])
def engine(request):
    if request.param == "oracle":
        from tests.fake_engine import OracleEngine

        yield OracleEngine()
    else:
        from pyani_plus_amd.engine import HipEngine

        eng = HipEngine(0)
        yield eng
        eng.close()


def _matrices(db):
    conn = sqlite3.connect(db)
    row = conn.execute("SELECT status, df_identity, df_cov_query, df_aln_length, df_sim_errors, df_hadamard FROM runs").fetchone()
    conn.close()
    assert row[0] == "Done"
    return [json.loads(x) for x in row[1:]]


def test_miby_large_contig_against_itself(engine, tmp_path):
    """/root/reference/tests/test_self_vs_self.py:121-122: ``comp.identity == 0.999953``."""
    indir = tmp_path / "fasta"
    indir.mkdir()
    (indir / "MIBY01000011.fasta").write_bytes((GOLDEN / "MIBY01000011.fasta").read_bytes())
    rundb.run_fastani_hip(indir, tmp_path / "self.sqlite", engine=engine, temp=tmp_path / "t")
    ident, cov, aln, err, _had = _matrices(tmp_path / "self.sqlite")
    assert ident["index"] == ident["columns"] == [LARGE]
    assert ident["data"][0][0] == 0.999953  # an `==` in the reference too; the point of the pin: not a perfect self hit
    assert cov["data"] == [[1.0]] and aln["data"] == [[18000.0]] and err["data"] == [[0.0]]  # 6 of 6 fragments of 3000


def test_miby_small_contig_against_itself(engine, tmp_path):
    """/root/reference/tests/test_self_vs_self.py:90-91: ``comp.identity == 1.0`` for MIBY01000005 (7 582 bp, 28 N) with
    fastANI at its defaults."""
    indir = tmp_path / "fasta"
    indir.mkdir()
    (indir / "MIBY01000005.fasta").write_bytes((GOLDEN / "MIBY01000005.fasta").read_bytes())
    rundb.run_fastani_hip(indir, tmp_path / "self.sqlite", engine=engine, temp=tmp_path / "t")
    ident, cov, aln, err, _had = _matrices(tmp_path / "self.sqlite")
    assert ident["index"] == ident["columns"] == [SMALL]
    assert ident["data"] == [[1.0]] and cov["data"] == [[1.0]] and aln["data"] == [[6000.0]] and err["data"] == [[0.0]]


def test_miby_coverage_matrices_with_non_default_settings(engine, tmp_path):
    """/root/reference/tests/test_coverage.py:143-160: kmersize 15, fragsize 2000, minmatch 0.15."""
    indir = tmp_path / "fasta"
    indir.mkdir()
    small, large = (GOLDEN / "MIBY01000005.fasta").read_bytes(), (GOLDEN / "MIBY01000011.fasta").read_bytes()
    (indir / "small.fasta").write_bytes(small)
    (indir / "large.fasta").write_bytes(large)
    (indir / "both.fasta").write_bytes(small + large)
    rundb.run_fastani_hip(indir, tmp_path / "cov.sqlite", kmersize=15, fragsize=2000, minmatch=0.15, engine=engine, temp=tmp_path / "t")
    ident, cov, _aln, _err, _had = _matrices(tmp_path / "cov.sqlite")
    assert ident["index"] == ident["columns"] == [SMALL, BOTH, LARGE]
    want_ident = [[1.0, 1.0, None], [1.0, 0.99997, 0.999959], [None, 0.999959, 0.999959]]
    want_cov = [[1.0, 1.0, None], [0.25, 1.0, 0.75], [None, 1.0, 1.0]]
    for q in range(3):
        for s in range(3):
            if want_ident[q][s] is None:
                assert ident["data"][q][s] is None and cov["data"][q][s] is None  # the NULL pattern, exactly
            else:
                assert ident["data"][q][s] == want_ident[q][s], (q, s, ident["data"][q][s])
                assert cov["data"][q][s] == want_cov[q][s], (q, s)  # 3/12, 9/12, 12/12: exact


def _bacterial_column_checks(rows, labels, want, stems, columns, total_frags):
    aln, err, had = want
    for (q, s), e in rows.items():
        qi, si = labels.index(stems[q]), labels.index(stems[s])
        if si not in columns:
            continue
        t = total_frags[stems[q]]
        assert e["aln_length"] % 3000 == 0
        assert e["aln_length"] == aln[qi, si], (stems[q], stems[s], e["aln_length"], aln[qi, si])
        assert e["sim_errors"] == err[qi, si], (stems[q], stems[s])
        assert e["aln_length"] // 3000 + e["sim_errors"] == t  # matched + unmatched = all fragments, exactly
        assert abs(e["identity"] * e["cov_query"] - had[qi, si]) <= 1e-9, (stems[q], stems[s], e["identity"] * e["cov_query"], had[qi, si])


def test_bacterial_proxy_matrices(engine, tmp_path):
    """/root/reference/tests/fixtures/bacterial_example/matrices/fastANI_{aln_lengths,sim_errors,hadamard}.tsv.
    On the device: all 16 pairs through the run driver.  With the oracle (about 5 s per pair on one core): the
    column of NC_011916 through the column worker, as the reference computes one subject column per process."""
    import logging

    from pyani_plus_amd.methods import fastani_hip
    from tests.helpers import FIXTURE_SETS

    name = "bacterial_example"
    labels, aln = load_matrix_tsv(GOLDEN / name / "matrices" / "fastANI_aln_lengths.tsv")
    _l, err = load_matrix_tsv(GOLDEN / name / "matrices" / "fastANI_sim_errors.tsv")
    _l, had = load_matrix_tsv(GOLDEN / name / "matrices" / "fastANI_hadamard.tsv")
    genomes = FIXTURE_SETS[name][1]
    stems = {h: rundb.filename_stem(f) for h, f in genomes.items()}
    total_frags = {"NC_002696": 1338, "NC_010338": 1825, "NC_011916": 1347, "NC_014100": 1551}  # last column of the .fastani rows
    if hasattr(engine, "torch"):  # the device
        rundb.run_fastani_hip(GOLDEN / name, tmp_path / "b.sqlite", engine=engine, temp=tmp_path / "t")
        conn = sqlite3.connect(tmp_path / "b.sqlite")
        rows = {
            (q, s): {"identity": i, "aln_length": a, "sim_errors": e, "cov_query": c}
            for q, s, i, a, e, c in conn.execute("SELECT query_hash, subject_hash, identity, aln_length, sim_errors, cov_query FROM comparisons")
        }
        conn.close()
        assert len(rows) == 16
        _bacterial_column_checks(rows, labels, (aln, err, had), stems, set(range(4)), total_frags)
        m_aln, m_err = _matrices(tmp_path / "b.sqlite")[2:4]
        assert m_aln["index"] == m_aln["columns"] == sorted(genomes)
        assert np.asarray(m_aln["data"]).shape == (4, 4) and np.asarray(m_err["data"]).min() >= 0
        return
    tool = fastani_hip.get_fastani_hip()
    cfg = rundb.Configuration(1, fastani_hip.METHOD, tool.exe_path.stem, tool.version, fragsize=3000, kmersize=16, minmatch=0.2)
    run = rundb.Run(1, cfg, str(GOLDEN / name), [], "Testing")
    subject = next(h for h, s in stems.items() if s == "NC_011916")
    out = tmp_path / "col.json"

    class _S:
        def commit(self):
            pass

    assert fastani_hip.compute_fastani_hip(logging.getLogger("t"), tmp_path, _S(), run, out, GOLDEN / name, dict(genomes), {},
                                           {h: 1 for h in genomes}, subject, engine=engine) == 0
    rows = {(e["query_hash"], e["subject_hash"]): e for e in json.loads(out.read_text())["comparisons"]}
    assert len(rows) == 4
    _bacterial_column_checks(rows, labels, (aln, err, had), stems, {labels.index("NC_011916")}, total_frags)


def test_identity_is_the_reference_double_for_every_value_its_fixtures_hold():
    """/root/reference/pyani_plus/methods/fastani.py:113 stores ``0.01 * float(text)``.  ``x / 100.0`` is another double for
    four of the distinct identities the reference's 25 ``.fastani`` rows hold (85.9835, 99.4912, 99.9386, 99.9946): the
    method must produce the product, bit for bit, in its scalar and in its array form."""
    from pyani_plus_amd.methods import fastani_hip

    texts = set()
    for path in sorted(GOLDEN.glob("*/fastANI/all_vs_*.fastani")):
        for line in path.read_text().splitlines():
            if line.strip():
                texts.add(line.split("\t")[2])
    texts |= {"99.9953", "99.997", "99.9959"}  # the pins of tests/test_self_vs_self.py:121-122 and tests/test_coverage.py:143-160
    assert len(texts) == 22
    differing = 0
    for text in sorted(texts):
        x = float(text)
        want = 0.01 * x
        differing += want != x / 100.0
        entry = fastani_hip.comparison_entry("q", "s", 10, 10, x, 3000, 0.2, 30000, 30000, {})
        assert entry["identity"] == want, text
        total, matched = np.array([10], dtype=np.uint32), np.array([[10]], dtype=np.uint32)
        # the float sum whose float mean prints as `text`: ten fragments of that identity
        ident_sum = np.array([[float(np.float32(x) * np.float32(10))]])
        ident, *_rest = fastani_hip.comparison_block(total, matched, ident_sum, np.array([30000]), [0], [0], 3000, 0.2)
        if fastani_hip.fastani_print_round(float(fastani_hip.fastani_mean(ident_sum, matched)[0, 0])) == x:
            assert ident[0, 0] == want, text
    assert differing >= 4  # the values on which the two expressions part
    assert PIN_TOL == 0.0
