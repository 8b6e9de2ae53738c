"""CPU tests (gloo, world size 2 and 3) of the multi-GPU PRODUCT path: ``rundb.run_sourmash_hip(gpus=N)`` and
``rundb.run_fastani_hip(gpus=N)`` start real worker processes (``pyani_plus_amd.launch`` / ``.worker``) whose engine
is the oracle-backed stand-in (no GPU here); the databases must equal the single-process ones row for row.
Also: device selection from the environment, ``resume`` and ``export_run`` of the standalone driver."""

from __future__ import annotations

import gzip
import logging
import sqlite3
from pathlib import Path

import numpy as np
import pytest

from pyani_plus_amd import rundb
from pyani_plus_amd.methods import fastani_hip, sourmash_hip
from pyani_plus_amd.synth import arena_to_ascii, synth_arena_numpy
from tests.fake_engine import OracleEngine
from tests.helpers import FIXTURE_SETS, GOLDEN

LOGGER = logging.getLogger("test")
FACTORY = "tests.fake_engine:OracleEngine"


def _write_genomes(indir: Path, lengths, n_species=2):
    arena = synth_arena_numpy(len(lengths), lengths, n_species=n_species)
    indir.mkdir()
    for g in range(len(lengths)):
        seq = arena_to_ascii(arena, g)
        half = len(seq) // 2
        text = b">g%d first part\n" % g + seq[:half] + b"\n>g%d second\n" % g + seq[half:] + b"\n"
        if g % 3 == 1:
            (indir / f"genome_{g}.fna.gz").write_bytes(gzip.compress(text))
        else:
            (indir / f"genome_{g}.fasta").write_bytes(text)
    return arena


def _dump(db: Path):
    conn = sqlite3.connect(db)
    out = {
        "genomes": conn.execute("SELECT genome_hash, length, description FROM genomes ORDER BY 1").fetchall(),
        "comparisons": conn.execute("SELECT query_hash, subject_hash, identity, aln_length, sim_errors, cov_query FROM comparisons ORDER BY 1, 2").fetchall(),
        "runs": conn.execute("SELECT status, df_identity, df_cov_query, df_aln_length, df_sim_errors, df_hadamard FROM runs").fetchall(),
        "links": conn.execute("SELECT genome_hash, fasta_filename FROM runs_genomes ORDER BY 1").fetchall(),
    }
    conn.close()
    return out


@pytest.mark.parametrize("world", [2, 3, 8])  # 8: the node size of BASELINE configs[2]; ranks with one genome each
def test_sharded_sourmash_driver_equals_the_single_process_one(tmp_path, world, monkeypatch):
    monkeypatch.setenv("PYANI_HIP_DIST_BACKEND", "gloo")
    lengths = [30_000, 4_000, 52_000, 64, 21_000, 33_000, 9_000]
    _write_genomes(tmp_path / "in", lengths)
    one = rundb.run_sourmash_hip(tmp_path / "in", tmp_path / "one.sqlite", cache=tmp_path / "cache1", scaled=50, kmersize=21,
                                 engine=OracleEngine(), temp=tmp_path / "t1", ingest="direct")
    many = rundb.run_sourmash_hip(tmp_path / "in", tmp_path / "many.sqlite", cache=tmp_path / "cacheN", scaled=50, kmersize=21,
                                  temp=tmp_path / "tN", gpus=world, engine_factory=FACTORY)
    assert one.status == many.status == "Done"
    a, b = _dump(tmp_path / "one.sqlite"), _dump(tmp_path / "many.sqlite")
    assert a == b
    assert len(a["comparisons"]) == len(lengths) ** 2
    sigs1 = sorted((tmp_path / "cache1" / "sourmash_k=21_scaled=50").glob("*.sig"))
    sigsN = sorted((tmp_path / "cacheN" / "sourmash_k=21_scaled=50").glob("*.sig"))
    assert [p.name for p in sigs1] == [p.name for p in sigsN] and len(sigs1) == len(lengths)
    for p, q in zip(sigs1, sigsN):
        assert p.read_bytes() == q.read_bytes()
    # every rank reported, on the collective backend asked for, and the ranks' tiles cover all columns once
    import json

    results = [json.loads(p.read_text()) for p in sorted((tmp_path / "tN" / "sourmash-hip.workers").glob("result_rank*.json"))]
    ranks = min(world, len(lengths))  # never more workers than genomes
    assert len(results) == ranks and all(r["ok"] and r["backend"] == "gloo" for r in results)


def test_sharded_sourmash_driver_reports_duplicates_and_bad_files(tmp_path, monkeypatch):
    monkeypatch.setenv("PYANI_HIP_DIST_BACKEND", "gloo")
    _write_genomes(tmp_path / "in", [20_000, 8_000, 12_000, 5_000])
    (tmp_path / "in" / "copy_of_3.fasta").write_bytes((tmp_path / "in" / "genome_3.fasta").read_bytes())
    with pytest.raises(SystemExit, match="Multiple genomes with same MD5 checksum"):
        rundb.run_sourmash_hip(tmp_path / "in", tmp_path / "dup.sqlite", cache=tmp_path / "c", scaled=50, temp=tmp_path / "t", gpus=2,
                               engine_factory=FACTORY)
    (tmp_path / "in" / "copy_of_3.fasta").unlink()
    (tmp_path / "in" / "broken.fa.gz").write_bytes(b"this is not gzip")
    with pytest.raises(SystemExit, match="NOT gzip compressed"):
        rundb.run_sourmash_hip(tmp_path / "in", tmp_path / "bad.sqlite", cache=tmp_path / "c2", scaled=50, temp=tmp_path / "t2", gpus=2,
                               engine_factory=FACTORY)


def test_sharded_fastani_driver_equals_the_single_process_one(tmp_path, monkeypatch):
    monkeypatch.setenv("PYANI_HIP_DIST_BACKEND", "gloo")
    name = "viral_example"
    one = rundb.run_fastani_hip(GOLDEN / name, tmp_path / "one.sqlite", engine=OracleEngine(), temp=tmp_path / "t1")
    many = rundb.run_fastani_hip(GOLDEN / name, tmp_path / "many.sqlite", temp=tmp_path / "tN", gpus=2, engine_factory=FACTORY)
    assert one.status == many.status == "Done"
    a, b = _dump(tmp_path / "one.sqlite"), _dump(tmp_path / "many.sqlite")
    assert a == b and len(a["comparisons"]) == 9
    # the reference's matrices for this set: the digits the matrix files hold
    from tests.helpers import load_matrix_tsv

    labels, want = load_matrix_tsv(GOLDEN / name / "matrices" / "fastANI_identity.tsv")
    _labels, want_cov = load_matrix_tsv(GOLDEN / name / "matrices" / "fastANI_coverage.tsv")
    stems = {rundb.filename_stem(f): h for h, f in b["links"]}
    got = {(q, s): (i, c) for q, s, i, _a, _e, c in b["comparisons"]}
    for qi, ql in enumerate(labels):
        for si, sl in enumerate(labels):
            ident, cov = got[(stems[ql], stems[sl])]
            if np.isnan(want[qi, si]):
                assert ident is None
            else:
                assert abs(ident - want[qi, si]) <= 1e-12 and abs(cov - want_cov[qi, si]) <= 1e-12


def test_device_selection_from_the_environment(monkeypatch):
    import torch

    monkeypatch.delenv("PYANI_HIP_DEVICE", raising=False)
    monkeypatch.delenv("LOCAL_RANK", raising=False)
    assert sourmash_hip.resolve_device() == 0
    monkeypatch.setenv("PYANI_HIP_DEVICE", "5")
    assert sourmash_hip.resolve_device() == 5 and sourmash_hip.resolve_device(spread_key=2) == 5
    monkeypatch.setenv("PYANI_HIP_DEVICE", "gpu-one")
    with pytest.raises(ValueError, match="PYANI_HIP_DEVICE"):
        sourmash_hip.resolve_device()
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.setenv("PYANI_HIP_DEVICE", "spread")
    assert [sourmash_hip.resolve_device(spread_key=c) for c in (1, 2, 8, 9, 17)] == [1, 2, 0, 1, 1]
    monkeypatch.delenv("PYANI_HIP_DEVICE")
    monkeypatch.setenv("LOCAL_RANK", "3")
    assert sourmash_hip.resolve_device() == 3
    monkeypatch.setenv("LOCAL_RANK", "11")
    assert sourmash_hip.resolve_device() == 3  # more ranks than devices: they share (the gloo plumbing mode)


def test_resume_completes_a_partial_run_without_recomputing_finished_columns(tmp_path):
    name = "bacterial_example"
    scaled, genomes = FIXTURE_SETS[name]
    db = tmp_path / "run.sqlite"
    rundb.run_sourmash_hip(GOLDEN / name, db, cache=tmp_path / "cache", scaled=scaled, engine=OracleEngine(), temp=tmp_path / "t")
    full = _dump(db)
    hashes = sorted(genomes)
    conn = sqlite3.connect(db)
    conn.execute("DELETE FROM comparisons WHERE subject_hash IN (?, ?)", (hashes[1], hashes[3]))
    conn.execute("UPDATE runs SET status='Worker interrupted', df_identity=NULL, df_cov_query=NULL, df_aln_length=NULL, df_sim_errors=NULL, df_hadamard=NULL")
    conn.commit()
    conn.close()

    class Counting(OracleEngine):
        columns = []

        def pair_counts(self, sk, q_range=None, s_range=None, algo=0):
            Counting.columns.append(tuple(s_range) if s_range else None)
            return super().pair_counts(sk, q_range, s_range, algo)

    for ingest in ("json", "direct"):
        Counting.columns = []
        run = rundb.resume(db, cache=tmp_path / "cache", engine=Counting(), temp=tmp_path / f"t_{ingest}", ingest=ingest)
        assert run.status == "Done"
        assert _dump(db) == full
        if ingest == "json":  # the two missing columns, one call each; never the whole square
            assert sorted(Counting.columns) == [(1, 2), (3, 4)]
        else:
            assert Counting.columns == []  # complete already: nothing is computed
    with pytest.raises(SystemExit, match="has no run-id 7"):
        rundb.resume(db, run_id=7, engine=OracleEngine())
    with pytest.raises(SystemExit, match="does not exist"):
        rundb.resume(tmp_path / "nothing.sqlite")
    conn = sqlite3.connect(db)
    conn.execute("UPDATE configurations SET version='0.0.1'")
    conn.commit()
    conn.close()
    with pytest.raises(SystemExit, match=r"We have libpyani_hip version .*, but run-id 1 used libpyani_hip version 0.0.1 instead."):
        rundb.resume(db, engine=OracleEngine())


def test_resume_on_worker_processes_computes_the_missing_columns_only(tmp_path, monkeypatch):
    """``resume(..., gpus=2)`` of a ``sourmash-hip`` run goes through the executor the run itself uses -- worker processes,
    one all-gather -- and the ranks hand back the missing subject columns only (the reference re-runs the missing
    columns through the workflow they came from, /root/reference/pyani_plus/public_cli.py:243-261)."""
    monkeypatch.setenv("PYANI_HIP_DIST_BACKEND", "gloo")
    lengths = [30_000, 4_000, 52_000, 21_000, 33_000, 9_000]
    _write_genomes(tmp_path / "in", lengths)
    db = tmp_path / "run.sqlite"
    rundb.run_sourmash_hip(tmp_path / "in", db, cache=tmp_path / "cache", scaled=50, kmersize=21, engine=OracleEngine(), temp=tmp_path / "t")
    full = _dump(db)
    hashes = sorted(g[0] for g in full["genomes"])
    missing = [hashes[0], hashes[2], hashes[5]]
    conn = sqlite3.connect(db)
    conn.execute("DELETE FROM comparisons WHERE subject_hash IN (?, ?, ?)", missing)
    conn.execute("UPDATE runs SET status='Worker interrupted', df_identity=NULL, df_cov_query=NULL, df_aln_length=NULL, df_sim_errors=NULL, df_hadamard=NULL")
    conn.commit()
    conn.close()
    run = rundb.resume(db, cache=tmp_path / "cache", temp=tmp_path / "t2", gpus=2, engine_factory=FACTORY)
    assert run.status == "Done" and _dump(db) == full
    # what left the ranks: the three missing columns and nothing else
    from pyani_plus_amd import wire

    columns = []
    for tile in sorted((tmp_path / "t2" / "sourmash-hip.workers").glob("*.tile.npz")):
        _cfg, queries, subjects, *_ = wire.load_tile(tile)
        assert len(queries) == len(lengths)
        columns += list(subjects)
    assert sorted(columns) == sorted(missing)
    # a complete run is left alone, with or without workers
    assert rundb.resume(db, cache=tmp_path / "cache", temp=tmp_path / "t3", gpus=2, engine_factory=FACTORY).status == "Done"
    assert not (tmp_path / "t3" / "sourmash-hip.workers").exists()


def test_resume_of_a_partial_fastani_run(tmp_path):
    name = "viral_example"
    db = tmp_path / "f.sqlite"
    rundb.run_fastani_hip(GOLDEN / name, db, engine=OracleEngine(), temp=tmp_path / "t")
    full = _dump(db)
    first = full["comparisons"][0][1]
    conn = sqlite3.connect(db)
    conn.execute("DELETE FROM comparisons WHERE subject_hash = ? AND query_hash != ?", (sorted({c[1] for c in full["comparisons"]})[2], first))
    conn.execute("UPDATE runs SET status='Running', df_identity=NULL")
    conn.commit()
    conn.close()
    eng = OracleEngine()
    assert rundb.resume(db, engine=eng, temp=tmp_path / "t2").status == "Done"
    assert _dump(db) == full
    assert [c["ref_range"] for c in eng.fragani_calls] == [(2, 3)]  # the one incomplete column


@pytest.mark.parametrize("name", ["viral_example", "bacterial_example", "bad_alignments"])
def test_export_run_matches_the_reference_matrices(name, tmp_path):
    """``export_run`` against the reference's own matrices (compared the way the reference's tests compare an export
    with them, tests/test_public_cli.py:77-96, 986-990) and against the golden export made by the reference's code
    on the reference-made database (tests/golden/make_export_golden.py)."""
    import pandas as pd

    scaled, genomes = FIXTURE_SETS[name]
    db = tmp_path / "run.sqlite"
    rundb.run_sourmash_hip(GOLDEN / name, db, cache=tmp_path / "cache", scaled=scaled, engine=OracleEngine(), temp=tmp_path / "t")
    out = tmp_path / "export"
    written = rundb.export_run(db, out)
    method = sourmash_hip.METHOD
    assert sorted(p.name for p in written) == sorted(
        [f"{method}_run_1.tsv"] + [f"{method}_{k}.tsv" for k in ("identity", "aln_lengths", "sim_errors", "query_cov", "hadamard", "tANI")]
    )
    for ours, theirs in (("identity", "sourmash_identity.tsv"), ("query_cov", "sourmash_coverage.tsv")):
        want = pd.read_csv(GOLDEN / name / "matrices" / theirs, sep="\t", header=0, index_col=0).sort_index(axis=0).sort_index(axis=1)
        got = pd.read_csv(out / f"{method}_{ours}.tsv", sep="\t", header=0, index_col=0).sort_index(axis=0).sort_index(axis=1)
        pd.testing.assert_frame_equal(want, got)
    golden = GOLDEN / name / "export"
    for kind in ("identity", "aln_lengths", "sim_errors", "query_cov", "hadamard", "tANI"):
        assert (out / f"{method}_{kind}.tsv").read_bytes() == (golden / f"sourmash_{kind}.tsv").read_bytes(), kind
    ours = sorted((out / f"{method}_run_1.tsv").read_text().splitlines())
    theirs = sorted((golden / "sourmash_run_1.tsv").read_text().splitlines())
    assert ours == theirs  # row order follows insertion order, which differs between the backends' importers
    # md5 and filename labels
    rundb.export_run(db, tmp_path / "by_md5", label="md5")
    got = pd.read_csv(tmp_path / "by_md5" / f"{method}_identity.tsv", sep="\t", header=0, index_col=0)
    assert list(got.index) == sorted(genomes) == list(got.columns)
    rundb.export_run(db, tmp_path / "by_file", label="filename")
    got = pd.read_csv(tmp_path / "by_file" / f"{method}_identity.tsv", sep="\t", header=0, index_col=0)
    assert list(got.index) == sorted(genomes.values())


def test_export_run_of_a_partial_run_writes_the_long_form_only(tmp_path):
    name = "viral_example"
    scaled, _ = FIXTURE_SETS[name]
    db = tmp_path / "run.sqlite"
    rundb.run_sourmash_hip(GOLDEN / name, db, cache=tmp_path / "cache", scaled=scaled, engine=OracleEngine(), temp=tmp_path / "t")
    conn = sqlite3.connect(db)
    conn.execute("DELETE FROM comparisons WHERE comparison_id = 2")
    conn.commit()
    conn.close()
    with pytest.raises(SystemExit, match=r"run-id 1 has 8 of 3\^2=9 comparisons, 1 needed"):
        rundb.export_run(db, tmp_path / "out")
    assert [p.name for p in (tmp_path / "out").iterdir()] == ["sourmash-hip_run_1.tsv"]
    assert len((tmp_path / "out" / "sourmash-hip_run_1.tsv").read_text().splitlines()) == 9
    with pytest.raises(SystemExit, match="does not exist"):
        rundb.export_run(tmp_path / "no.sqlite", tmp_path / "out")


def test_fastani_worker_flushes_every_query_batch_and_keeps_them_on_interrupt(tmp_path):
    """The column file is rewritten after every batch of queries and an interrupt keeps the finished batches
    (pyani_plus/private_cli.py:1029-1110); the reference index is built by the first batch only."""
    import json

    from tests.test_host_logic import _make_run, _Session

    name = "viral_example"
    _scaled, genomes = FIXTURE_SETS[name]
    run = _make_run(GOLDEN / name, genomes, 300, method=fastani_hip.METHOD)
    tool = fastani_hip.get_fastani_hip()
    run.configuration.program, run.configuration.version = tool.exe_path.stem, tool.version
    run.configuration.fragsize, run.configuration.kmersize, run.configuration.minmatch = 3000, 16, 0.2
    hash_to_filename = dict(genomes)
    queries = {h: 1000 for h in hash_to_filename}

    class Interrupted(OracleEngine):
        def fragani(self, *args, **kwargs):
            if kwargs["query_range"][0] == 2:
                raise KeyboardInterrupt
            return super().fragani(*args, **kwargs)

    eng, session = Interrupted(), _Session()
    out = tmp_path / "f.json"
    assert fastani_hip.compute_fastani_hip(LOGGER, tmp_path, session, run, out, GOLDEN / name, hash_to_filename, {}, queries, "",
                                           engine=eng, query_batch=1) == 0
    assert run.status == "Worker interrupted"
    rows = json.loads(out.read_text())["comparisons"]
    assert len(rows) == 6 and {r["query_hash"] for r in rows} == set(sorted(genomes)[:2])  # two of three query batches
    assert [c["reuse_index"] for c in eng.fragani_calls] == [False, True]
    # uninterrupted, batch by batch, equals the one-batch result
    whole, batched = tmp_path / "whole.json", tmp_path / "batched.json"
    for path, qb in ((whole, 500), (batched, 2)):
        run.status = "Running"
        assert fastani_hip.compute_fastani_hip(LOGGER, tmp_path, _Session(), run, path, GOLDEN / name, hash_to_filename, {}, queries, "",
                                               engine=OracleEngine(), query_batch=qb) == 0
    assert json.loads(whole.read_text()) == json.loads(batched.read_text())


def test_fastani_query_batches_end_where_the_queries_leave_a_gap(tmp_path):
    """A subject that is not among the queries sits between them in the sorted genome list: a batch is mapped as a RANGE
    of genomes, so it ends at the gap instead of spanning the subject (whose rows would be computed and thrown away);
    and a one-column call gets that column only back from the engine (O(n) results, not n x n)."""
    import json

    from tests.test_host_logic import _make_run, _Session

    name = "viral_example"
    _scaled, genomes = FIXTURE_SETS[name]
    run = _make_run(GOLDEN / name, genomes, 300, method=fastani_hip.METHOD)
    tool = fastani_hip.get_fastani_hip()
    run.configuration.program, run.configuration.version = tool.exe_path.stem, tool.version
    run.configuration.fragsize, run.configuration.kmersize, run.configuration.minmatch = 3000, 16, 0.2
    hash_to_filename = dict(genomes)
    ordered = sorted(hash_to_filename)
    subject = ordered[1]  # the middle genome: the two queries are genomes 0 and 2
    queries = {h: 1000 for h in ordered if h != subject}
    eng = OracleEngine()
    out = tmp_path / "gap.json"
    assert fastani_hip.compute_fastani_hip(LOGGER, tmp_path, _Session(), run, out, GOLDEN / name, hash_to_filename, {}, queries, subject,
                                           engine=eng) == 0
    assert [c["query_range"] for c in eng.fragani_calls] == [(0, 1), (2, 3)]  # not (0, 3)
    assert all(c["ref_range"] == (1, 2) for c in eng.fragani_calls)
    rows = json.loads(out.read_text())["comparisons"]
    assert [(r["query_hash"], r["subject_hash"]) for r in rows] == [(ordered[0], subject), (ordered[2], subject)]
    # the same two rows as in the all-vs-all column file
    whole = tmp_path / "whole.json"
    assert fastani_hip.compute_fastani_hip(LOGGER, tmp_path, _Session(), run, whole, GOLDEN / name, hash_to_filename, {},
                                           {h: 1000 for h in ordered}, "", engine=OracleEngine()) == 0
    want = {(r["query_hash"], r["subject_hash"]): r for r in json.loads(whole.read_text())["comparisons"]}
    assert all(want[(r["query_hash"], r["subject_hash"])] == r for r in rows)


def test_fastani_rows_as_arrays_equal_the_row_by_row_form_and_json_dumps(tmp_path):
    """The worker's vectorised rows (``comparison_block``) against the per-row statement of the reference's mapping
    (``comparison_entry``, pyani_plus/private_cli.py:1066-1098), and the natively written six-key rows against
    ``json.dumps`` of those dicts, byte for byte."""
    import json

    from pyani_plus_amd import wire

    rng = np.random.default_rng(5)
    n = 7
    total = rng.integers(0, 40, size=n).astype(np.uint32)
    total[2] = 0
    matched = np.minimum(rng.integers(0, 45, size=(n, n)), total[:, None]).astype(np.uint32)
    matched[rng.random((n, n)) < 0.3] = 0
    ani = rng.uniform(74.0, 100.0, size=(n, n))
    ani[0, 1], ani[1, 0], ani[3, 3] = 100.0, 99.99995, 82.91245
    ident_sum = ani * matched
    lengths = rng.integers(30_000, 200_000, size=n)
    rows, cols = [0, 1, 3, 4, 6], [1, 2, 3, 5]
    ident, aln, sim, cov, null = fastani_hip.comparison_block(total, matched, ident_sum, lengths, rows, cols, 3000, 0.2)
    hashes = [f"{i:032x}" for i in range(n)]
    entries = []
    for a, qi in enumerate(rows):
        for b, si in enumerate(cols):
            m = int(matched[qi, si])
            want = fastani_hip.comparison_entry(hashes[qi], hashes[si], int(total[qi]), m, ident_sum[qi, si] / m if m else float("nan"), 3000, 0.2,
                                                int(lengths[qi]), int(lengths[si]), {})
            entries.append(want)
            if want["identity"] is None:
                assert null[a, b]
            else:
                assert not null[a, b] and ident[a, b] == want["identity"] and cov[a, b] == want["cov_query"]
                assert aln[a, b] == want["aln_length"] and sim[a, b] == want["sim_errors"]
    assert null.any() and not null.all()
    cfg = rundb.Configuration(1, fastani_hip.METHOD, "libpyani_hip", "0", fragsize=3000, kmersize=16, minmatch=0.2)
    writer = wire.ColumnFileWriter(LOGGER, tmp_path / "native.json", cfg)
    writer.append([hashes[i] for i in rows[:2]], [hashes[i] for i in cols], ident[:2], cov[:2], null[:2], aln_length=aln[:2], sim_errors=sim[:2])
    writer.append([hashes[i] for i in rows[2:]], [hashes[i] for i in cols], ident[2:], cov[2:], null[2:], aln_length=aln[2:], sim_errors=sim[2:])
    wire.export_json_db_entries(LOGGER, tmp_path / "python.json", cfg, entries)
    assert (tmp_path / "native.json").read_bytes() == (tmp_path / "python.json").read_bytes()
    assert json.loads((tmp_path / "native.json").read_text())["comparisons"] == entries


@pytest.mark.parametrize("gpus", [1, 2])
def test_fastani_direct_ingest_equals_the_json_route(tmp_path, gpus, monkeypatch):
    monkeypatch.setenv("PYANI_HIP_DIST_BACKEND", "gloo")
    name = "viral_example"
    rundb.run_fastani_hip(GOLDEN / name, tmp_path / "json.sqlite", engine=OracleEngine(), temp=tmp_path / "t1")
    kwargs = {"engine": OracleEngine()} if gpus == 1 else {"gpus": gpus, "engine_factory": FACTORY}
    run = rundb.run_fastani_hip(GOLDEN / name, tmp_path / "direct.sqlite", temp=tmp_path / "t2", ingest="direct", **kwargs)
    assert run.status == "Done"
    assert _dump(tmp_path / "json.sqlite") == _dump(tmp_path / "direct.sqlite")
    # a resumed run takes the direct route block by block
    conn = sqlite3.connect(tmp_path / "direct.sqlite")
    victim = sorted({r[1] for r in conn.execute("SELECT query_hash, subject_hash FROM comparisons")})[1]
    conn.execute("DELETE FROM comparisons WHERE subject_hash = ?", (victim,))
    conn.execute("UPDATE runs SET status='Running', df_identity=NULL, df_aln_length=NULL")
    conn.commit()
    conn.close()
    assert rundb.resume(tmp_path / "direct.sqlite", temp=tmp_path / "t3", ingest="direct", **kwargs).status == "Done"
    assert _dump(tmp_path / "json.sqlite") == _dump(tmp_path / "direct.sqlite")
