"""The method module against the reference's OWN objects (CPU, this container only).

Everywhere else the plugin is driven through duck-typed stand-ins (``rundb.Run``, ``rundb.Session``).  Here the
real thing is imported from ``/root/reference``: ``private_cli.log_run`` creates the database through SQLAlchemy,
``db_orm.load_run`` returns the ORM ``Run`` the worker would hand to ``prepare_genomes`` / ``compute_sourmash_hip``
(pyani_plus/private_cli.py:725-752, 956-968), and the reference's ``import_json_comparisons`` +
``Run.cache_comparisons`` read what the plugin wrote.  The other direction too: a database written by
``rundb.run_sourmash_hip`` is opened with the reference's ORM.

The reference does not travel to the GPU box, so these tests skip themselves wherever ``/root/reference`` (or
SQLAlchemy) is missing; the device is replaced by the oracle-backed test engine -- what is under test is the
boundary, not the arithmetic.
"""

from __future__ import annotations

import datetime
import json
import logging
import sys
from pathlib import Path

import numpy as np
import pytest

from pyani_plus_amd import rundb
from pyani_plus_amd.methods import sourmash_hip
from tests.fake_engine import OracleEngine
from tests.helpers import FIXTURE_SETS, GOLDEN, load_matrix_tsv

REFERENCE = Path("/root/reference")
LOGGER = logging.getLogger("interop")

pytestmark = pytest.mark.skipif(not (REFERENCE / "pyani_plus" / "db_orm.py").is_file(), reason="the reference checkout is not here")


@pytest.fixture(scope="module")
def reference():
    """``(db_orm, private_cli)`` of the reference, imported without leaving bytecode in its tree."""
    pytest.importorskip("sqlalchemy")
    old_flag, old_path = sys.dont_write_bytecode, list(sys.path)
    sys.dont_write_bytecode = True
    if not hasattr(datetime, "UTC"):
        datetime.UTC = datetime.timezone.utc  # the reference wants Python >= 3.11 (db_orm.add_run)
    sys.path.insert(0, str(REFERENCE))
    try:
        from pyani_plus import db_orm, private_cli
    except ImportError as err:  # a dependency of the reference this image lacks
        pytest.skip(f"the reference does not import here: {err}")
    finally:
        sys.path[:] = old_path
        sys.dont_write_bytecode = old_flag
    return db_orm, private_cli


def _stem(fasta_filename: str) -> str:
    name = fasta_filename[:-3] if fasta_filename.endswith(".gz") else fasta_filename
    return name.rsplit(".", 1)[0]


def _golden_matrix(name: str, which: str, stems: list[str]) -> np.ndarray:
    """The reference's own ``matrices/sourmash_{identity,coverage}.tsv`` (labels are file stems), re-ordered."""
    labels, values = load_matrix_tsv(GOLDEN / name / "matrices" / f"sourmash_{which}.tsv")
    order = [labels.index(x) for x in stems]
    return values[np.ix_(order, order)]


@pytest.mark.parametrize("name", sorted(FIXTURE_SETS))
def test_plugin_serves_the_references_orm_run(reference, name, tmp_path):
    """log_run (reference) -> prepare_genomes + compute_sourmash_hip (this package, given the ORM objects) ->
    import_json_comparisons + cache_comparisons (reference): complete run, matrices equal the reference's fixtures."""
    db_orm, private_cli = reference
    scaled, _genomes = FIXTURE_SETS[name]
    fasta_dir = GOLDEN / name
    database = tmp_path / "reference.sqlite"
    tool = sourmash_hip.get_sourmash_hip()
    private_cli.log_run(
        fasta=fasta_dir, database=database, cmdline="pyani-plus sourmash-hip ...", status="Initialising",
        name=f"interop {name}", method=sourmash_hip.METHOD, program=tool.exe_path.stem, version=tool.version,
        kmersize=31, extra=f"scaled={scaled}", create_db=True,
    )  # fmt: skip
    cache = tmp_path / "cache"
    cache.mkdir()
    engine = OracleEngine()
    with db_orm.connect_to_db(LOGGER, database) as session:
        run = db_orm.load_run(session, run_id=1)
        n = run.genomes.count()
        # what private_cli.prepare does with the module (private_cli.py:746-752)
        assert len(list(sourmash_hip.prepare_genomes(LOGGER, run, cache, engine=engine))) == n
        assert len(list((cache / f"sourmash_k=31_scaled={scaled}").glob("*.sig"))) == n
        # what private_cli.compute_column builds before calling compute[method] (private_cli.py:880-905, 956-968)
        filename_to_hash = {a.fasta_filename: a.genome_hash for a in run.fasta_hashes}
        hash_to_filename = {h: f for f, h in filename_to_hash.items()}
        query_hashes = {a.genome_hash: a.genome.length for a in run.fasta_hashes}
        json_file = tmp_path / f"{sourmash_hip.METHOD}.run_1.column_0.json"
        rc = sourmash_hip.compute_sourmash_hip(
            LOGGER, tmp_path, session, run, json_file, fasta_dir, hash_to_filename, filename_to_hash, query_hashes, "",
            cache=cache, engine=engine,
        )  # fmt: skip
        assert rc == 0
        assert run.comparisons().count() == 0  # the worker never writes comparisons itself
        # the parent's side of the wire (workflows/__init__.py:75-87)
        private_cli.import_json_comparisons(LOGGER, session, json_file)
        assert run.comparisons().count() == n * n  # the reference's completion test (public_cli.py:223-226)
        run.cache_comparisons()
        session.commit()
        hashes = sorted(query_hashes)
        identity, coverage = run.identities, run.cov_query
        assert list(identity.index) == hashes == list(identity.columns)
        stems = [_stem(hash_to_filename[h]) for h in hashes]
        np.testing.assert_allclose(identity.to_numpy(dtype=float), _golden_matrix(name, "identity", stems), rtol=0, atol=2e-8, equal_nan=True)
        np.testing.assert_allclose(coverage.to_numpy(dtype=float), _golden_matrix(name, "coverage", stems), rtol=0, atol=2e-8, equal_nan=True)
        # and bit for bit what the reference itself produced from its manysearch.csv (tests/golden/make_boundary_golden.py)
        boundary = json.loads((GOLDEN / name / "boundary.json").read_text())
        assert run.df_identity == boundary["df_identity"] and run.df_cov_query == boundary["df_cov_query"]
        assert run.df_hadamard == boundary["df_hadamard"]
        config = run.configuration
        assert (config.method, config.program, config.version) == (sourmash_hip.METHOD, tool.exe_path.stem, tool.version)


def test_single_column_worker_with_the_references_run(reference, tmp_path):
    """The unpatched reference calls the worker once per subject column (private_cli.py:855-860): every column file
    imports, and together they complete the run with the same matrices as the all-columns call."""
    db_orm, private_cli = reference
    name = "viral_example"
    scaled, _genomes = FIXTURE_SETS[name]
    database = tmp_path / "reference.sqlite"
    tool = sourmash_hip.get_sourmash_hip()
    private_cli.log_run(
        fasta=GOLDEN / name, database=database, cmdline="x", status="Initialising", name="columns",
        method=sourmash_hip.METHOD, program=tool.exe_path.stem, version=tool.version, kmersize=31,
        extra=f"scaled={scaled}", create_db=True,
    )  # fmt: skip
    cache = tmp_path / "cache"
    cache.mkdir()
    engine = OracleEngine()
    with db_orm.connect_to_db(LOGGER, database) as session:
        run = db_orm.load_run(session, run_id=1)
        list(sourmash_hip.prepare_genomes(LOGGER, run, cache, engine=engine))
        query_hashes = {a.genome_hash: a.genome.length for a in run.fasta_hashes}
        hashes = sorted(query_hashes)
        for column, subject in enumerate(hashes, start=1):
            json_file = tmp_path / f"column_{column}.json"
            assert sourmash_hip.compute_sourmash_hip(LOGGER, tmp_path, session, run, json_file, GOLDEN / name, {}, {}, query_hashes,
                                                     subject, cache=cache, engine=engine) == 0
            column_rows = json.loads(json_file.read_text())["comparisons"]
            assert [r["subject_hash"] for r in column_rows] == [subject] * len(hashes)
            private_cli.import_json_comparisons(LOGGER, session, json_file)
            assert run.comparisons().count() == column * len(hashes)
        run.cache_comparisons()
        boundary = json.loads((GOLDEN / name / "boundary.json").read_text())
        assert run.df_identity == boundary["df_identity"] and run.df_cov_query == boundary["df_cov_query"]


def test_fastani_plugin_with_the_references_run(reference, tmp_path):
    """fastANI-hip through the reference's objects: log_run with fragsize / kmersize / minmatch, the column worker
    given the ORM run, the reference's importer and cache -- identity, aln_length, sim_errors, cov_query and hadamard
    against the reference's own fastANI matrices of the viral fixture: the digits the matrix files hold
    (tests/test_fragani_oracle.py: every fastANI row is reproduced exactly)."""
    from pyani_plus_amd.methods import fastani_hip
    from tests.test_fragani_oracle import ANI_TOL, MATCHED_TOL

    db_orm, private_cli = reference
    name = "viral_example"
    database = tmp_path / "reference.sqlite"
    tool = fastani_hip.get_fastani_hip()
    private_cli.log_run(
        fasta=GOLDEN / name, database=database, cmdline="pyani-plus fastANI-hip ...", status="Initialising", name="fastani",
        method=fastani_hip.METHOD, program=tool.exe_path.stem, version=tool.version, fragsize=fastani_hip.FRAG_LEN,
        kmersize=fastani_hip.KMER_SIZE, minmatch=fastani_hip.MIN_FRACTION, create_db=True,
    )  # fmt: skip
    with db_orm.connect_to_db(LOGGER, database) as session:
        run = db_orm.load_run(session, run_id=1)
        filename_to_hash = {a.fasta_filename: a.genome_hash for a in run.fasta_hashes}
        hash_to_filename = {h: f for f, h in filename_to_hash.items()}
        query_hashes = {a.genome_hash: a.genome.length for a in run.fasta_hashes}
        hashes = sorted(query_hashes)
        json_file = tmp_path / "fastani.json"
        assert fastani_hip.compute_fastani_hip(LOGGER, tmp_path, session, run, json_file, GOLDEN / name, hash_to_filename,
                                               filename_to_hash, query_hashes, "", engine=OracleEngine()) == 0
        private_cli.import_json_comparisons(LOGGER, session, json_file)
        assert run.comparisons().count() == len(hashes) ** 2
        run.cache_comparisons()
        stems = [_stem(hash_to_filename[h]) for h in hashes]

        def golden(which: str) -> np.ndarray:
            labels, values = load_matrix_tsv(GOLDEN / name / "matrices" / f"fastANI_{which}.tsv")
            order = [labels.index(x) for x in stems]
            return values[np.ix_(order, order)]

        identity, want = run.identities.to_numpy(dtype=float), golden("identity")
        assert np.array_equal(np.isnan(identity), np.isnan(want))  # the same pairs are reported
        np.testing.assert_allclose(identity, want, rtol=0, atol=ANI_TOL / 100 + 1e-12, equal_nan=True)
        frags = {h: (query_hashes[h] // fastani_hip.FRAG_LEN) for h in hashes}
        slack = np.array([[max(1.0, MATCHED_TOL * frags[q]) for _s in hashes] for q in hashes])
        aln, want_aln = run.aln_length.to_numpy(dtype=float), golden("aln_lengths")
        assert np.all(np.abs(np.nan_to_num(aln - want_aln)) <= slack * fastani_hip.FRAG_LEN)
        errs, want_errs = run.sim_errors.to_numpy(dtype=float), golden("sim_errors")
        assert np.all(np.abs(np.nan_to_num(errs - want_errs)) <= slack)
        cov, want_cov = run.cov_query.to_numpy(dtype=float), golden("coverage")
        np.testing.assert_allclose(cov, want_cov, rtol=0, atol=max(0.01, MATCHED_TOL) + 1e-9, equal_nan=True)


def test_interrupted_worker_marks_the_references_run(reference, tmp_path, monkeypatch):
    """KeyboardInterrupt inside the comparison: ``run.status`` of the ORM object is set and committed through the
    SQLAlchemy session, the column file stays a complete document, return code 0 (private_cli.py:1889-1902)."""
    db_orm, private_cli = reference
    name = "viral_example"
    scaled, _genomes = FIXTURE_SETS[name]
    database = tmp_path / "reference.sqlite"
    tool = sourmash_hip.get_sourmash_hip()
    private_cli.log_run(
        fasta=GOLDEN / name, database=database, cmdline="x", status="Initialising", name="interrupted",
        method=sourmash_hip.METHOD, program=tool.exe_path.stem, version=tool.version, kmersize=31,
        extra=f"scaled={scaled}", create_db=True,
    )  # fmt: skip
    cache = tmp_path / "cache"
    cache.mkdir()
    engine = OracleEngine()
    with db_orm.connect_to_db(LOGGER, database) as session:
        run = db_orm.load_run(session, run_id=1)
        list(sourmash_hip.prepare_genomes(LOGGER, run, cache, engine=engine))
        query_hashes = {a.genome_hash: a.genome.length for a in run.fasta_hashes}
        calls = {"n": 0}
        real = engine.pair_counts

        def interrupted_after_first_tile(*args, **kwargs):
            calls["n"] += 1
            if calls["n"] > 1:
                raise KeyboardInterrupt
            return real(*args, **kwargs)

        monkeypatch.setattr(engine, "pair_counts", interrupted_after_first_tile, raising=False)
        json_file = tmp_path / "column.json"
        rc = sourmash_hip.compute_sourmash_hip(
            LOGGER, tmp_path, session, run, json_file, GOLDEN / name, {}, {}, query_hashes, "", cache=cache, engine=engine,
            tile_columns=1,
        )  # fmt: skip
        assert rc == 0
    with db_orm.connect_to_db(LOGGER, database) as session:
        run = db_orm.load_run(session, run_id=1)
        assert run.status == "Worker interrupted"
        private_cli.import_json_comparisons(LOGGER, session, json_file)  # the finished tile is importable
        assert run.comparisons().count() == len(query_hashes)  # one subject column of the three


@pytest.mark.parametrize("ingest", ["json", "direct"])
def test_reference_orm_reads_a_database_written_here(reference, ingest, tmp_path):
    """rundb.run_sourmash_hip (stdlib sqlite3, hand-written DDL) -> the reference's ORM: the run loads, is complete,
    its cached matrices parse, and recomputing the cache with the reference's code gives the same strings."""
    db_orm, _private_cli = reference
    name = "bacterial_example"
    scaled, _genomes = FIXTURE_SETS[name]
    database = tmp_path / "ours.sqlite"
    rundb.run_sourmash_hip(GOLDEN / name, database, cache=tmp_path / "cache", scaled=scaled, engine=OracleEngine(), temp=tmp_path, ingest=ingest)
    with db_orm.connect_to_db(LOGGER, database) as session:
        run = db_orm.load_run(session, run_id=1, check_complete=True)
        n = run.genomes.count()
        assert run.comparisons().count() == n * n and run.status == "Done"
        by_hash = {a.genome_hash: a.fasta_filename for a in run.fasta_hashes}
        stems = [_stem(by_hash[h]) for h in sorted(by_hash)]
        np.testing.assert_allclose(run.identities.to_numpy(dtype=float), _golden_matrix(name, "identity", stems), rtol=0, atol=2e-8)
        before = (run.df_identity, run.df_cov_query, run.df_aln_length, run.df_sim_errors, run.df_hadamard)
        run.df_identity = None  # "not cached yet" (db_orm.py:393-405)
        run.cache_comparisons()
        assert (run.df_identity, run.df_cov_query, run.df_aln_length, run.df_sim_errors, run.df_hadamard) == before
        genome = run.fasta_hashes[0].genome
        assert genome.length > 0 and genome.description


@pytest.mark.parametrize("name", ["viral_example", "bad_alignments"])
def test_reference_worker_commands_with_the_integration_edits(reference, name, tmp_path):
    """The edits of INTEGRATION.md applied to a scratch copy of the reference (outside this repository), then the
    reference's own `prepare-genomes` and `compute-column --subject 0` commands run the method: module found by name,
    tool version checked, all-columns mode accepted, column file imported, run complete, matrices as the reference's."""
    import subprocess

    scaled, genomes = FIXTURE_SETS[name]
    root = Path(__file__).resolve().parent.parent
    with_fastani = ["fastani"] if name == "viral_example" else []  # the reference has fastANI matrices for this set
    done = subprocess.run([sys.executable, str(root / "tests" / "tools" / "patched_reference_worker.py"), str(tmp_path), str(GOLDEN / name), str(scaled),
                           *with_fastani], capture_output=True, text=True, timeout=600, cwd=root)
    assert done.returncode == 0, done.stdout[-3000:] + done.stderr[-3000:]
    result = json.loads(done.stdout.strip().splitlines()[-1])
    n = len(genomes)
    assert result["compute_column"] == 0 and result["comparisons"] == n * n and result["genomes"] == n == result["signatures"]
    assert (result["method"], result["program"]) == ("sourmash-hip", "libpyani_hip")
    boundary = json.loads((GOLDEN / name / "boundary.json").read_text())
    for key in ("df_identity", "df_cov_query", "df_hadamard"):
        assert result[key] == boundary[key], key
    if with_fastani:
        # fastANI-hip, one `compute-column --subject <column>` call per subject as the unpatched scheduler issues them
        import io

        import pandas as pd

        from tests.test_fragani_oracle import ANI_TOL

        fast = result["fastani"]
        assert fast["codes"] == [0] * n and fast["comparisons"] == n * n
        identity = pd.read_json(io.StringIO(fast["df_identity"]), orient="split", dtype=float)
        labels, want = load_matrix_tsv(GOLDEN / name / "matrices" / "fastANI_identity.tsv")
        by_hash = {g["genome_hash"]: _stem(g["fasta_filename"]) for g in boundary["genomes"]}
        order = [labels.index(by_hash[h]) for h in identity.index]
        np.testing.assert_allclose(identity.to_numpy(dtype=float), want[np.ix_(order, order)], rtol=0, atol=ANI_TOL / 100 + 1e-12, equal_nan=True)
