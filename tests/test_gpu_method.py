"""GPU: the sourmash-hip plugin and run driver against the reference-generated boundary vectors."""

from __future__ import annotations

import json
import logging
import sqlite3
from pathlib import Path

import pytest

from pyani_plus_amd import rundb, wire
from pyani_plus_amd.methods import sourmash_hip
from tests.helpers import FIXTURE_SETS, GOLDEN, load_sig
from tests.test_host_logic import _make_run, _Session

pytestmark = pytest.mark.gpu
LOGGER = logging.getLogger("test")
K = 31


@pytest.mark.parametrize("name", list(FIXTURE_SETS))
def test_plugin_prepare_and_compute_on_gpu(name, tmp_path):
    scaled, genomes = FIXTURE_SETS[name]
    boundary = json.loads((GOLDEN / name / "boundary.json").read_text())
    run = _make_run(GOLDEN / name, genomes, scaled)
    cache = tmp_path / "cache"
    cache.mkdir()
    assert len(list(sourmash_hip.prepare_genomes(LOGGER, run, cache))) == len(genomes)  # default engine = the GPU
    sig_dir = cache / f"sourmash_k={K}_scaled={scaled}"
    for md5 in genomes:
        got, want = load_sig(sig_dir / f"{md5}.sig"), load_sig(GOLDEN / name / "sourmash" / f"{md5}.sig")
        for key in set(got) | set(want):
            if key == "filename":
                assert Path(got[key]).name == Path(want[key]).name
            else:
                assert got[key] == want[key], key
    json_file = tmp_path / "column_0.json"
    query_hashes = {g["genome_hash"]: g["length"] for g in boundary["genomes"]}
    assert sourmash_hip.compute_sourmash_hip(LOGGER, tmp_path, _Session(), run, json_file, GOLDEN / name, {}, {}, query_hashes, "", cache=cache) == 0
    key = lambda e: (e["query_hash"], e["subject_hash"])  # noqa: E731
    got = sorted(wire.load_json_comparisons(json_file)["comparisons"], key=key)
    assert got == sorted(boundary["column_json"]["comparisons"], key=key)


@pytest.mark.parametrize("name", list(FIXTURE_SETS))
def test_run_driver_on_gpu_matches_reference_database(name, tmp_path):
    scaled, _ = FIXTURE_SETS[name]
    boundary = json.loads((GOLDEN / name / "boundary.json").read_text())
    db = tmp_path / "run.sqlite"
    run = rundb.run_sourmash_hip(GOLDEN / name, db, cache=tmp_path / "cache", scaled=scaled, temp=tmp_path)
    assert run.status == "Done"
    conn = sqlite3.connect(db)
    row = conn.execute("SELECT df_identity, df_cov_query, df_hadamard FROM runs").fetchone()
    assert row == (boundary["df_identity"], boundary["df_cov_query"], boundary["df_hadamard"])
    rows = conn.execute("SELECT query_hash, subject_hash, identity, cov_query FROM comparisons ORDER BY 1, 2").fetchall()
    assert rows == [(c["query_hash"], c["subject_hash"], c["identity"], c["cov_query"]) for c in boundary["comparisons"]]
    conn.close()
