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


def test_run_driver_end_to_end_on_synthetic_fasta_files(tmp_path):
    """FASTA files on disk (plain and gzip, multi-record) -> threaded loader -> GPU -> JSON -> SQLite
    matrices, checked against the oracle on every pair."""
    import gzip

    import numpy as np

    import oracle
    from pyani_plus_amd.synth import arena_to_ascii, synth_arena_numpy

    n, scaled = 24, 200
    arena = synth_arena_numpy(n, [120_000 + 1000 * g for g in range(n)], n_species=3)
    indir = tmp_path / "genomes"
    indir.mkdir()
    seqs = []
    for g in range(n):
        seq = arena_to_ascii(arena, g)
        seqs.append(seq)
        half = len(seq) // 2
        text = b">g%d contig1 desc\n" % g + seq[:half] + b"\n>g%d contig2\n" % g + b"\n".join(seq[half + i : half + i + 80] for i in range(0, len(seq) - half, 80)) + b"\n"
        path = indir / (f"genome_{g:02d}.fna.gz" if g % 3 == 0 else f"genome_{g:02d}.fasta")
        path.write_bytes(gzip.compress(text) if g % 3 == 0 else text)
    db = tmp_path / "run.sqlite"
    run = rundb.run_sourmash_hip(indir, db, cache=tmp_path / "cache", scaled=scaled, temp=tmp_path)
    assert run.status == "Done"
    conn = sqlite3.connect(db)
    assert conn.execute("SELECT COUNT(*) FROM comparisons").fetchone()[0] == n * n
    hashes = [r[0] for r in conn.execute("SELECT genome_hash FROM genomes ORDER BY 1")]
    by_file = dict(conn.execute("SELECT fasta_filename, genome_hash FROM runs_genomes"))
    order = [by_file[p.name] for p in sorted(indir.iterdir())]
    # oracle: two records per genome, windows never span them
    sketches = [np.union1d(oracle.sketch_seq(s[: len(s) // 2], K, scaled), oracle.sketch_seq(s[len(s) // 2 :], K, scaled)) for s in seqs]
    counts = oracle.pair_counts(sketches)
    sizes = [len(s) for s in sketches]
    ident, cov, null = oracle.ani(counts, sizes, sizes, K)
    got = json.loads(conn.execute("SELECT df_identity FROM runs").fetchone()[0])
    got_cov = json.loads(conn.execute("SELECT df_cov_query FROM runs").fetchone()[0])
    assert got["index"] == hashes == sorted(order)
    for qi, q in enumerate(order):
        for si, s in enumerate(order):
            r, c = got["index"].index(q), got["columns"].index(s)
            if null[qi, si]:
                assert got["data"][r][c] is None and got_cov["data"][r][c] is None
            else:  # pandas writes 10 decimals
                assert abs(got["data"][r][c] - ident[qi, si]) < 6e-11 and abs(got_cov["data"][r][c] - cov[qi, si]) < 6e-11
    rows = conn.execute("SELECT query_hash, subject_hash, identity, cov_query FROM comparisons").fetchall()
    pos = {h: i for i, h in enumerate(order)}
    for q, s, i_val, c_val in rows:
        qi, si = pos[q], pos[s]
        assert (i_val is None) == bool(null[qi, si])
        if i_val is not None:
            assert i_val == ident[qi, si] and c_val == cov[qi, si]  # full precision in the comparisons table
    conn.close()


def test_batched_front_end_with_prefetch_and_direct_ingest_on_gpu(tmp_path):
    """SURVEY.md 8f rows 1 and 2 on the device: the files go through several loader batches (background prefetch,
    pinned arenas, streamed upload) and the database is filled by the direct route; sketches equal the oracle, the
    database equals the one the JSON route writes."""
    import gzip

    import numpy as np

    import oracle
    from pyani_plus_amd.synth import arena_to_ascii, synth_arena_numpy

    n, scaled = 14, 100
    arena = synth_arena_numpy(n, [90_000 + 7_000 * g for g in range(n)], n_species=2)
    indir = tmp_path / "genomes"
    indir.mkdir()
    seqs = [arena_to_ascii(arena, g) for g in range(n)]
    paths = []
    for g, seq in enumerate(seqs):
        text = b">g%d\n" % g + b"\n".join(seq[i : i + 70] for i in range(0, len(seq), 70)) + b"\n"
        path = indir / (f"g{g:02d}.fa.gz" if g % 4 == 1 else f"g{g:02d}.fa")
        path.write_bytes(gzip.compress(text) if g % 4 == 1 else text)
        paths.append(path)
    # ~3 files per batch: five batches, the loader of batch i+1 runs while batch i is on the device
    batches = list(sourmash_hip.sketch_fasta_batches(LOGGER, sorted(paths), kmersize=K, scaled=scaled, batch_bases=300_000))
    assert len(batches) >= 4 and sum(len(b[0]) for b in batches) == n
    by_path = {p: mins for batch_paths, _infos, sketches in batches for p, mins in zip(batch_paths, sketches)}
    for g, path in enumerate(paths):
        assert np.array_equal(by_path[path], oracle.sketch_seq(seqs[g], K, scaled)), path.name
    timings = {}
    rundb.run_sourmash_hip(indir, tmp_path / "direct.sqlite", cache=tmp_path / "c1", scaled=scaled, temp=tmp_path, ingest="direct", timings=timings)
    rundb.run_sourmash_hip(indir, tmp_path / "json.sqlite", cache=tmp_path / "c2", scaled=scaled, temp=tmp_path)

    def dump(db):
        conn = sqlite3.connect(db)
        rows = conn.execute("SELECT query_hash, subject_hash, identity, cov_query FROM comparisons ORDER BY 1, 2").fetchall()
        dfs = conn.execute("SELECT status, df_identity, df_cov_query, df_hadamard FROM runs").fetchall()
        conn.close()
        return rows, dfs

    assert dump(tmp_path / "direct.sqlite") == dump(tmp_path / "json.sqlite")
    assert len(dump(tmp_path / "direct.sqlite")[0]) == n * n and "insert_rows" in timings
    # the signature files written by both runs are the same bytes apart from the cache directory in no field
    for sig_file in (tmp_path / "c1" / f"sourmash_k={K}_scaled={scaled}").glob("*.sig"):
        assert sig_file.read_bytes() == (tmp_path / "c2" / sig_file.parent.name / sig_file.name).read_bytes()


@pytest.mark.parametrize("kmersize", [21, 51])
def test_run_driver_at_other_kmer_sizes_equals_the_oracle_backed_run(kmersize, tmp_path):
    """sourmash's other default sizes through the whole driver on the device (k = 51 takes the 128-bit kernel): the
    same signature files and the same comparison rows, bit for bit, as the oracle-backed engine produces."""
    from tests.fake_engine import OracleEngine

    name = "bacterial_example"
    scaled, genomes = FIXTURE_SETS[name]
    rundb.run_sourmash_hip(GOLDEN / name, tmp_path / "hip.sqlite", cache=tmp_path / "c_hip", kmersize=kmersize, scaled=scaled, temp=tmp_path,
                           ingest="direct")
    rundb.run_sourmash_hip(GOLDEN / name, tmp_path / "cpu.sqlite", cache=tmp_path / "c_cpu", kmersize=kmersize, scaled=scaled, temp=tmp_path,
                           ingest="direct", engine=OracleEngine())
    sig_dir = f"sourmash_k={kmersize}_scaled={scaled}"
    sigs = sorted(p.name for p in (tmp_path / "c_hip" / sig_dir).glob("*.sig"))
    assert len(sigs) == len(genomes) and sigs == sorted(p.name for p in (tmp_path / "c_cpu" / sig_dir).glob("*.sig"))
    for sig_name in sigs:
        assert (tmp_path / "c_hip" / sig_dir / sig_name).read_bytes() == (tmp_path / "c_cpu" / sig_dir / sig_name).read_bytes()

    def rows(db):
        conn = sqlite3.connect(db)
        out = conn.execute("SELECT query_hash, subject_hash, identity, cov_query FROM comparisons ORDER BY 1, 2").fetchall()
        conn.close()
        return out

    hip_rows = rows(tmp_path / "hip.sqlite")
    assert hip_rows == rows(tmp_path / "cpu.sqlite") and len(hip_rows) == len(genomes) ** 2
    assert all(r[2] == 1.0 for r in hip_rows if r[0] == r[1])


def test_integration_md_ctypes_snippet_runs(tmp_path, monkeypatch):
    """The minimal ctypes binding printed in INTEGRATION.md, executed as it stands: it has to keep up with the ABI."""
    import re

    import numpy as np

    import oracle

    root = Path(__file__).resolve().parent.parent
    text = (root / "INTEGRATION.md").read_text()
    code = re.search(r"```python\n(import ctypes as C\n.*?)```", text, flags=re.S).group(1)
    code = code.replace('"pyani_plus_amd/_lib/libpyani_hip.so"', repr(str(root / "pyani_plus_amd" / "_lib" / "libpyani_hip.so")))
    seq = np.frombuffer(b"ACGT", dtype=np.uint8)[np.random.default_rng(1).integers(0, 4, 300_000)].tobytes()
    (tmp_path / "genome.fasta").write_bytes(b">g\n" + seq + b"\n")
    monkeypatch.chdir(tmp_path)
    scope: dict = {}
    exec(code, scope)  # noqa: S102 - the document's own example
    assert np.array_equal(np.array(list(scope["mins"]), dtype=np.uint64), oracle.sketch_seq(seq, 31, 1000))
