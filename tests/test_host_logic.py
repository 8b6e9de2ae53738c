"""CPU tests: C-ABI surface, host packer, .sig / JSON / DB boundary (no GPU compute calls)."""

from __future__ import annotations

import json
import logging
import os
import re
import sqlite3
from pathlib import Path

import numpy as np
import pytest

import oracle
from pyani_plus_amd import _capi, rundb, sig, wire
from pyani_plus_amd.engine import ani_host, max_hash_for_scaled, pack_genomes
from pyani_plus_amd.methods import sourmash_hip
from pyani_plus_amd.synth import arena_to_ascii, synth_arena_numpy
from tests.fake_engine import OracleEngine
from tests.helpers import FIXTURE_SETS, GOLDEN, load_sig, read_fasta_bytes, sig_mins

ROOT = Path(__file__).resolve().parent.parent
LOGGER = logging.getLogger("test")
K = 31


# ------------------------------------------------------------------ C ABI surface
def test_library_exports_every_header_symbol():
    header = (ROOT / "include" / "pyani_hip.h").read_text()
    declared = set(re.findall(r"^PA_API [^;(]*?\b(pa_[a-z0-9_]+)\(", header, flags=re.M))
    assert len(declared) >= 20
    assert declared == set(_capi.SIGNATURES), declared ^ set(_capi.SIGNATURES)
    lib = _capi.load_library()  # raises if the .so is missing or lacks a symbol
    for name in declared:
        assert hasattr(lib, name)
    assert lib.pa_abi_version() == 5
    assert lib.pa_max_hash(300) == 61489146912365176 and lib.pa_max_hash(1000) == 18446744073709552
    assert max_hash_for_scaled(1) == 2**64 - 1


def test_tool_switches_are_not_in_the_product_library():
    """The environment switches that force a rare path, cut a kernel short or select an ablation variant are read through
    ``PA_TOOL_ENV`` and exist in ``libpyani_hip_tools.so`` (-DPA_TOOLS) only: the product library does not hold their names,
    so no variable in a worker's environment can change what it computes.  Both libraries export the whole header."""
    names = set()
    csrc = ROOT / "pyani_plus_amd" / "csrc"
    for src in [*csrc.glob("*.hip"), *csrc.glob("*.inc")]:
        text = src.read_text()
        names |= set(re.findall(r'PA_TOOL_ENV\("([A-Z_]+)"\)', text))
        # nothing in the kernels' sources reads the environment any other way
        assert not re.search(r"(?<![A-Za-z_])getenv\(", text), src
    assert {"PA_MAP_CUT", "PA_FRAGANI_NO_FREQ_CUT", "PA_KMER_VARIANT", "PA_PAIRS_SYMMETRIC", "PA_FRAGANI_HITS", "PA_FRAGANI_SORT_MAX"} <= names
    product, tools = _capi.LIB_PATH.read_bytes(), _capi.TOOLS_LIB_PATH.read_bytes()
    for name in names:
        assert name.encode() not in product, name
        assert name.encode() in tools, name
    tools_lib = _capi.load_library(tools=True)
    assert tools_lib is not _capi.load_library() and tools_lib.pa_abi_version() == _capi.load_library().pa_abi_version()
    # a host-side call with every switch set: the product library's answer does not move (the device-side form of this
    # check is tests/test_gpu_fragani.py::test_product_library_ignores_the_tool_switches)
    before = _capi.load_library().pa_fragani_window(16, 3000)
    saved = {n: os.environ.get(n) for n in names}
    try:
        for n in names:
            os.environ[n] = "1"
        assert _capi.load_library().pa_fragani_window(16, 3000) == before == 24
    finally:
        for n, v in saved.items():
            if v is None:
                os.environ.pop(n, None)
            else:
                os.environ[n] = v


def test_no_cpu_fallback_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from pyani_plus_amd.engine import HipEngine

    with pytest.raises(_capi.HipBackendError):
        HipEngine(0)
    with pytest.raises(_capi.HipBackendError):
        sourmash_hip.get_engine()


def test_product_never_imports_the_oracle():
    # the package, and the helper scripts beside it: only tests/, bench.py's checking legs and __graft_entry__ may
    for path in [*(ROOT / "pyani_plus_amd").rglob("*"), *(ROOT / "tools").glob("*")]:
        if path.suffix in {".py", ".hip", ".cpp", ".h"} and "_build" not in path.parts:
            text = path.read_text()
            assert not re.search(r"^\s*(import|from)\s+oracle\b", text, flags=re.M), path
            assert not re.search(r"#\s*include\s*[<\"][^>\"]*oracle", text), path  # never compiled in
            assert "liboracle" not in text and "orc_" not in text and "pyoracle" not in text, path  # never linked or called


# ------------------------------------------------------------------ host packer
def test_pack_fasta_semantics():
    text = b"junk before\n>r1 first title  \nACGTN\nacgt\r\n  \n>r2\nGG TT\tAA\n"
    arena = pack_genomes([text])
    assert arena.records == [2] and arena.residues == [15] and arena.invalid == [1]
    # r1 = ACGTNACGT (lower-case folded), one separator, r2 = GGTTAA; then padding
    assert arena_to_ascii(arena, 0)[: 9 + 1 + 6] == b"ACGTNACGTNGGTTAA"
    assert arena.arena_bases == 64 and rundb.fasta_length_and_description(text) == (15, "r1 first title")
    empty = pack_genomes([b"", b">only title\n"])
    assert empty.residues == [0, 0] and empty.records == [0, 1]
    bare = pack_genomes([b"ACGTX" * 20], fasta=False)
    assert bare.residues == [100] and bare.invalid == [20] and bare.arena_bases == 128
    # a genome whose length is a multiple of 64 still ends with an invalid position (then padding)
    exact = pack_genomes([b"ACGT" * 16, b"ACGT" * 16], fasta=False)
    assert exact.genome_start.tolist() == [0, 128, 256]
    assert arena_to_ascii(exact, 0) == b"ACGT" * 16 and (int(exact.mask[2]) & 1) == 1


def test_packers_list_the_residues_that_are_neither_acgt_nor_n(tmp_path):
    """The arena's mask bit stands for N; IUPAC codes and every other byte are listed beside it -- (arena position, upper-cased
    byte), ascending -- by pa_text_ambiguous (texts packed one by one) and pa_fasta_batch_ambiguous (the threaded loader),
    so that the fragment-ANI kernels hash them as fastANI does: as the characters they are."""
    import gzip

    from pyani_plus_amd.engine import load_fasta_files

    text_a = b"junk\n>r1 t\nACGTRYKMnnACGT\n>r2\nSWacgtBDHVxACGT*\n"
    text_b = b">y\nRRRRACGT\nACGTK\n"
    arena = pack_genomes([text_a, b">x\nACGTNNNN\n", text_b])
    # r1 holds 14 residues at 0..13, one separator, r2 starts at 15; genome 2 starts at 128
    assert arena.ambig_pos.tolist() == [4, 5, 6, 7, 15, 16, 21, 22, 23, 24, 25, 30, 128, 129, 130, 131, 140]
    assert bytes(arena.ambig_byte) == b"RYKMSWBDHVX*RRRRK"  # lower case folded, N and n left to the mask bit
    assert arena_to_ascii(arena, 0) == b"ACGTRYKMNNACGTNSWACGTBDHVXACGT*" and arena_to_ascii(arena, 1) == b"ACGTNNNN"
    assert arena.invalid == [14, 4, 5]
    bare = pack_genomes([b"ACGTRYn* "], fasta=False)  # a bare sequence: every byte is a residue
    assert bare.ambig_pos.tolist() == [4, 5, 7, 8] and bytes(bare.ambig_byte) == b"RY* "
    (tmp_path / "a.fasta").write_bytes(text_a)
    with gzip.open(tmp_path / "b.fna.gz", "wb") as fh:
        fh.write(text_b)
    infos, loaded = load_fasta_files([tmp_path / "a.fasta", tmp_path / "b.fna.gz"])
    assert [i.status for i in infos] == [0, 0]
    assert loaded.ambig_pos.tolist() == [4, 5, 6, 7, 15, 16, 21, 22, 23, 24, 25, 30, 64, 65, 66, 67, 76]
    assert bytes(loaded.ambig_byte) == b"RYKMSWBDHVX*RRRRK"
    clean = pack_genomes([b">c\nACGTNNACGT\n"])
    assert len(clean.ambig_pos) == 0


@pytest.mark.parametrize("name", ["viral_example", "bad_alignments"])
def test_pack_matches_reference_lengths(name, golden):
    boundary = json.loads((golden / name / "boundary.json").read_text())
    for g in boundary["genomes"]:
        text = read_fasta_bytes(golden / name / g["fasta_filename"])
        arena = pack_genomes([text])
        assert arena.residues[0] == g["length"]
        assert rundb.fasta_length_and_description(text) == (g["length"], g["description"])


def test_capacity_and_argument_errors():
    import ctypes as C

    lib = _capi.load_library()
    packed = np.zeros(4, dtype=np.uint32)
    mask = np.zeros(2, dtype=np.uint32)
    nb = C.c_uint64(0)
    st = lib.pa_pack_seq(b"A" * 100, 100, packed.ctypes.data, mask.ctypes.data, 64, C.byref(nb), None)
    assert st == _capi.PA_E_CAPACITY and nb.value == 128 and b"too small" in lib.pa_last_error()
    assert lib.pa_pack_seq(b"A", 1, packed.ctypes.data, mask.ctypes.data, 63, C.byref(nb), None) == -1
    if lib.pa_device_count() == 0:
        ctx = C.c_void_p()
        assert lib.pa_ctx_create(0, C.byref(ctx)) == -5 and b"no CPU fallback" in lib.pa_last_error()


# ------------------------------------------------------------------ ANI transform
def test_ani_host_equals_oracle_and_fixture_strings():
    rng = np.random.default_rng(5)
    sizes = rng.integers(1, 6000, size=40).astype(np.uint64)
    counts = np.minimum(rng.integers(0, 6000, size=(40, 40)), np.minimum.outer(sizes, sizes)).astype(np.uint32)
    np.fill_diagonal(counts, sizes)
    a = ani_host(counts, sizes, sizes, K)
    b = oracle.ani(counts, sizes, sizes, K)
    assert np.array_equal(a[2], b[2])
    assert np.array_equal(a[0][~a[2]], b[0][~b[2]]) and np.array_equal(a[1][~a[2]], b[1][~b[2]])
    assert np.all(np.diag(a[0]) == 1.0) and np.all(np.isnan(a[0][a[2]]))


# ------------------------------------------------------------------ .sig files
@pytest.mark.parametrize("name", list(FIXTURE_SETS))
def test_sig_writer_reproduces_fixture_files(name, tmp_path):
    scaled, genomes = FIXTURE_SETS[name]
    for md5, fasta in genomes.items():
        ref = GOLDEN / name / "sourmash" / f"{md5}.sig"
        want = load_sig(ref)
        out = tmp_path / f"{md5}.sig"
        sig.write_sig(out, name=md5, filename=f"somewhere/{fasta}", ksize=K, max_hash=max_hash_for_scaled(scaled), mins=sig_mins(ref))
        got = load_sig(out)
        for key in set(got) | set(want):  # the reference's own comparison: every key, filename by basename
            if key == "filename":
                assert Path(got[key]).name == Path(want[key]).name
            else:
                assert got[key] == want[key], key
        # byte layout: single line, compact separators, same key order as sourmash writes
        ref_text = ref.read_text()
        assert out.read_text().replace(f"somewhere/{fasta}", want["filename"]) == ref_text
        mins, sketch = sig.read_sig(out, ksize=K, max_hash=max_hash_for_scaled(scaled))
        assert np.array_equal(mins, sig_mins(ref)) and sketch["md5sum"] == want["signatures"][0]["md5sum"]
    with pytest.raises(ValueError, match="no DNA sketch"):
        sig.read_sig(out, ksize=21)


# ------------------------------------------------------------------ plugin + driver (host logic, oracle-backed engine)
def _make_run(fasta_dir: Path, genomes: dict[str, str], scaled: int, method=sourmash_hip.METHOD):
    tool = sourmash_hip.get_sourmash_hip()
    config = rundb.Configuration(1, method, tool.exe_path.stem, tool.version, kmersize=K, extra=f"scaled={scaled}")
    assoc = [rundb.RunGenomeAssociation(md5, fasta) for md5, fasta in genomes.items()]
    return rundb.Run(1, config, str(fasta_dir), assoc, "Testing")


class _Session:
    commits = 0

    def commit(self):
        self.commits += 1


@pytest.mark.parametrize("name", list(FIXTURE_SETS))
def test_prepare_and_compute_match_reference_boundary(name, tmp_path):
    scaled, genomes = FIXTURE_SETS[name]
    boundary = json.loads((GOLDEN / name / "boundary.json").read_text())
    run = _make_run(GOLDEN / name, genomes, scaled)
    cache = tmp_path / "cache"
    with pytest.raises(ValueError, match="does not exist"):
        list(sourmash_hip.prepare_genomes(LOGGER, run, cache, engine=OracleEngine()))
    cache.mkdir()
    done = list(sourmash_hip.prepare_genomes(LOGGER, run, cache, engine=OracleEngine()))
    assert [e.genome_hash for e in done] == list(genomes)
    sig_dir = cache / f"sourmash_k={K}_scaled={scaled}"
    for md5 in genomes:
        got, want = load_sig(sig_dir / f"{md5}.sig"), load_sig(GOLDEN / name / "sourmash" / f"{md5}.sig")
        for key in set(got) | set(want):
            if key == "filename":
                assert Path(got[key]).name == Path(want[key]).name
            else:
                assert got[key] == want[key], key
    # idempotent: existing signatures are not recomputed (engine=None would raise without a GPU)
    stamp = {p.name: p.stat().st_mtime_ns for p in sig_dir.glob("*.sig")}
    assert len(list(sourmash_hip.prepare_genomes(LOGGER, run, cache, engine=None))) == len(genomes)
    assert stamp == {p.name: p.stat().st_mtime_ns for p in sig_dir.glob("*.sig")}

    json_file = tmp_path / "sourmash-hip.run_1.column_0.json"
    query_hashes = {g["genome_hash"]: g["length"] for g in boundary["genomes"]}
    rc = sourmash_hip.compute_sourmash_hip(
        LOGGER, tmp_path, _Session(), run, json_file, GOLDEN / name, {}, {}, query_hashes, "", cache=cache, engine=OracleEngine()
    )
    assert rc == 0
    data = wire.load_json_comparisons(json_file)
    assert set(data["configuration"]) == set(boundary["column_json"]["configuration"])
    assert data["configuration"]["method"] == "sourmash-hip" and data["configuration"]["extra"] == f"scaled={scaled}"
    assert set(data["uname"]) == {"system", "release", "machine"}
    key = lambda e: (e["query_hash"], e["subject_hash"])  # noqa: E731
    got = sorted(data["comparisons"], key=key)
    want = sorted(boundary["column_json"]["comparisons"], key=key)
    assert got == want  # same keys, None where NULL, bit-identical floats

    # single subject column (compute-column --subject <hash>)
    subject = sorted(genomes)[-1]
    rc = sourmash_hip.compute_sourmash_hip(
        LOGGER, tmp_path, _Session(), run, json_file, GOLDEN / name, {}, {}, query_hashes, subject, cache=cache, engine=OracleEngine()
    )
    col = sorted(wire.load_json_comparisons(json_file)["comparisons"], key=key)
    assert rc == 0 and col == [e for e in want if e["subject_hash"] == subject]


def test_compute_error_paths(tmp_path):
    scaled, genomes = FIXTURE_SETS["viral_example"]
    run = _make_run(GOLDEN / "viral_example", genomes, scaled)
    with pytest.raises(SystemExit, match="Missing sourmash signatures directory"):
        sourmash_hip.compute_sourmash_hip(LOGGER, tmp_path, _Session(), run, tmp_path / "x.json", tmp_path, {}, {}, {}, "", cache=tmp_path)
    bad = _make_run(GOLDEN / "viral_example", genomes, scaled, method="sourmash")
    with pytest.raises(SystemExit, match="Expected run to be for sourmash-hip"):
        list(sourmash_hip.prepare_genomes(LOGGER, bad, tmp_path))
    run.configuration.version = "0.0.0"
    with pytest.raises(SystemExit, match="Run configuration was"):
        sourmash_hip.compute_sourmash_hip(LOGGER, tmp_path, _Session(), run, tmp_path / "x.json", tmp_path, {}, {}, {}, "", cache=tmp_path)
    run = _make_run(GOLDEN / "viral_example", genomes, scaled)
    run.configuration.extra = "num=500"
    with pytest.raises(ValueError, match="scaled=N"):
        list(sourmash_hip.prepare_genomes(LOGGER, run, tmp_path))
    run.configuration.extra = None
    with pytest.raises(SystemExit, match="requires extra setting"):
        list(sourmash_hip.prepare_genomes(LOGGER, run, tmp_path))
    run.configuration.kmersize = None
    with pytest.raises(SystemExit, match="requires a k-mer size"):
        list(sourmash_hip.prepare_genomes(LOGGER, run, tmp_path))
    (tmp_path / "sourmash_k=31_scaled=300").mkdir()
    run = _make_run(GOLDEN / "viral_example", genomes, scaled)
    with pytest.raises(SystemExit, match="Missing sourmash signature file"):
        sourmash_hip.compute_sourmash_hip(
            LOGGER, tmp_path, _Session(), run, tmp_path / "x.json", tmp_path, {}, {}, {h: 1 for h in genomes}, "", cache=tmp_path, engine=OracleEngine()
        )


def _prepared_bacteria(tmp_path):
    scaled, genomes = FIXTURE_SETS["bacterial_example"]
    run = _make_run(GOLDEN / "bacterial_example", genomes, scaled)
    cache = tmp_path / "cache"
    cache.mkdir()
    list(sourmash_hip.prepare_genomes(LOGGER, run, cache, engine=OracleEngine()))
    boundary = json.loads((GOLDEN / "bacterial_example" / "boundary.json").read_text())
    return run, cache, {g["genome_hash"]: g["length"] for g in boundary["genomes"]}, boundary


def test_compute_in_subject_tiles_keeps_finished_tiles_on_interrupt(tmp_path):
    """Subject tiles are flushed one by one (the reference's 100 000-row flush, private_cli.py:1863-1894):
    tiled output == single-tile output; Ctrl-C after tile 1 keeps tile 1, sets the status, returns 0."""
    run, cache, query_hashes, boundary = _prepared_bacteria(tmp_path)
    key = lambda e: (e["query_hash"], e["subject_hash"])  # noqa: E731
    want = sorted(boundary["column_json"]["comparisons"], key=key)
    json_file = tmp_path / "tiled.json"
    rc = sourmash_hip.compute_sourmash_hip(
        LOGGER, tmp_path, _Session(), run, json_file, tmp_path, {}, {}, query_hashes, "", cache=cache, engine=OracleEngine(), tile_columns=1
    )
    data = wire.load_json_comparisons(json_file)
    assert rc == 0 and sorted(data["comparisons"], key=key) == want
    # rows arrive tile by tile: the first four rows are all queries against the first (sorted) subject
    subjects = sorted(query_hashes)
    assert {e["subject_hash"] for e in data["comparisons"][:4]} == {subjects[0]}

    class Interrupting(OracleEngine):
        calls = 0

        def pair_counts(self, *args, **kwargs):
            self.calls += 1
            if self.calls == 2:
                raise KeyboardInterrupt
            return super().pair_counts(*args, **kwargs)

    session = _Session()
    rc = sourmash_hip.compute_sourmash_hip(
        LOGGER, tmp_path, session, run, json_file, tmp_path, {}, {}, query_hashes, "", cache=cache, engine=Interrupting(), tile_columns=1
    )
    kept = wire.load_json_comparisons(json_file)["comparisons"]
    assert rc == 0 and run.status == "Worker interrupted" and session.commits == 1
    assert sorted(kept, key=key) == [e for e in want if e["subject_hash"] == subjects[0]]


def test_backend_failure_and_save_failure_follow_the_reference_conventions(tmp_path):
    """A failing library call ends the worker like a failing tool (utils.py:262-283: SystemExit with the message);
    a column file that cannot be written returns RECORDING_FAILED = 2 (private_cli.py:1896-1902)."""
    from pyani_plus_amd._capi import HipBackendError

    run, cache, query_hashes, _boundary = _prepared_bacteria(tmp_path)

    class Failing(OracleEngine):
        def pair_counts(self, *args, **kwargs):
            raise HipBackendError("pa_pair_counts failed with status -2: hipErrorLaunchFailure (test)")

    with pytest.raises(SystemExit, match=r"sourmash-hip comparison failed in libpyani_hip\.so: pa_pair_counts failed with status -2"):
        sourmash_hip.compute_sourmash_hip(
            LOGGER, tmp_path, _Session(), run, tmp_path / "f.json", tmp_path, {}, {}, query_hashes, "", cache=cache, engine=Failing()
        )
    rc = sourmash_hip.compute_sourmash_hip(
        LOGGER, tmp_path, _Session(), run, tmp_path / "no_such_dir" / "f.json", tmp_path, {}, {}, query_hashes, "", cache=cache, engine=OracleEngine()
    )
    assert rc == sourmash_hip.RECORDING_FAILED == 2


def test_strict_ani_threaded_symmetric_equals_single_thread():
    """pa_ani_host on the host pool, with one pow per ordered pair for a square block, gives the very doubles of the
    plain single-threaded two-pow form (the reference's arithmetic)."""
    from pyani_plus_amd.engine import ani_host

    rng = np.random.default_rng(5)
    n = 257
    sizes = rng.integers(50, 6000, n).astype(np.uint64)
    counts = rng.integers(0, 50, (n, n)).astype(np.uint32)
    counts = np.minimum(counts, counts.T)
    counts[rng.random((n, n)) < 0.6] = 0
    counts = np.minimum(counts, counts.T)
    np.fill_diagonal(counts, sizes.astype(np.uint32))
    ref = ani_host(counts, sizes, sizes, 31, symmetric=False, threads=1)
    for kwargs in ({"symmetric": True, "threads": 0}, {"symmetric": False, "threads": 7}, {"symmetric": True, "threads": 3}):
        got = ani_host(counts, sizes, sizes, 31, **kwargs)
        assert all(np.array_equal(a, b, equal_nan=True) for a, b in zip(ref, got)), kwargs
    o_ident, o_cov, o_null = oracle.ani(counts, sizes, sizes, 31)
    assert np.array_equal(ref[2], o_null) and np.array_equal(ref[0][~o_null], o_ident[~o_null]) and np.array_equal(ref[1][~o_null], o_cov[~o_null])
    with pytest.raises(_capi.HipBackendError, match="square"):
        ani_host(counts[:3], sizes[:3], sizes, 31, symmetric=True)


def test_fastani_output_conventions():
    """What fastANI prints, as the reference parses it (pyani_plus/methods/fastani.py:98-120): six significant
    digits, and a line only when the kept fragments cover minFraction of the shorter genome."""
    from pyani_plus_amd.methods import fastani_hip

    assert fastani_hip.fastani_print_round(82.91243619) == 82.9124
    assert fastani_hip.fastani_print_round(99.99531) == 99.9953
    assert fastani_hip.fastani_print_round(100.0) == 100.0
    assert fastani_hip.fastani_print_round(99.999996) == 100.0
    # 300 of 1666 fragments of a 5 Mb query against a 1 Mb reference: 18 % of the query's fragments, 90 % of the reference
    assert fastani_hip.is_reported(300, 1666, 3000, 0.2, 5_000_000, 1_000_000)
    assert not fastani_hip.is_reported(300, 1666, 3000, 0.2, 5_000_000, 5_000_000)
    assert not fastani_hip.is_reported(0, 10, 3000, 0.2, 30_000, 30_000) and not fastani_hip.is_reported(0, 0, 3000, 0.2, 0, 0)
    lens = fastani_hip.mappable_lengths([5000, 2999, 3000, 100], [0, 0, 1, 1], 3, 3000)
    assert lens.tolist() == [5000, 3000, 0]


@pytest.mark.parametrize("name", list(FIXTURE_SETS))
def test_run_driver_database_matches_reference(name, tmp_path):
    scaled, genomes = FIXTURE_SETS[name]
    boundary = json.loads((GOLDEN / name / "boundary.json").read_text())
    db = tmp_path / "run.sqlite"
    run = rundb.run_sourmash_hip(GOLDEN / name, db, cache=tmp_path / "cache", scaled=scaled, engine=OracleEngine(), temp=tmp_path)
    assert run.status == "Done"
    conn = sqlite3.connect(db)
    row = conn.execute("SELECT status, df_identity, df_cov_query, df_hadamard, df_aln_length, df_sim_errors FROM runs").fetchone()
    assert row[0] == "Done"
    for got, key in zip(row[1:], ("df_identity", "df_cov_query", "df_hadamard", "df_aln_length", "df_sim_errors")):
        assert got == boundary[key], key  # identical strings, incl. pandas' 10-decimal rounding and nulls
    rows = conn.execute(
        "SELECT query_hash, subject_hash, identity, aln_length, sim_errors, cov_query, cov_subject FROM comparisons ORDER BY 1, 2"
    ).fetchall()
    want = [(c["query_hash"], c["subject_hash"], c["identity"], c["aln_length"], c["sim_errors"], c["cov_query"], c["cov_subject"]) for c in boundary["comparisons"]]
    assert rows == want
    g_rows = conn.execute("SELECT genome_hash, length, description FROM genomes ORDER BY 1").fetchall()
    assert g_rows == [(g["genome_hash"], g["length"], g["description"]) for g in boundary["genomes"]]
    conn.close()
    # a second run over the same inputs re-uses every comparison (no engine needed)
    again = rundb.run_sourmash_hip(GOLDEN / name, db, cache=tmp_path / "cache", scaled=scaled, engine=None)
    assert again.run_id == 2 and again.status == "Done"


@pytest.mark.parametrize("name", list(FIXTURE_SETS))
def test_direct_ingest_writes_the_same_database_as_the_json_route(name, tmp_path):
    """binary tile files + rows inserted from the matrices + matrix cache from memory give the
    database the reference's JSON route gives (rows, NULLs and the five cached matrix strings)."""
    import sqlite3

    scaled, _genomes = FIXTURE_SETS[name]
    timings = {}
    rundb.run_sourmash_hip(GOLDEN / name, tmp_path / "json.sqlite", cache=tmp_path / "c1", scaled=scaled, engine=OracleEngine(), temp=tmp_path)
    run = rundb.run_sourmash_hip(GOLDEN / name, tmp_path / "direct.sqlite", cache=tmp_path / "c2", scaled=scaled, engine=OracleEngine(),
                                 temp=tmp_path, ingest="direct", timings=timings)

    def dump(db):
        conn = sqlite3.connect(db)
        rows = conn.execute("SELECT query_hash, subject_hash, configuration_id, identity, cov_query, aln_length, sim_errors, uname_system "
                            "FROM comparisons ORDER BY 1, 2").fetchall()
        dfs = conn.execute("SELECT status, df_identity, df_cov_query, df_aln_length, df_sim_errors, df_hadamard FROM runs").fetchall()
        conn.close()
        return rows, dfs

    assert dump(tmp_path / "json.sqlite") == dump(tmp_path / "direct.sqlite")
    assert {"fasta_front_end_and_sketch", "signature_files", "pairs_and_tile_files", "insert_rows", "matrix_cache"} <= set(timings)
    # the tile file alone restores the rows (resume): import it into a database that has the run but no comparisons
    tile = next(tmp_path.glob("*.tile_0.npz"))
    conn = rundb.connect_to_db(tmp_path / "direct.sqlite")
    conn.execute("DELETE FROM comparisons")
    conn.commit()
    assert rundb.import_tile(LOGGER, conn, rundb.load_run(conn, run.run_id), tile) == len(_genomes) ** 2
    conn.close()
    assert dump(tmp_path / "json.sqlite")[0] == [(*r[:2], r[2], *r[3:]) for r in dump(tmp_path / "direct.sqlite")[0]]


def test_native_row_insert_equals_the_python_one(tmp_path):
    """pa_sqlite_insert_comparisons (prepared statement stepped from C) against sqlite3.executemany: same rows in
    the same order, NULLs where is_null, INSERT OR IGNORE semantics on a second call, loud failure on a bad path."""
    import ctypes as C
    import hashlib
    import sqlite3

    from pyani_plus_amd import _capi

    n = 37
    hashes = sorted(hashlib.md5(str(i).encode()).hexdigest() for i in range(n))
    rng = np.random.default_rng(5)
    ident, cov, null = rng.random((n, n)), rng.random((n, n)), rng.random((n, n)) < 0.2
    ident[0, 0], cov[0, 1] = 1.0, 5e-324

    class RunStub:
        configuration_id = 0

    dumps = {}
    for native in (True, False):
        conn = rundb.connect_to_db(tmp_path / f"{native}.sqlite")
        RunStub.configuration_id = rundb.db_configuration(conn, "sourmash-hip", "libpyani_hip", "0.1.0", kmersize=31, extra="scaled=1000").configuration_id
        conn.commit()
        assert rundb.ingest_matrices(conn, RunStub, hashes, hashes, ident, cov, null, native=native) == n * n
        query = ("SELECT comparison_id, query_hash, subject_hash, configuration_id, identity, aln_length, sim_errors, cov_query, "
                 "cov_subject, uname_system, uname_release, uname_machine FROM comparisons ORDER BY comparison_id")
        dumps[native] = conn.execute(query).fetchall()
        # every pair is there already: nothing is added, nothing fails
        assert rundb.ingest_matrices(conn, RunStub, hashes[:5], hashes, ident[:5] * 0.5, cov[:5], null[:5], native=native) == 5 * n
        assert conn.execute(query).fetchall() == dumps[native]
        conn.close()
    assert dumps[True] == dumps[False] and len(dumps[True]) == n * n
    assert [r[4] for r in dumps[True][:2]] == [None if null[0, 0] else 1.0, None if null[0, 1] else ident[0, 1]]
    # an in-memory database has no file another connection could open: the Python route serves it
    mem = sqlite3.connect(":memory:")
    mem.executescript(rundb.SCHEMA)
    assert rundb.ingest_matrices_native(mem, RunStub, hashes, hashes, ident, cov, null) is None
    assert rundb.ingest_matrices(mem, RunStub, hashes, hashes, ident, cov, null) == n * n
    assert mem.execute("SELECT COUNT(*) FROM comparisons").fetchone()[0] == n * n
    # failures are reported, not swallowed
    lib = _capi.load_library()
    arr = (C.c_char_p * 1)(b"a")
    one = np.zeros(1)
    status = lib.pa_sqlite_insert_comparisons(str(tmp_path / "absent" / "x.sqlite").encode(), 1, b"s", b"r", b"m", arr, 1, arr, 1,
                                              one.ctypes.data, one.ctypes.data, np.zeros(1, np.uint8).ctypes.data, None)
    assert status == _capi.PA_E_IO and "cannot open" in _capi.last_error()
    (tmp_path / "empty.sqlite").write_bytes(b"")
    status = lib.pa_sqlite_insert_comparisons(str(tmp_path / "empty.sqlite").encode(), 1, b"s", b"r", b"m", arr, 1, arr, 1,
                                              one.ctypes.data, one.ctypes.data, np.zeros(1, np.uint8).ctypes.data, None)
    assert status == _capi.PA_E_IO and "no such table" in _capi.last_error()


def test_environment_selectors_pick_equivalent_implementations(tmp_path):
    """The product library reads four environment variables, each choosing between implementations that must be
    indistinguishable by their results: PA_GUNZIP=z (zlib instead of the library's own inflate, csrc/fasta_batch.cpp),
    PA_PACK_SCALAR (the per-character packer instead of the AVX2 one, csrc/pack_host.cpp), PA_HOST_SLAB_CACHE=0 (no
    reuse of host mappings between batches) and PA_SQLITE_SYNCHRONOUS=OFF (csrc/sqlite_ingest.cpp).  A child process per
    setting -- three of the four are read once per process -- loads the reference's gzipped bacteria and two plain files
    twice, packs a text that takes every packer path, and inserts a matrix natively: checksums (the reference's genome
    identity, utils.py:178-196), lengths, arenas, ambiguity lists and database rows are the same under every setting."""
    import os
    import subprocess
    import sys

    paths = [GOLDEN / "bacterial_example" / f for f in ("NC_002696.fasta.gz", "NC_010338.fna.gz")] + [GOLDEN / "viral_example" / "OP073605.fasta", GOLDEN / "MIBY01000005.fasta"]
    settings = {"default": {}, "zlib": {"PA_GUNZIP": "z"}, "scalar": {"PA_PACK_SCALAR": "1"}, "no_slab_cache": {"PA_HOST_SLAB_CACHE": "0"},
                "sync_off": {"PA_SQLITE_SYNCHRONOUS": "OFF"}, "all": {"PA_GUNZIP": "z", "PA_PACK_SCALAR": "1", "PA_HOST_SLAB_CACHE": "0", "PA_SQLITE_SYNCHRONOUS": "OFF"}}
    results = {}
    for name, extra in settings.items():
        work = tmp_path / name
        work.mkdir()
        env = {k: v for k, v in os.environ.items() if not k.startswith("PA_")}
        env.update(extra)
        done = subprocess.run([sys.executable, str(Path(__file__).parent / "tools" / "env_selectors_child.py"), str(work), *map(str, paths)],
                              env=env, capture_output=True, text=True, timeout=600, check=False)
        assert done.returncode == 0, f"{name}: {done.stderr[-2000:]}"
        results[name] = json.loads(done.stdout.strip().splitlines()[-1])
    base = results["default"]
    assert base["load_0"] == base["load_1"] and base["n_rows"] == 23 * 23
    assert base["load_0"]["md5"][0] == "f19cb07198a41a4406a22b2f57a6b5e7"  # NC_002696 as the reference names it (tests/fixtures/bacterial_example)
    for name, got in results.items():
        assert got == base, f"PA_* setting {name!r} changed a result"


def test_driver_rejects_duplicates_and_bad_gzip(tmp_path):
    d = tmp_path / "in"
    d.mkdir()
    (d / "a.fasta").write_bytes(b">a\nACGT\n")
    (d / "b.fna").write_bytes(b">a\nACGT\n")
    with pytest.raises(SystemExit, match="Multiple genomes with same MD5"):
        rundb.run_sourmash_hip(d, tmp_path / "x.sqlite", engine=OracleEngine())
    (d / "b.fna").unlink()
    (d / "c.fa.gz").write_bytes(b">c\nACGT\n")
    with pytest.raises(SystemExit, match="NOT gzip compressed"):
        rundb.run_sourmash_hip(d, tmp_path / "y.sqlite", engine=OracleEngine())
    with pytest.raises(SystemExit, match="No FASTA input genomes"):
        rundb.run_sourmash_hip(tmp_path, tmp_path / "z.sqlite", engine=OracleEngine())
    # parameters are checked before any file is read (the reference passes any --kmersize on, public_cli_args.py:229)
    with pytest.raises(SystemExit, match="k-mer sizes 1 to 64"):
        rundb.run_sourmash_hip(d, tmp_path / "k.sqlite", kmersize=65, engine=OracleEngine())
    with pytest.raises(SystemExit, match="ingest must be"):
        rundb.run_sourmash_hip(d, tmp_path / "k.sqlite", ingest="csv", engine=OracleEngine())


def test_synthetic_generator_roundtrip():
    arena = synth_arena_numpy(5, [1000, 64, 777, 1000, 1000], n_species=2)
    assert arena.arena_bases % 64 == 0 and arena.n_genomes == 5
    a0, a2, a4 = (arena_to_ascii(arena, g) for g in (0, 2, 4))
    assert len(a0) == 1000 and len(a2) == 777 and set(a0) <= set(b"ACGT")
    # same species (0, 2, 4 -> species 0): mostly identical prefixes
    same = sum(x == y for x, y in zip(a0, a4))
    assert same > 900
    again = synth_arena_numpy(5, [1000, 64, 777, 1000, 1000], n_species=2)
    assert np.array_equal(again.packed, arena.packed) and np.array_equal(again.mask, arena.mask)


def test_bulk_json_writer_is_byte_identical_to_json_dumps(tmp_path):
    """pa_write_comparisons_json must produce exactly what the reference's json.dumps produces."""
    import platform

    rng = np.random.default_rng(3)
    nq, ns = 7, 5
    ident = rng.random((nq, ns))
    cov = rng.random((nq, ns)) ** 8
    special = [1.0, 0.0, 1e-4, 9.999e-5, 1e-5, 1.5e-7, 5e-324, 1e15, 1e16, 1.2345e22, 0.1, 1 / 3, 0.9997081124294064, 2.0**-31]
    ident.ravel()[: len(special)] = special
    cov.ravel()[-len(special) :] = special
    null = rng.random((nq, ns)) < 0.2
    queries = [f"{i:032x}" for i in range(nq)]
    subjects = [f"{i + 100:032x}" for i in range(ns)]
    cfg = rundb.Configuration(3, "sourmash-hip", "libpyani_hip", "0.1.0", kmersize=31, extra="scaled=1000")
    out = tmp_path / "bulk.json"
    wire.export_json_matrices(LOGGER, out, cfg, queries, subjects, ident, cov, null)
    uname = platform.uname()
    want = json.dumps(
        {
            "configuration": wire.configuration_dict(cfg),
            "uname": {"system": uname.system, "release": uname.release, "machine": uname.machine},
            "comparisons": [
                {
                    "query_hash": q,
                    "subject_hash": s,
                    "identity": None if null[i, j] else float(ident[i, j]),
                    "cov_query": None if null[i, j] else float(cov[i, j]),
                }
                for i, q in enumerate(queries)
                for j, s in enumerate(subjects)
            ],
        }
    )
    assert out.read_text() == want
    via_dicts = tmp_path / "dicts.json"
    wire.export_json_db_entries(LOGGER, via_dicts, cfg, json.loads(want)["comparisons"])
    assert via_dicts.read_text() == want
    wire.export_json_matrices(LOGGER, out, cfg, [], [], np.zeros((0, 0)), np.zeros((0, 0)), np.zeros((0, 0), bool))
    assert json.loads(out.read_text())["comparisons"] == []


def test_threaded_fasta_loader_matches_python_reader(tmp_path):
    from pyani_plus_amd.engine import load_fasta_files
    from tests.helpers import md5_hex

    paths = [GOLDEN / "bacterial_example" / "NC_002696.fasta.gz", GOLDEN / "viral_example" / "OP073605.fasta", GOLDEN / "MIBY01000005.fasta"]
    bad_gz = tmp_path / "plain.fa.gz"
    bad_gz.write_bytes(b">x\nACGT\n")
    import gzip as _gzip

    hidden_gz = tmp_path / "zipped.fasta"
    hidden_gz.write_bytes(_gzip.compress(b">x\nACGT\n"))
    empty = tmp_path / "empty.fna"
    empty.write_bytes(b"\n\n")
    two_members = tmp_path / "two.fa.gz"
    two_members.write_bytes(_gzip.compress(b">a desc one  \nACGTNN\n") + _gzip.compress(b">b\nGGCC\n"))
    infos, arena = load_fasta_files([*paths, bad_gz, hidden_gz, empty, tmp_path / "missing.fa", two_members], threads=3)
    for info, path in zip(infos[:3], paths):
        text = read_fasta_bytes(path)
        assert info.status == 0 and info.md5 == md5_hex(text)
        assert (info.length, info.description) == rundb.fasta_length_and_description(text)
    assert infos[0].records == 2 and infos[0].gzip and not infos[1].gzip and infos[2].invalid == 28
    assert infos[3].status != 0 and infos[3].message == "Has .gz ending, but plain.fa.gz is NOT gzip compressed"
    assert infos[4].status != 0 and infos[4].message == "No .gz ending, but zipped.fasta is gzip compressed"
    assert infos[5].status != 0 and infos[5].message == "File empty.fna is not recognised as a FASTA record"
    assert infos[6].status != 0 and "not found" in infos[6].message
    assert infos[7].status == 0 and infos[7].length == 10 and infos[7].records == 2 and infos[7].description == "a desc one"
    assert infos[7].md5 == md5_hex(b">a desc one  \nACGTNN\n>b\nGGCC\n")
    # the arena holds exactly the files that loaded, identical to packing their text one by one
    ok_texts = [read_fasta_bytes(p) for p in paths] + [b">a desc one  \nACGTNN\n>b\nGGCC\n"]
    ref = pack_genomes(ok_texts)
    assert np.array_equal(arena.packed, ref.packed) and np.array_equal(arena.mask, ref.mask)
    assert np.array_equal(arena.genome_start, ref.genome_start) and arena.residues == ref.residues


def _pack_truth(text: bytes):
    """FASTA text -> (packed words, mask words, residues, records, invalid, record starts, record lengths) by the
    rules of pyani_plus/utils.py:67-90, one character at a time in Python: the independent statement the vectorised
    packer is held to."""
    codes, bad, rec_start, rec_len = [], [], [], []
    residues = invalid = 0
    in_record = False
    for line in text.split(b"\n"):
        if line[:1] == b">":
            if in_record:
                codes.append(0)
                bad.append(1)
            in_record = True
            rec_start.append(len(codes))
            rec_len.append(0)
            continue
        if not in_record:
            continue
        for ch in line:
            if ch in b" \t\r":
                continue
            residues += 1
            rec_len[-1] += 1
            code = b"ACGT".find(bytes([ch]).upper())
            codes.append(max(code, 0))
            bad.append(int(code < 0))
            invalid += int(code < 0)
    codes.append(0)
    bad.append(1)
    while len(codes) % 64:
        codes.append(0)
        bad.append(1)
    c = np.array(codes, dtype=np.uint64).reshape(-1, 16)
    packed = (c << (2 * np.arange(16, dtype=np.uint64))).sum(axis=1).astype(np.uint32)
    m = np.array(bad, dtype=np.uint64).reshape(-1, 32)
    mask = (m << np.arange(32, dtype=np.uint64)).sum(axis=1).astype(np.uint32)
    return packed, mask, residues, len(rec_start), invalid, rec_start, rec_len


def test_vector_packer_equals_the_character_rules(tmp_path):
    """The AVX2 chunks of pa_pack_fasta (32 / 16 clean bases at a time) against a character-by-character packer, on
    texts built to break chunking: every line length around the chunk sizes, Ns and IUPAC codes, lower case, blanks,
    CR LF, records of all sizes, text before the first record, no final line feed."""
    import ctypes as C

    from pyani_plus_amd.engine import load_fasta_files

    lib = _capi.load_library()
    rng = np.random.default_rng(11)
    texts = []
    for case in range(60):
        parts = [b"junk before the first record\n"] if case % 7 == 0 else []
        for rec in range(int(rng.integers(1, 5))):
            parts.append(b">rec%d some title\n" % rec)
            for _line in range(int(rng.integers(0, 9))):
                n = int(rng.choice([0, 1, 15, 16, 17, 31, 32, 33, 47, 48, 60, 63, 64, 65, 70, 80, 95, 96, 97, 200, 1000]))
                seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n)].copy()
                style = case % 5
                if style == 1 and n:  # sprinkled non-bases
                    at = rng.integers(0, n, max(1, n // 20))
                    seq[at] = np.frombuffer(b"NnRYKM- \t*", dtype=np.uint8)[rng.integers(0, 10, len(at))]
                if style == 2 and n > 40:  # an N run inside a clean line
                    a = int(rng.integers(0, n - 30))
                    seq[a : a + int(rng.integers(1, 30))] = ord("N")
                if style == 3:  # lower case
                    seq = np.where(rng.random(n) < 0.5, seq | 0x20, seq).astype(np.uint8)
                parts.append(seq.tobytes() + (b"\r\n" if style == 4 else b"\n"))
        text = b"".join(parts)
        if case % 3 == 0 and text.endswith(b"\n"):
            text = text[:-1]
        texts.append(text)
    for text in texts:
        cap = int(lib.pa_pack_bound(len(text)))
        packed, mask = np.full(cap // 16, 0xDEADBEEF, dtype=np.uint32), np.full(cap // 32, 0xDEADBEEF, dtype=np.uint32)
        nb, nres, nrec, ninv = (C.c_uint64() for _ in range(4))
        _capi.check(lib.pa_pack_fasta(text, len(text), packed.ctypes.data, mask.ctypes.data, cap, C.byref(nb), C.byref(nres),
                                      C.byref(nrec), C.byref(ninv)), "pa_pack_fasta")
        t_packed, t_mask, t_res, t_rec, t_inv, t_start, t_len = _pack_truth(text)
        assert (nb.value, nres.value, nrec.value, ninv.value) == (16 * len(t_packed), t_res, t_rec, t_inv), text[:80]
        assert np.array_equal(packed[: len(t_packed)], t_packed) and np.array_equal(mask[: len(t_mask)], t_mask), text[:80]
        rs, rl = np.zeros(max(t_rec, 1), dtype=np.uint64), np.zeros(max(t_rec, 1), dtype=np.uint64)
        assert lib.pa_fasta_records(text, len(text), rs.ctypes.data, rl.ctypes.data, t_rec) == t_rec
        assert rs[:t_rec].tolist() == t_start and rl[:t_rec].tolist() == t_len
    # the same texts as files through the loader (checksums side by side, record tables from the packing pass)
    paths = []
    for i, text in enumerate(texts):
        paths.append(tmp_path / f"t{i:02d}.fasta")
        paths[-1].write_bytes(text)
    from tests.helpers import md5_hex

    for threads in (1, 3):
        infos, arena = load_fasta_files(paths, threads=threads)
        assert [i.status for i in infos] == [0] * len(texts)
        assert [i.md5 for i in infos] == [md5_hex(t) for t in texts]
        ref = pack_genomes(texts)
        assert np.array_equal(arena.packed, ref.packed) and np.array_equal(arena.mask, ref.mask)
        assert np.array_equal(arena.contig_start, ref.contig_start) and np.array_equal(arena.contig_len, ref.contig_len)
        assert np.array_equal(arena.contig_genome, ref.contig_genome) and arena.residues == ref.residues


def test_packer_on_arbitrary_bytes():
    """Property test: whatever the bytes (binary junk, '>' inside lines, lone CRs, NULs), pa_pack_fasta -- vector and
    scalar chunks alike -- equals the character-by-character statement of the rules, and so does the record table."""
    import ctypes as C

    from hypothesis import given, settings
    from hypothesis import strategies as st

    lib = _capi.load_library()
    alphabet = st.sampled_from([b"A", b"C", b"G", b"T", b"a", b"c", b"g", b"t", b"N", b"n", b">", b"\n", b"\r", b" ", b"\t", b"\x00", b"-", b"X", b"\xff"])
    runs = st.builds(lambda ch, n: ch * n, st.sampled_from([b"A", b"C", b"G", b"T", b"ACGT", b"acgtn", b"N", b"ACGTACGTACGTACGTACGTACGTACGTACGTACGTACGT"]), st.integers(1, 70))
    piece = st.one_of(alphabet, runs, st.just(b"\n>title line\n"), st.binary(min_size=0, max_size=8))

    @settings(max_examples=300, deadline=None)
    @given(st.lists(piece, min_size=0, max_size=40))
    def check(pieces):
        text = b"".join(pieces)
        cap = int(lib.pa_pack_bound(len(text)))
        packed, mask = np.full(cap // 16, 0xDEADBEEF, dtype=np.uint32), np.full(cap // 32, 0xDEADBEEF, dtype=np.uint32)
        nb, nres, nrec, ninv = (C.c_uint64() for _ in range(4))
        status = lib.pa_pack_fasta(text or None, len(text), packed.ctypes.data, mask.ctypes.data, cap, C.byref(nb), C.byref(nres), C.byref(nrec), C.byref(ninv))
        assert status == 0
        t_packed, t_mask, t_res, t_rec, t_inv, t_start, t_len = _pack_truth(text)
        assert (nb.value, nres.value, nrec.value, ninv.value) == (16 * len(t_packed), t_res, t_rec, t_inv)
        assert np.array_equal(packed[: len(t_packed)], t_packed) and np.array_equal(mask[: len(t_mask)], t_mask)
        rs, rl = np.zeros(max(t_rec, 1), dtype=np.uint64), np.zeros(max(t_rec, 1), dtype=np.uint64)
        assert lib.pa_fasta_records(text or None, len(text), rs.ctypes.data, rl.ctypes.data, t_rec) == t_rec
        assert rs[:t_rec].tolist() == t_start and rl[:t_rec].tolist() == t_len

    check()


def test_fastani_worker_with_an_unreadable_input_ends_through_log_sys_exit(tmp_path):
    """A FASTA file that has gone missing (or is not FASTA) ends the fastANI-hip worker with a message naming it -- what a
    failing fastANI process is to the reference (private_cli.py:1044-1063 through utils.check_output)."""
    from pyani_plus_amd.methods import fastani_hip

    name = "viral_example"
    _scaled, genomes = FIXTURE_SETS[name]
    run = _make_run(GOLDEN / name, genomes, 300, method=fastani_hip.METHOD)
    tool = fastani_hip.get_fastani_hip()
    run.configuration.program, run.configuration.version = tool.exe_path.stem, tool.version
    run.configuration.fragsize, run.configuration.kmersize, run.configuration.minmatch = 3000, 16, 0.2
    hash_to_filename = {h: f for f, h in genomes.items()}
    first = sorted(hash_to_filename)[0]
    hash_to_filename[first] = "gone.fasta"
    with pytest.raises(SystemExit, match="fastANI-hip comparison failed: .*gone.fasta"):
        fastani_hip.compute_fastani_hip(LOGGER, tmp_path, _Session(), run, tmp_path / "f.json", GOLDEN / name, hash_to_filename, {},
                                        {h: 1000 for h in hash_to_filename}, "", engine=OracleEngine())


def test_damaged_signature_in_the_cache_ends_the_worker(tmp_path):
    """A truncated, foreign or wrong-k `.sig` in the cache: the worker exits through log_sys_exit naming the file, as
    the reference's `sourmash sig collect` step would (methods/sourmash.py:170-183), instead of a raw traceback."""
    name = "viral_example"
    scaled, genomes = FIXTURE_SETS[name]
    run = rundb.run_sourmash_hip(GOLDEN / name, tmp_path / "ok.sqlite", cache=tmp_path / "cache", scaled=scaled, engine=OracleEngine(), temp=tmp_path)
    sig_dir = tmp_path / "cache" / f"sourmash_k=31_scaled={scaled}"
    victim = sorted(sig_dir.glob("*.sig"))[0]
    good = victim.read_bytes()
    hashes = {a.genome_hash: 1000 for a in run.fasta_hashes}
    other_k = json.loads(good)
    other_k[0]["signatures"][0]["ksize"] = 21
    for damage in (good[: len(good) // 2], b"not json at all", b"[]", json.dumps(other_k).encode()):
        victim.write_bytes(damage)
        with pytest.raises(SystemExit, match="Unreadable sourmash signature file"):
            sourmash_hip.compute_sourmash_hip(LOGGER, tmp_path, _Session(), run, tmp_path / "x.json", GOLDEN / name, {}, {}, hashes, "",
                                              cache=tmp_path / "cache", engine=OracleEngine())
    victim.write_bytes(good)


def test_driver_with_a_long_kmer_size(tmp_path):
    """--kmersize 51 (sourmash's third default; the reference passes any size on, public_cli_args.py:229) through the
    whole host side: cache directory name, signature files with ksize 51, complete database."""
    name = "viral_example"
    scaled, genomes = FIXTURE_SETS[name]
    run = rundb.run_sourmash_hip(GOLDEN / name, tmp_path / "k51.sqlite", cache=tmp_path / "cache", kmersize=51, scaled=scaled,
                                 engine=OracleEngine(), temp=tmp_path, ingest="direct")
    sigs = sorted((tmp_path / "cache" / f"sourmash_k=51_scaled={scaled}").glob("*.sig"))
    assert len(sigs) == len(genomes) == len(run.fasta_hashes)
    by_hash = {a.genome_hash: a.fasta_filename for a in run.fasta_hashes}
    for path in sigs:
        data = load_sig(path)
        assert data["signatures"][0]["ksize"] == 51
        want, _total = oracle.sketch_fasta_text(read_fasta_bytes(GOLDEN / name / by_hash[path.stem]), 51, scaled)
        assert np.array_equal(np.array(data["signatures"][0]["mins"], dtype=np.uint64), want)
    conn = sqlite3.connect(tmp_path / "k51.sqlite")
    assert conn.execute("SELECT kmersize FROM configurations").fetchall() == [(51,)]
    assert conn.execute("SELECT COUNT(*) FROM comparisons WHERE identity = 1.0 AND query_hash = subject_hash").fetchone()[0] == len(genomes)
    conn.close()


def test_host_native_code_under_sanitizers():
    """AddressSanitizer + UBSan over the host-side native code (no sanitizer runs exist on the GPU pool): the inflate
    decoder on thousands of damaged streams in exact-size heap buffers, the AVX2 packer against the scalar one on
    random texts, sixteen-lane md5 against the one-message form."""
    import shutil
    import subprocess

    if shutil.which("g++") is None:
        pytest.skip("no host compiler")
    done = subprocess.run(["bash", str(ROOT / "tests" / "tools" / "sanitize" / "run.sh"), "2000"], capture_output=True, text=True, timeout=900)
    assert done.returncode == 0 and "sanitizer runs clean" in done.stdout, done.stdout[-2000:] + done.stderr[-2000:]
    assert "ACCEPTED WRONG DATA" not in done.stdout and "MISMATCH" not in done.stdout


def test_mask_runs_across_chunks():
    """pa_mask_runs scans the mask in chunks on the host pool: runs that cross a chunk boundary (here: a run over the
    middle of the arena, where two workers meet) come out joined, in order, with the count right when cap is small."""
    lib = _capi.load_library()
    bases = 150_000_000 // 64 * 64  # 4.7M mask words: two workers
    mask = np.zeros(bases // 32, dtype=np.uint32)
    bits = np.unpackbits(mask.view(np.uint8), bitorder="little")
    rng = np.random.default_rng(8)
    truth = []
    for start in sorted(rng.integers(0, bases - 5000, 40).tolist()) + [bases // 2 - 100_000]:
        length = int(rng.integers(1, 4000)) if start != bases // 2 - 100_000 else 300_000
        bits[start : start + length] = 1
    bits[-1] = 1
    edges = np.flatnonzero(np.diff(np.concatenate([[0], bits, [0]]).astype(np.int8)))
    truth = list(zip(edges[0::2].tolist(), (edges[1::2] - edges[0::2]).tolist()))
    mask = np.packbits(bits, bitorder="little").view(np.uint32)
    starts, lens = np.zeros(64, dtype=np.uint64), np.zeros(64, dtype=np.uint64)
    n = lib.pa_mask_runs(mask.ctypes.data, bases, starts.ctypes.data, lens.ctypes.data, 64)
    assert n == len(truth) and list(zip(starts[:n].tolist(), lens[:n].tolist())) == truth
    assert any(s < bases // 2 < s + l for s, l in truth)  # the run over the chunk boundary
    assert lib.pa_mask_runs(mask.ctypes.data, bases, starts.ctypes.data, lens.ctypes.data, 3) == len(truth)
    assert list(zip(starts[:3].tolist(), lens[:3].tolist())) == truth[:3]
    assert lib.pa_mask_runs(mask.ctypes.data, bases, None, None, 0) == len(truth)


def test_host_pool_serves_two_callers_at_once(tmp_path):
    """The FASTA loader runs on a background thread of the batched front-end while the main thread uses the same
    host pool (strict ANI, writers).  Two callers at once must both get correct results."""
    import threading

    from pyani_plus_amd.engine import load_fasta_files
    from tests.helpers import md5_hex

    rng = np.random.default_rng(2)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    texts = [b">g%d\n" % i + acgt[rng.integers(0, 4, 20_000)].tobytes() + b"\n" for i in range(24)]
    paths = []
    for i, text in enumerate(texts):
        paths.append(tmp_path / f"g{i}.fna")
        paths[-1].write_bytes(text)
    counts = rng.integers(0, 50, size=(300, 300)).astype(np.uint32)
    sizes = rng.integers(50, 5000, size=300).astype(np.uint64)
    want = ani_host(counts, sizes, sizes, K, threads=1)
    errors = []

    def loader():
        try:
            for _ in range(15):
                infos, _arena = load_fasta_files(paths, threads=4)
                assert [i.md5 for i in infos] == [md5_hex(t) for t in texts]
        except Exception as err:  # noqa: BLE001
            errors.append(err)

    def transform():
        try:
            for _ in range(60):
                got = ani_host(counts, sizes, sizes, K, threads=4)
                for a, b in zip(got, want):
                    assert np.array_equal(a, b, equal_nan=True)
        except Exception as err:  # noqa: BLE001
            errors.append(err)

    workers = [threading.Thread(target=loader), threading.Thread(target=transform)]
    for w in workers:
        w.start()
    for w in workers:
        w.join()
    assert not errors, errors


def _pa_gunzip(data: bytes, decoder: int, cap: int | None = None):
    import ctypes as C

    lib = _capi.load_library()
    n = C.c_uint64()
    cap = max(64, len(data) * 1100) if cap is None else cap
    out = np.empty(cap, dtype=np.uint8)
    status = lib.pa_gunzip(data, len(data), out.ctypes.data, cap, C.byref(n), decoder)
    return status, (out[: n.value].tobytes() if status == 0 else None)


def test_own_inflate_equals_zlib():
    """inflate_fast.h (decoder 1) against what zlib wrote: every block type and strategy zlib can be made to emit,
    literal-only and match-heavy data, headers with names, flushes, several members, zero padding -- and damaged
    streams, where the loader's route (decoder 0) must accept exactly what Python's gzip module accepts."""
    import gzip
    import io
    import zlib

    rng = np.random.default_rng(5)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)

    def dna(n: int, width: int = 80) -> bytes:
        seq = acgt[rng.integers(0, 4, n, dtype=np.uint8)].tobytes()
        return b">g\n" + b"\n".join(seq[i : i + width] for i in range(0, n, width)) + b"\n"

    payloads = {
        "empty": b"", "one": b"A", "dna_small": dna(1000), "dna": dna(400_000), "dna_one_line": dna(100_000, 100_000),
        "zeros": bytes(300_000), "random": rng.integers(0, 256, 200_000, dtype=np.uint8).tobytes(),
        "text": b"the quick brown fox jumps over the lazy dog. " * 5000, "period3": b"abc" * 50_000,
        "two_letters": acgt[rng.integers(0, 2, 100_000, dtype=np.uint8)].tobytes(),
    }  # fmt: skip
    for name, data in payloads.items():
        for level in (0, 1, 6, 9):
            for strategy in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE):
                comp = zlib.compressobj(level, zlib.DEFLATED, 31, 8, strategy)
                status, out = _pa_gunzip(comp.compress(data) + comp.flush(), 1)
                assert status == 0 and out == data, (name, level, strategy, _capi.last_error())
        buf = io.BytesIO()
        with gzip.GzipFile(filename="some_name.fasta", mode="wb", fileobj=buf, mtime=0) as handle:
            handle.write(data)
        assert _pa_gunzip(buf.getvalue(), 1) == (0, data)
        comp = zlib.compressobj(6, zlib.DEFLATED, 31, 1)  # little memory: many dynamic blocks; flushes: empty stored blocks
        third = len(data) // 3
        parts = [comp.compress(data[:third]), comp.flush(zlib.Z_SYNC_FLUSH), comp.compress(data[third:]), comp.flush(zlib.Z_FULL_FLUSH), comp.flush()]
        assert _pa_gunzip(b"".join(parts), 1) == (0, data)
    a, b = payloads["dna_small"], payloads["text"]
    multi = gzip.compress(a) + gzip.compress(b) + gzip.compress(b"")
    for decoder in (0, 1, 2):
        assert _pa_gunzip(multi, decoder) == (0, a + b)
        assert _pa_gunzip(multi + bytes(512), decoder) == (0, a + b)  # zero padding, as Python's gzip accepts it
        assert _pa_gunzip(multi + b"junk", decoder)[0] != 0
    assert _pa_gunzip(multi, 0, cap=10)[0] == _capi.PA_E_CAPACITY
    # damage: never a wrong answer; the loader's route agrees with Python's gzip module on what is readable
    good = gzip.compress(payloads["dna"][:60_000])
    for trial in range(150):
        bad = bytearray(good)
        if trial % 3 == 0:
            bad[int(rng.integers(10, len(bad)))] ^= 1 << int(rng.integers(0, 8))
        elif trial % 3 == 1:
            bad = bad[: int(rng.integers(1, len(bad)))]
        else:
            at = int(rng.integers(10, len(bad) - 8))
            bad[at : at + 8] = rng.integers(0, 256, 8, dtype=np.uint8).tobytes()
        try:
            want = gzip.decompress(bytes(bad))
        except Exception:  # noqa: BLE001 - zlib.error, EOFError, BadGzipFile, ...
            want = None
        status, out = _pa_gunzip(bytes(bad), 1)
        assert status != 0 or out == payloads["dna"][:60_000]
        status, out = _pa_gunzip(bytes(bad), 0)
        assert (status == 0) == (want is not None) and out == want, trial
    # the reference's own compressed fixtures
    for path in sorted((GOLDEN / "bacterial_example").glob("*.gz")):
        raw = path.read_bytes()
        assert _pa_gunzip(raw, 1) == (0, gzip.decompress(raw))


@pytest.mark.parametrize("name", list(FIXTURE_SETS))
def test_manysearch_csv_export_equals_fixture_text(name, tmp_path):
    """All 15 columns of the reference's intermediate manysearch.csv, as text, from counts + sizes."""
    from pyani_plus_amd.sig import signature_md5
    from pyani_plus_amd.wire import _rust_float, export_manysearch_csv

    scaled, genomes = FIXTURE_SETS[name]
    md5s = sorted(genomes)
    sketches = [np.array(load_sig(GOLDEN / name / "sourmash" / f"{m}.sig")["signatures"][0]["mins"], dtype=np.uint64) for m in md5s]
    sig_md5 = [signature_md5(31, s) for s in sketches]
    for m, got in zip(md5s, sig_md5):
        assert got == load_sig(GOLDEN / name / "sourmash" / f"{m}.sig")["signatures"][0]["md5sum"]
    counts = oracle.pair_counts(sketches)
    sizes = [len(s) for s in sketches]
    out = tmp_path / "manysearch.csv"
    n_rows = export_manysearch_csv(out, md5s, sig_md5, md5s, sig_md5, counts, sizes, sizes, 31, scaled)
    got_lines = out.read_text().splitlines()
    want_lines = (GOLDEN / name / "sourmash" / "manysearch.csv").read_text().splitlines()
    assert got_lines[0] == want_lines[0]
    assert sorted(got_lines[1:]) == sorted(line for line in want_lines[1:] if line)  # upstream row order is thread-dependent
    assert n_rows == len(got_lines) - 1 == int((counts > 0).sum())
    assert _rust_float(5e-05) == "0.00005" and _rust_float(1.0) == "1.0" and _rust_float(1e-7) == "0.0000001"


def test_mask_runs_round_trip():
    from pyani_plus_amd.engine import mask_runs

    rng = np.random.default_rng(3)
    bits = np.zeros(64 * 200, dtype=np.uint8)
    for _ in range(60):
        a = int(rng.integers(0, bits.size))
        bits[a : a + int(rng.integers(1, 130))] = 1
    bits[:3] = 1
    bits[-70:] = 1
    mask = (bits.reshape(-1, 32).astype(np.uint64) << np.arange(32, dtype=np.uint64)[None, :]).sum(axis=1).astype(np.uint32)
    start, length = mask_runs(mask, bits.size)
    rebuilt = np.zeros_like(bits)
    for s, n in zip(start, length):
        assert n > 0 and not rebuilt[int(s) : int(s + n)].any()
        rebuilt[int(s) : int(s + n)] = 1
    assert np.array_equal(rebuilt, bits)
    assert np.all(start[1:] > start[:-1] + length[:-1])  # maximal runs: never adjacent
    assert mask_runs(np.zeros(4, dtype=np.uint32), 128)[0].size == 0


def test_native_sig_writer_equals_python_writer_and_fixture(tmp_path):
    """pa_write_sigs (threaded, native decimals + md5) writes the same bytes as sig.write_sig, which writes the
    same bytes as `sourmash scripts singlesketch` (fixture .sig files)."""
    rng = np.random.default_rng(1)
    sketches = [np.sort(rng.integers(0, 2**64 - 1, size=n, dtype=np.uint64)) for n in (0, 1, 5000, 17)]
    names = ["a", 'b"q', "ü-name", "d"]
    files = ["/x/y.fa", "C:\\\\p\\q.fa", 'we"ird "mins":[],"md5sum":"" name.fa', "ŝ.fasta"]
    for i, (mins, name, filename) in enumerate(zip(sketches, names, files)):
        sig.write_sig(tmp_path / f"py{i}.sig", name=name, filename=filename, ksize=31, max_hash=123456789, mins=mins)
    sig.write_sigs([tmp_path / f"c{i}.sig" for i in range(4)], names=names, filenames=files, ksize=31, max_hash=123456789, sketches=sketches)
    for i in range(4):
        assert (tmp_path / f"c{i}.sig").read_bytes() == (tmp_path / f"py{i}.sig").read_bytes()
    # and against a reference fixture, through its own name / filename strings
    name = "viral_example"
    md5 = sorted(FIXTURE_SETS[name][1])[0]
    fixture = GOLDEN / name / "sourmash" / f"{md5}.sig"
    want = json.loads(fixture.read_text())[0]
    sketch = want["signatures"][0]
    sig.write_sigs(
        [tmp_path / "fixture.sig"], names=[want["name"]], filenames=[want["filename"]], ksize=sketch["ksize"],
        max_hash=sketch["max_hash"], sketches=[np.array(sketch["mins"], dtype=np.uint64)],
    )  # fmt: skip
    assert (tmp_path / "fixture.sig").read_bytes() == fixture.read_bytes()


# ------------------------------------------------------------------ measurement plumbing that needs no GPU
def test_bench_dry_run_plan_of_the_eight_gpu_configs():
    """``bench.py --gpus 8 --dry-run-plan``: the 8-rank run on paper (no GPU, no torch).  The per-rank shards, the padded
    all-gather sizes and the strong_basis memory need follow from the same shard arithmetic the ranks use."""
    import subprocess
    import sys

    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "8", "--dry-run-plan"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    plan = json.loads(out.stdout)
    assert plan["world"] == 8 and plan["genomes"] == 10_000 and len(plan["ranks"]) == 8
    assert [r["genomes"] for r in plan["ranks"]] == [[1250 * i, 1250 * (i + 1)] for i in range(8)]
    pay = plan["collectives_per_step"][1]
    assert pay["bytes_per_rank"] == plan["padded_payload_hashes"] * 8 >= 1250 * 5000 * 8
    for r in plan["ranks"]:
        assert r["allgather_payload_bytes_received"] == 8 * pay["bytes_per_rank"]
        assert r["torch_cat_after_gather"] and r["payload_staging_copy"]  # FracMinHash totals differ between ranks: both are taken
        assert r["device_bytes_estimate"] < 0.05 * plan["hbm_bytes"]
    basis = plan["strong_basis"]
    assert basis["fits_hbm"] and basis["within_watchdog"] and basis["full_arena_bytes"] > 18e9
    assert json.loads((ROOT / "profiles" / "r05_plan_config2_8gpu.json").read_text())["ranks"] == plan["ranks"]  # the committed plan is this one
    mixed = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "8", "--genomes", "2000", "--mixed-lengths", "--dry-run-plan"],
                           capture_output=True, text=True, timeout=120)
    plan4 = json.loads(mixed.stdout)
    assert sum(r["n_genomes"] for r in plan4["ranks"]) == 2000 and plan4["shard_balance_bases_max_over_mean"] < 1.02


def test_work_based_roofline_is_reproducible_from_the_committed_profiles():
    """Every ``also.fragment_ani.roofline*.frac`` of the bench line follows from files under profiles/ and from nothing else:
    ``profiles/fragani_counters.json`` is what ``tools/pmc_fragani_to_json.py r06`` makes of the committed counter summaries,
    the per-phase instruction counts (the mapping kernel cut short after each phase under ``--pmc``), the event counts of the
    stats build and the FETCH_SIZE calibration -- and the fractions are re-derived here by hand from those numbers."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("pmc_fragani_to_json", ROOT / "tools" / "pmc_fragani_to_json.py")
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    prof = ROOT / "profiles"
    committed = json.loads((prof / "fragani_counters.json").read_text())
    rebuilt = tool.build(committed["tag"], prof)
    assert json.loads(json.dumps(rebuilt)) == committed  # the committed file is what the tool makes of the committed measurements
    tag = committed["tag"]
    # ---- map_segments_kernel: needed = V[3] + (V[24] - V[3]) x share, issued = V[9]
    cuts = tool.parse_cuts(prof / f"{tag}_map_cut_valu.txt")
    v = {c: d["SQ_INSTS_VALU"] for c, d in cuts.items()}
    ev = tool.parse_events(prof / f"{tag}_fragani_n1000_one_batch_trace.txt")
    share = (ev["candidates"] * 237.0 + ev["tying_states"]) / (ev["candidates"] * ev["stretch_entries_ranked"] / (ev["rounds"] - ev["rounds_ended_by_the_tight_bound"]))
    work = committed["map_segments_kernel"]["work"]
    assert 0.5 < share < 1.0 and abs(work["share_of_a_round_needed"] - share) < 1e-12
    assert abs(work["frac"] - (v[3] + (v[24] - v[3]) * share) / v[9]) < 1e-12 and 0.4 < work["frac"] < 0.8
    assert abs(work["frac_first_group"] - v[23] / v[9]) < 1e-12 and work["frac"] < work["frac_first_group"] < 1.0
    assert v[10] < v[11] < v[1] < v[2] < v[3] < v[5] < v[7] < v[8] < v[24] < v[23] < v[9]  # every cut ends the kernel later than the one before
    # the one-run segments' L1 scan is not charged: the L1 phase is what the kernel issues, with the scan skipped where it is
    assert ev["one_run_segments_no_l1_scan"] > 0.9 * ev["segments"] and work["valu_instructions_per_phase_per_segment"]["l1"] < 400
    # hits ordered by counting for (nearly) all segments: pricing every segment at the counting sort moves the fraction by < 0.01
    assert abs(work["frac_minimal_sort"] - work["frac"]) < 0.01 and work["counting_sort_valu_per_segment"] < 0.5 * work["network_sort_valu_per_segment"]
    # the product build's own counter pass agrees with the tools build's uncut run within the cut checks' few instructions
    product = tool.parse(prof / f"{tag}_pmc_map_segments_summary.txt")["SQ_INSTS_VALU"]
    assert committed["map_segments_kernel"]["valu_instructions"] == product and abs(product / v[9] - 1.0) < 0.03
    # ---- map_sparse_kernel: groups that hold a candidate's first or last tying begin / groups evaluated
    sp = committed["map_sparse_kernel"]
    se = ev["sparse"]
    assert abs(sp["frac"] - se["groups_with_the_first_or_last_tie"] / se["groups"]) < 1e-12 and 0.3 < sp["frac"] <= 1.0 and sp["valu_busy"] > 0.8
    assert se["candidates"] <= se["groups_with_the_first_or_last_tie"] <= 2 * se["candidates"]  # one or two per candidate
    # ---- minimizer_kernel: the two hashes per position / instructions issued per position
    mi = committed["minimizer_kernel"]
    issued = tool.parse(prof / f"{tag}_pmc_minimizer_summary.txt")["SQ_INSTS_VALU"] * 64 / (1000 * 5_000_064)
    assert abs(mi["frac"] - 126.0 / issued) < 1e-12 and 150 < issued < 230
    # ---- bucket_hits_staged_kernel: 18 algorithmic bytes per hit / time against 8 TB/s; traffic = calibrated FETCH_SIZE + WRITE_SIZE
    bh = committed["bucket_hits_kernel"]
    summary = tool.parse(prof / f"{tag}_pmc_bucket_hits_summary.txt")
    hits = ev["seed_hits_of_the_batch"]
    assert abs(bh["algorithmic_gbs"] - 18.0 * hits / (summary["duration_ms"] * 1e-3) / 1e9) < 1e-6
    cal = tool.parse_calibration(prof / f"{tag}_fetch_calibration.txt")
    assert abs(cal["calib_stream16"]["counted_over_asked"] - 0.5) < 0.01  # the guide's rule, reproduced in the same run
    k16, k64 = cal["calib_runs<unsigned short>"], cal["calib_runs<unsigned long>"]
    c16, c64 = 2.0 * k16["counted_over_asked"], 8.0 * k64["counted_over_asked"]
    factor = (c16 / k16["counted_over_line_bytes"] + c64 / k64["counted_over_line_bytes"]) / (c16 + c64)
    traffic = (summary["FETCH_SIZE"] * 1024 * factor + summary["WRITE_SIZE"] * 1024) / hits
    assert abs(bh["counter_bytes_per_hit"] - traffic) < 1e-9 and 18.0 < traffic < 40.0
    # what bench.py prints is this file's content (the roofline entries copy these fields)
    bench_src = (ROOT / "bench.py").read_text()
    for key in ("frac_first_group", "frac_minimal_sort", "roofline_sparse", "roofline_index", "traffic_over_algorithmic"):
        assert key in bench_src
