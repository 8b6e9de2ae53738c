"""GPU: the multi-GPU code path (BASELINE configs[2]) exercised on ONE MI355X.

* ``bench.py --gpus 2`` launches its own ranks (fresh children of a parent that never touches the GPU);
  with ``PA_BENCH_BACKEND=gloo`` the two ranks share the box's single GPU and the collectives run on host
  tensors -- a plumbing check of exactly the code a SCALE run executes (reference analogue: one worker for
  the whole matrix, pyani_plus/public_cli.py:232-235).
* the RCCL (``nccl``) all-gather + column tile on DEVICE tensors with world size 1, including the
  dictionary build that overlaps the all-gather, against ``engine.pair_counts`` and the oracle.
* the 10 000-genome workload on one GPU: sampled sketches, the five-tile bit-row result against the merge
  kernel on a block that straddles a tile boundary, one block against ``oracle.pair_counts``.

This file sorts first among the GPU tests so that the bench launch happens before this pytest process has
initialised HIP itself.
"""

from __future__ import annotations

import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

import oracle
from pyani_plus_amd.synth import arena_to_ascii, synth_arena_numpy

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_bench(extra_env: dict, *argv: str) -> dict:
    env = dict(os.environ)
    env.update(extra_env)
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    proc = subprocess.run([sys.executable, str(ROOT / "bench.py"), *argv], env=env, capture_output=True, text=True, timeout=900)
    assert proc.returncode == 0, f"bench.py failed ({proc.returncode}):\n{proc.stdout[-2000:]}\n{proc.stderr[-4000:]}"
    lines = [x for x in proc.stdout.splitlines() if x.strip()]
    assert len(lines) == 1, f"expected ONE line on stdout, got {len(lines)}: {lines[:3]}"
    return json.loads(lines[0])


def test_bench_launches_its_own_ranks_two_ranks_on_one_gpu():
    res = _run_bench({"PA_BENCH_BACKEND": "gloo"}, "--gpus", "2", "--genomes", "64", "--length", "200000", "--steps", "2", "--warmup", "1")
    assert res["n_gpus"] == 2 and res["rccl_ranks"] == 2 and res["collective_backend"] == "gloo"
    assert res["config"]["genomes"] == 64 and res["config"]["genomes_per_gpu"] == 32
    assert res["value"] > 0 and res["scaling"] == "weak"
    assert len(res["shard_balance"]["busy_ms_per_step_by_rank"]) == 2
    # the same 64 genomes on one GPU, measured in the same run: what a speed-up has to be taken against
    assert res["strong_basis"]["genomes"] == 64 and res["strong_basis"]["one_gpu_pairs_per_s"] > 0
    assert res["speedup_vs_one_gpu_same_workload"] == pytest.approx(res["value"] / res["strong_basis"]["one_gpu_pairs_per_s"])


def test_bench_eight_ranks_on_one_gpu_with_the_one_gpu_basis():
    """The form the driver's SCALE run takes at N = 8 (`bench.py --gpus 8`: eight ranks, genome shards, the all-gather of
    unequal sketch payloads, column tiles, rank 0's one-GPU basis while seven ranks wait in the barrier), with the eight
    ranks sharing this box's one device and the collectives on host copies (gloo).  No scaling figure comes out of this --
    eight ranks on one GPU -- only that the eight-rank path runs to its end and its line is consistent."""
    res = _run_bench({"PA_BENCH_BACKEND": "gloo"}, "--gpus", "8", "--genomes", "72", "--length", "120000", "--steps", "2", "--warmup", "1")
    assert res["n_gpus"] == 8 and res["rccl_ranks"] == 8 and res["collective_backend"] == "gloo"
    assert res["config"]["genomes"] == 72 and res["config"]["genomes_per_gpu"] == 9
    assert res["value"] > 0 and res["scaling"] == "weak" and res["value"] == pytest.approx(72 * 72 / (res["ms_per_step"] * 1e-3))
    assert len(res["shard_balance"]["busy_ms_per_step_by_rank"]) == 8 and all(b > 0 for b in res["shard_balance"]["busy_ms_per_step_by_rank"])
    assert res["strong_basis"]["genomes"] == 72 and res["strong_basis"]["one_gpu_pairs_per_s"] > 0
    assert res["speedup_vs_one_gpu_same_workload"] == pytest.approx(res["value"] / res["strong_basis"]["one_gpu_pairs_per_s"])


def test_bench_rccl_path_on_one_rank():
    res = _run_bench({"PA_BENCH_FORCE_DIST": "1", "MASTER_PORT": str(_free_port())}, "--gpus", "1", "--genomes", "48", "--length", "300000",
                     "--steps", "2", "--warmup", "1", "--no-also")
    assert res["rccl_ranks"] == 1 and res["collective_backend"] == "nccl"
    assert "overlapped" in res["config"]["parallelism"]


def _dump(db):
    import sqlite3

    conn = sqlite3.connect(db)
    out = (
        conn.execute("SELECT genome_hash, length, description FROM genomes ORDER BY 1").fetchall(),
        conn.execute("SELECT query_hash, subject_hash, identity, aln_length, sim_errors, cov_query FROM comparisons ORDER BY 1, 2").fetchall(),
        conn.execute("SELECT status, df_identity, df_cov_query, df_aln_length, df_sim_errors, df_hadamard FROM runs").fetchall(),
    )
    conn.close()
    return out


def test_product_drivers_with_two_and_eight_ranks_sharing_the_gpu(tmp_path, monkeypatch):
    """The multi-GPU PRODUCT path on the device: ``rundb.run_sourmash_hip(gpus=2)`` and ``rundb.run_fastani_hip(gpus=2)``
    start two worker processes (before this process has initialised HIP), which share the box's one GPU -- collectives
    on host copies (gloo), kernels on the device -- and must give the databases the single-process drivers give."""
    import gzip
    import json

    from pyani_plus_amd import rundb

    monkeypatch.setenv("PYANI_HIP_DIST_BACKEND", "gloo")
    monkeypatch.delenv("PYANI_HIP_DEVICE", raising=False)
    lengths = [150_000, 40_000, 210_000, 64, 90_000, 120_000, 33_000, 175_000, 60_000]
    arena = synth_arena_numpy(len(lengths), lengths, n_species=2)
    indir = tmp_path / "in"
    indir.mkdir()
    for g in range(len(lengths)):
        seq = arena_to_ascii(arena, g)
        text = b">g%d a\n" % g + seq[: len(seq) // 3] + b"\n>g%d b\n" % g + seq[len(seq) // 3 :] + b"\n"
        if g % 2:
            (indir / f"g{g}.fna.gz").write_bytes(gzip.compress(text))
        else:
            (indir / f"g{g}.fasta").write_bytes(text)
    many = rundb.run_sourmash_hip(indir, tmp_path / "s2.sqlite", cache=tmp_path / "c2", scaled=100, temp=tmp_path / "t2", gpus=2)
    fmany = rundb.run_fastani_hip(indir, tmp_path / "f2.sqlite", temp=tmp_path / "tf2", gpus=2)
    # eight ranks (the node the reference's column fan-out would cover, pyani_plus/public_cli.py:232-261), all on this one device:
    # nine genomes over eight shards, eight column ranges, the tile placement and the watchdogs at world size 8
    many8 = rundb.run_sourmash_hip(indir, tmp_path / "s8.sqlite", cache=tmp_path / "c8", scaled=100, temp=tmp_path / "t8", gpus=8)
    fmany8 = rundb.run_fastani_hip(indir, tmp_path / "f8.sqlite", temp=tmp_path / "tf8", gpus=8)
    results8 = [json.loads(q.read_text()) for q in sorted((tmp_path / "t8" / "sourmash-hip.workers").glob("result_rank*.json"))]
    assert len(results8) == 8 and all(r["ok"] and r["device"].startswith("cuda") for r in results8)
    fresults8 = [json.loads(q.read_text()) for q in sorted((tmp_path / "tf8").glob("*.workers/result_rank*.json"))]
    # (nine subject columns dealt to eight ranks by length: a rank may be left without a column, and then reports no device)
    assert len(fresults8) == 8 and all(r["ok"] and r.get("device", "cuda").startswith("cuda") for r in fresults8)
    assert sum("device" in r for r in fresults8) >= 6
    # the same worker code over RCCL (backend nccl, collectives on device tensors) with the one rank this box has a GPU for
    monkeypatch.delenv("PYANI_HIP_DIST_BACKEND")
    monkeypatch.setenv("PYANI_HIP_FORCE_WORKERS", "1")
    rccl = rundb.run_sourmash_hip(indir, tmp_path / "s_rccl.sqlite", cache=tmp_path / "c_rccl", scaled=100, temp=tmp_path / "t_rccl", gpus=1)
    monkeypatch.delenv("PYANI_HIP_FORCE_WORKERS")
    monkeypatch.setenv("PYANI_HIP_DIST_BACKEND", "gloo")
    r1 = [json.loads(q.read_text()) for q in sorted((tmp_path / "t_rccl" / "sourmash-hip.workers").glob("result_rank*.json"))]
    assert len(r1) == 1 and r1[0]["ok"] and r1[0]["backend"] == "nccl" and rccl.status == "Done"
    results = [json.loads(q.read_text()) for q in sorted((tmp_path / "t2" / "sourmash-hip.workers").glob("result_rank*.json"))]
    assert len(results) == 2 and all(r["ok"] and r["device"].startswith("cuda") for r in results)
    fresults = [json.loads(q.read_text()) for q in sorted((tmp_path / "tf2").glob("*.workers/result_rank*.json"))]
    assert len(fresults) == 2 and all(r["ok"] and r["device"].startswith("cuda") for r in fresults)
    # the single-process drivers, in this process (which initialises HIP only now)
    one = rundb.run_sourmash_hip(indir, tmp_path / "s1.sqlite", cache=tmp_path / "c1", scaled=100, temp=tmp_path / "t1", ingest="direct")
    fone = rundb.run_fastani_hip(indir, tmp_path / "f1.sqlite", temp=tmp_path / "tf1")
    assert many.status == one.status == fmany.status == fone.status == "Done"
    assert _dump(tmp_path / "s1.sqlite") == _dump(tmp_path / "s2.sqlite") == _dump(tmp_path / "s_rccl.sqlite")
    assert _dump(tmp_path / "f1.sqlite") == _dump(tmp_path / "f2.sqlite")
    assert many8.status == fmany8.status == "Done"
    assert _dump(tmp_path / "s1.sqlite") == _dump(tmp_path / "s8.sqlite") and _dump(tmp_path / "f1.sqlite") == _dump(tmp_path / "f8.sqlite")
    rows = _dump(tmp_path / "f2.sqlite")[1]
    assert len(rows) == len(lengths) ** 2 and sum(r[2] is not None for r in rows) > len(lengths)  # related genomes do map
    # against the oracle: every sourmash pair
    sk = [oracle.sketch_fasta_text((gzip.decompress(q.read_bytes()) if q.suffix == ".gz" else q.read_bytes()), 31, 100)[0] for q in sorted(indir.iterdir())]
    counts = oracle.pair_counts(sk)
    sizes = [len(x) for x in sk]
    o_ident, o_cov, o_null = oracle.ani(counts, sizes, sizes, 31)
    import hashlib

    md5s = [hashlib.md5(gzip.decompress(q.read_bytes()) if q.suffix == ".gz" else q.read_bytes()).hexdigest() for q in sorted(indir.iterdir())]  # noqa: S324
    got = {(q, s): (i, c) for q, s, i, _a, _e, c in _dump(tmp_path / "s2.sqlite")[1]}
    for qi, q in enumerate(md5s):
        for si, s_ in enumerate(md5s):
            if o_null[qi, si]:
                assert got[(q, s_)] == (None, None)
            else:
                assert got[(q, s_)] == (o_ident[qi, si], o_cov[qi, si])


@pytest.fixture(scope="module")
def engine():
    from pyani_plus_amd.engine import HipEngine

    eng = HipEngine(0)
    yield eng
    eng.close()


@pytest.fixture(scope="module")
def nccl_world1():
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [True, False])
def test_rccl_allgather_and_column_tile_on_device_tensors(engine, nccl_world1, overlap):
    import torch

    from pyani_plus_amd import _capi
    from pyani_plus_amd.distributed import sharded_pair_step

    k, scaled = 31, 100
    lengths = [40_000, 25_000, 64, 40_000, 0, 31_000, 40_000, 12_345, 40_000, 30]
    arena = synth_arena_numpy(len(lengths), lengths, n_species=2)
    sk_local = engine.sketch(engine.upload(arena), k, scaled)
    n = len(lengths)
    sk, counts = sharded_pair_step(engine, torch, nccl_world1, sk_local, [n], (0, n), (0, n), backend="nccl", overlap=overlap)
    assert sk.hashes.is_cuda and sk.n == n and sk.total == sk_local.total
    want_sk = [oracle.sketch_seq(arena_to_ascii(arena, g), k, scaled) for g in range(n)]
    for a, b in zip(sk.to_host(), want_sk):
        assert np.array_equal(a, b)
    got = counts.cpu().numpy().view(np.uint32)
    assert np.array_equal(got, engine.pair_counts(sk_local).cpu().numpy().view(np.uint32))
    assert np.array_equal(got, engine.pair_counts(sk_local, algo=_capi.PA_PAIRS_MERGE).cpu().numpy().view(np.uint32))
    assert np.array_equal(got, oracle.pair_counts(want_sk))


def test_prepared_dictionary_must_match_the_tile(engine):
    from pyani_plus_amd._capi import HipBackendError

    arena = synth_arena_numpy(4, [20_000, 20_000, 20_000, 20_000], n_species=1)
    sk = engine.sketch(engine.upload(arena), 21, 50)
    engine.pair_dict_prepare(sk.hashes, sk.total)
    ok = engine.pair_counts(sk).cpu().numpy()
    assert np.array_equal(ok, engine.pair_counts(sk).cpu().numpy())  # the preparation is consumed by one call
    engine.pair_dict_prepare(sk.hashes, sk.total - 1)
    with pytest.raises(HipBackendError, match="prepared dictionary"):
        engine.pair_counts(sk)
    assert np.array_equal(ok, engine.pair_counts(sk).cpu().numpy())  # and dropped after the failure
    # the same NUMBER of postings but other hashes (another sketch set of the same total): refused by content
    other = sk.hashes.clone()
    other[: sk.total] = other[: sk.total] + 1
    engine.pair_dict_prepare(other, sk.total)
    with pytest.raises(HipBackendError, match="other postings"):
        engine.pair_counts(sk)
    assert np.array_equal(ok, engine.pair_counts(sk).cpu().numpy())
    # ... and a call of another phase in between (sketching uses the context's scalars) does not disturb a valid one
    engine.pair_dict_prepare(sk.hashes, sk.total)
    engine.sketch(engine.upload(arena), 21, 50)
    assert np.array_equal(ok, engine.pair_counts(sk).cpu().numpy())


def test_ten_thousand_genomes_on_one_gpu(engine):
    """BASELINE configs[2]'s genome count through the five-tile pair phase (genomes shortened to 100 kb so the
    synthetic set takes seconds to make; the tile logic depends on the count, not on the length)."""
    import torch

    from pyani_plus_amd import _capi
    from pyani_plus_amd.synth import synth_arena_torch

    n, length, k, scaled = 10_000, 100_000, 31, 1000
    arena = synth_arena_torch(engine, n, length, n_species=40)
    sk = engine.sketch(arena, k, scaled)
    counts = engine.pair_counts(sk)
    assert counts.shape == (n, n)
    off = sk.offsets_host().astype(np.int64)

    def host_sketch(g):
        return sk.hashes[int(off[g]) : int(off[g + 1])].cpu().numpy().view(np.uint64)

    # sampled sketches against the oracle (genome text unpacked on the GPU)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=engine.device)
    shifts = (torch.arange(16, device=engine.device, dtype=torch.int32) * 2)[None, :]
    for g in (0, 4_097, n - 1):
        s0 = int(arena.genome_start[g])
        words = arena.packed[s0 // 16 : s0 // 16 + (length + 15) // 16]
        seq = lut[((words[:, None] >> shifts) & 3).reshape(-1)[:length].to(torch.int64)].cpu().numpy().tobytes()
        assert np.array_equal(host_sketch(g), oracle.sketch_seq(seq, k, scaled)), f"genome {g}"
    # the five-tile bit-row result against the merge kernel on a block across the first tile boundary
    chk = engine.pair_counts(sk, (5_000, 5_128), (2_000, 2_100), algo=_capi.PA_PAIRS_MERGE)
    assert torch.equal(chk, counts[5_000:5_128, 2_000:2_100])
    chk = engine.pair_counts(sk, (0, 64), (8_150, 8_250), algo=_capi.PA_PAIRS_MERGE)
    assert torch.equal(chk, counts[0:64, 8_150:8_250])
    # one block against the oracle, rows and columns from different tiles
    rows, cols = list(range(40, 60)), list(range(6_140, 6_160))
    sub = [host_sketch(g) for g in rows + cols]
    want = oracle.pair_counts(sub)[: len(rows), len(rows) :]
    assert np.array_equal(counts[40:60, 6_140:6_160].cpu().numpy().view(np.uint32), want)
    # properties at full size: symmetric, diagonal = sketch sizes
    assert torch.equal(counts, counts.T)
    assert np.array_equal(counts.diagonal().cpu().numpy().astype(np.int64), off[1:] - off[:-1])
