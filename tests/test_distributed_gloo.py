"""world_size-2 CPU test (gloo) of the multi-GPU plumbing: shards, sketch all-gather, column tiles."""

from __future__ import annotations

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from pyani_plus_amd.distributed import allgather_sketches, shard_bounds, shard_bounds_by_cost
from pyani_plus_amd.synth import arena_to_ascii, synth_arena_numpy

LENGTHS = [9000, 300, 12000, 64, 7000, 0, 5000, 8000, 2500]
K, SCALED = 21, 20


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, out_dir: str) -> None:
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        arena = synth_arena_numpy(len(LENGTHS), LENGTHS, n_species=2)
        bounds = shard_bounds(len(LENGTHS), world)
        g0, g1 = bounds[rank]
        # local sketch of this rank's genomes (oracle stands in for the GPU here)
        local = [oracle.sketch_seq(arena_to_ascii(arena, g), K, SCALED) for g in range(g0, g1)]
        sizes = torch.tensor([len(s) for s in local], dtype=torch.int64)
        flat = np.concatenate(local) if local else np.zeros(0, np.uint64)
        hashes = torch.from_numpy(flat.view(np.int64).copy()) if flat.size else torch.zeros(1, dtype=torch.int64)
        all_hashes, off, off_host = allgather_sketches(torch, dist, hashes, sizes, [b - a for a, b in bounds])
        assert np.array_equal(off_host.astype(np.int64), off.numpy())
        off_np = off.numpy()
        gathered = [all_hashes.numpy().view(np.uint64)[off_np[g] : off_np[g + 1]] for g in range(len(LENGTHS))]
        counts = oracle.pair_counts(gathered, (0, len(LENGTHS)), (g0, g1))  # this rank's subject columns
        np.save(os.path.join(out_dir, f"counts_{rank}.npy"), counts)
        np.save(os.path.join(out_dir, f"off_{rank}.npy"), off_np)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])  # 8: the node size of BASELINE configs[2]; ranks with one genome each
def test_allgather_and_column_tiles(tmp_path, world):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    arena = synth_arena_numpy(len(LENGTHS), LENGTHS, n_species=2)
    full = [oracle.sketch_seq(arena_to_ascii(arena, g), K, SCALED) for g in range(len(LENGTHS))]
    want = oracle.pair_counts(full)
    got = np.concatenate([np.load(tmp_path / f"counts_{r}.npy") for r in range(world)], axis=1)
    assert np.array_equal(got, want)
    want_off = np.concatenate([[0], np.cumsum([len(s) for s in full])])
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"off_{r}.npy"), want_off)


class _TorchSpy:
    """The ``torch`` module as ``allgather_sketches`` takes it, counting the calls that tell its two payload paths apart."""

    def __init__(self):
        self.cat_calls = 0
        self.payload_staged = 0

    def __getattr__(self, name):
        return getattr(torch, name)

    def cat(self, *a, **kw):
        self.cat_calls += 1
        return torch.cat(*a, **kw)

    def zeros(self, *a, **kw):
        if a and isinstance(a[0], int) and a[0] > 64:  # the padded payload (the size vector is short, offsets are not made here)
            self.payload_staged += 1
        return torch.zeros(*a, **kw)


def _payload_worker(rank: int, world: int, port: int, out_dir: str, equal: bool) -> None:
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # 8 ranks; uneven genome counts per rank (3, 2, 3, 2, ...) in both cases.  equal: every rank's hashes add up to 600
        # (the gathered buffer IS the result, the payload is sent from the rank's own buffer); unequal: totals 100 .. 800
        # and own buffers exactly as long as the own total (padded staging copy on all ranks but the largest, torch.cat).
        shard_sizes = [3 if r % 2 == 0 else 2 for r in range(world)]
        per_rank_total = [600] * world if equal else [100 * (r + 1) for r in range(world)]
        n_local, total = shard_sizes[rank], per_rank_total[rank]
        sizes = np.full(n_local, total // n_local, dtype=np.int64)
        sizes[-1] += total - int(sizes.sum())
        g0 = sum(shard_sizes[:rank])
        payload = np.concatenate([np.arange(sz, dtype=np.int64) + ((g0 + i) << 32) for i, sz in enumerate(sizes)])
        spy = _TorchSpy()
        hashes, off, off_host = allgather_sketches(spy, dist, torch.from_numpy(payload), torch.from_numpy(sizes), shard_sizes)
        # every genome's slice holds its own numbers, in rank order
        n_total = sum(shard_sizes)
        assert len(off_host) == n_total + 1 and int(off_host[-1]) == sum(per_rank_total) == hashes.numel()
        assert np.array_equal(off.numpy().astype(np.uint64), off_host)
        flat = hashes.numpy()
        for g in range(n_total):
            seg = flat[int(off_host[g]) : int(off_host[g + 1])]
            assert np.array_equal(seg, np.arange(len(seg), dtype=np.int64) + (g << 32)), (rank, g)
        np.save(os.path.join(out_dir, f"paths_{rank}.npy"), np.array([spy.cat_calls, spy.payload_staged]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("equal", [True, False])
def test_allgather_payload_paths_at_world_8(tmp_path, equal):
    """Both payload paths of ``allgather_sketches`` (distributed.py: padded staging copy + ``torch.cat`` when the per-rank totals
    differ, the gathered buffer as it is when they are equal) at the node size of BASELINE configs[2], with uneven genome
    counts per rank.  The RCCL form of the same calls has never run with more than one rank (no multi-GPU node so far)."""
    world = 8
    mp.spawn(_payload_worker, args=(world, _free_port(), str(tmp_path), equal), nprocs=world, join=True)
    paths = np.stack([np.load(tmp_path / f"paths_{r}.npy") for r in range(world)])
    if equal:
        assert not paths.any()  # no torch.cat, no staging copy on any rank
    else:
        assert (paths[:, 0] == 1).all()  # every rank cuts and joins the gathered buffer
        assert paths[:-1, 1].all() and paths[-1, 1] == 0  # all ranks but the one with the largest total pad their payload


def test_shard_bounds():
    assert shard_bounds(10, 3) == [(0, 4), (4, 7), (7, 10)]
    assert shard_bounds(2, 4) == [(0, 1), (1, 2), (2, 2), (2, 2)]
    assert shard_bounds(0, 2) == [(0, 0), (0, 0)]
    b = shard_bounds_by_cost([10, 1, 1, 1, 1, 10, 1, 1], 2)
    assert b[0][0] == 0 and b[-1][1] == 8 and all(x[1] == y[0] for x, y in zip(b, b[1:]))
    costs = np.array([10, 1, 1, 1, 1, 10, 1, 1], float)
    loads = [costs[a:b_].sum() for a, b_ in b]
    assert max(loads) <= 16  # 26 total: a contiguous split cannot beat 14/12, must not be worse than 16/10
    rng = np.random.default_rng(0)
    lens = rng.integers(100_000, 10_000_000, size=2000)
    b8 = shard_bounds_by_cost(lens, 8)
    loads = np.array([lens[a:b_].sum() for a, b_ in b8], float)
    assert loads.max() / loads.mean() < 1.01
