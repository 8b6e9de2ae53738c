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
