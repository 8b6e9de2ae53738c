"""Multi-GPU plumbing: genome shards, the sketch all-gather and subject tiles.

One process per GPU (``torch.distributed``; backend ``nccl`` is RCCL over xGMI
on ROCm, ``gloo`` in the CPU tests).  The path has exactly one exchange step
(SURVEY.md section 8e): after each rank has sketched its own genomes, every
rank needs every sketch, then evaluates its own subject columns locally.

The reference has no analogue -- its workers exchange results through JSON
files on a shared filesystem (pyani_plus/workflows/__init__.py:71-109) -- and
the sketch exchange replaces the `.sig` file lists handed to
``sourmash sig collect`` (pyani_plus/methods/sourmash.py:162-183).
"""

from __future__ import annotations

import numpy as np


def shard_bounds(n_items: int, world: int) -> list[tuple[int, int]]:
    """Contiguous, balanced-by-count split of ``range(n_items)`` over ``world`` ranks."""
    base, rem = divmod(n_items, world)
    bounds, start = [], 0
    for r in range(world):
        stop = start + base + (1 if r < rem else 0)
        bounds.append((start, stop))
        start = stop
    return bounds


def shard_bounds_by_cost(costs, world: int) -> list[tuple[int, int]]:
    """Contiguous split balancing cumulative cost (e.g. genome length) instead of count.

    Keeps genome order (rank r owns a contiguous index range, so the gathered
    CSR is simply the concatenation of the shards)."""
    costs = np.asarray(costs, dtype=np.float64)
    n = len(costs)
    if n == 0:
        return [(0, 0)] * world
    cum = np.concatenate([[0.0], np.cumsum(costs)])
    total = cum[-1]
    cuts = [0]
    for r in range(1, world):
        target = total * r / world
        idx = int(np.searchsorted(cum, target, side="left"))
        # choose the nearer boundary, never go backwards
        if idx > 0 and abs(cum[idx - 1] - target) <= abs(cum[min(idx, n)] - target):
            idx -= 1
        cuts.append(min(max(idx, cuts[-1]), n))
    cuts.append(n)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def allgather_sketches(torch, dist, local_hashes, local_sizes, shard_sizes: list[int], group=None):
    """All-gather variable-length CSR sketches.

    local_hashes: int64 tensor [>= sum(local_sizes)] (u64 bit patterns), this rank's
                  concatenated sketches; local_sizes: int64 tensor [n_local].
    shard_sizes:  number of genomes owned by each rank (known to all by construction).
    Returns (hashes int64 [total], off int64 [n_total+1]) with genomes in rank order.

    Two collectives: the per-genome sizes (padded to the largest shard) and the payload
    (padded to the largest per-rank total).  At N=10^4, |S|=5*10^3 the payload is 400 MB,
    i.e. ~50 MB per xGMI link: not worth a hand-rolled ring.
    """
    world = len(shard_sizes)
    dev = local_hashes.device
    max_n = max(shard_sizes) if shard_sizes else 0
    sizes_pad = torch.zeros(max(max_n, 1), dtype=torch.int64, device=dev)
    sizes_pad[: local_sizes.numel()] = local_sizes
    all_sizes = torch.empty(world * max(max_n, 1), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(all_sizes, sizes_pad, group=group)
    all_sizes = all_sizes.view(world, -1)
    totals = all_sizes.sum(dim=1)
    totals_host = totals.cpu().tolist()
    max_total = max(1, int(max(totals_host)))
    payload = torch.zeros(max_total, dtype=torch.int64, device=dev)
    local_total = int(local_sizes.sum().item()) if local_sizes.numel() else 0
    payload[:local_total] = local_hashes[:local_total]
    gathered = torch.empty(world * max_total, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(gathered, payload, group=group)
    gathered = gathered.view(world, max_total)
    hashes = torch.cat([gathered[r, : int(totals_host[r])] for r in range(world)]) if world else payload[:0]
    sizes = torch.cat([all_sizes[r, : shard_sizes[r]] for r in range(world)])
    off = torch.zeros(sizes.numel() + 1, dtype=torch.int64, device=dev)
    off[1:] = torch.cumsum(sizes, dim=0)
    if hashes.numel() == 0:
        hashes = torch.zeros(1, dtype=torch.int64, device=dev)
    return hashes.contiguous(), off
