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


def allgather_sketches(torch, dist, local_hashes, local_sizes, shard_sizes: list[int], group=None, *, while_in_flight=None):
    """All-gather variable-length CSR sketches.

    local_hashes: int64 tensor [>= sum(local_sizes)] (u64 bit patterns), this rank's
                  concatenated sketches; local_sizes: int64 tensor [n_local].
    shard_sizes:  number of genomes owned by each rank (known to all by construction).
    while_in_flight: optional callable, run after the payload all-gather has been started and
                  before it is waited for -- work that needs only this rank's own sketches
                  (``HipEngine.pair_dict_prepare``) overlaps the exchange that way.
    Returns (hashes int64 [total], off int64 [n_total+1] on the device, off_host uint64 [n_total+1])
    with genomes in rank order.

    Two collectives: the per-genome sizes (padded to the largest shard) and the payload
    (padded to the largest per-rank total).  Between them sits the path's ONE host round trip:
    the gathered sizes (n_total integers) come to the host, which gives every rank the padded
    payload length and the CSR offsets the pair phase wants on the host anyway.  The payload is
    sent straight from ``local_hashes`` when that buffer is long enough (no staging copy) and
    the gathered buffer is returned as it is when all ranks hold the same number of hashes.
    At N=10^4, |S|=5*10^3 the payload is 400 MB, i.e. ~50 MB per xGMI link: not worth a
    hand-rolled ring.
    """
    world = len(shard_sizes)
    dev = local_hashes.device
    max_n = max(max(shard_sizes) if shard_sizes else 0, 1)
    sizes_pad = torch.zeros(max_n, dtype=torch.int64, device=dev)
    sizes_pad[: local_sizes.numel()] = local_sizes
    all_sizes = torch.empty(world * max_n, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(all_sizes, sizes_pad, group=group)
    sizes_host = all_sizes.view(world, max_n).cpu().numpy()  # the one host round trip
    per_rank = [sizes_host[r, : shard_sizes[r]] for r in range(world)]
    totals = [int(x.sum()) for x in per_rank]
    max_total = max(1, max(totals))
    off_host = np.zeros(sum(shard_sizes) + 1, dtype=np.uint64)
    np.cumsum(np.concatenate(per_rank) if per_rank else np.zeros(0, np.int64), out=off_host[1:])

    if local_hashes.numel() >= max_total:
        payload = local_hashes[:max_total]  # the tail past this rank's own total is never read back
    else:
        payload = torch.zeros(max_total, dtype=torch.int64, device=dev)
        rank = dist.get_rank(group)
        payload[: totals[rank]] = local_hashes[: totals[rank]]
    gathered = torch.empty(world * max_total, dtype=torch.int64, device=dev)
    work = dist.all_gather_into_tensor(gathered, payload, group=group, async_op=True)
    if while_in_flight is not None:
        while_in_flight()
    work.wait()
    if all(t == max_total for t in totals):
        hashes = gathered
    else:
        gathered = gathered.view(world, max_total)
        hashes = torch.cat([gathered[r, : totals[r]] for r in range(world)])
    if hashes.numel() == 0:
        hashes = torch.zeros(1, dtype=torch.int64, device=dev)
    off = torch.from_numpy(off_host.astype(np.int64)).to(dev, non_blocking=True)
    return hashes.contiguous(), off, off_host


def sharded_pair_step(engine, torch, dist, sk_local, shard_sizes: list[int], q_range, s_range, *, backend: str = "nccl", group=None,
                      overlap: bool = True):
    """Phase 2 + 3 of the multi-GPU path on one rank: all-gather the sketches, count this rank's subject columns.

    ``sk_local`` are this rank's sketches (``HipEngine.sketch``), ``s_range`` its subject columns in global
    genome numbering.  When those columns are exactly the rank's own genomes (the uniform-length case) and fit
    one subject tile, the dictionary of the tile is built from ``sk_local`` while the payload is in flight.
    With ``backend != "nccl"`` the collectives run on host copies (plumbing check on boxes with fewer GPUs
    than ranks).  Returns (all sketches as DeviceSketches, counts tensor [nq, ns])."""
    rank = dist.get_rank(group)
    own0 = sum(shard_sizes[:rank])
    own = (own0, own0 + shard_sizes[rank])
    n_total = sum(shard_sizes)
    can_overlap = (overlap and tuple(s_range) == own and 0 < shard_sizes[rank] <= 2048 and sk_local.total > 0
                   and hasattr(engine, "pair_dict_prepare"))
    hook = (lambda: engine.pair_dict_prepare(sk_local.hashes, sk_local.total)) if can_overlap else None
    sizes = sk_local.off[1:] - sk_local.off[:-1]
    if backend == "nccl":
        hashes, off, off_host = allgather_sketches(torch, dist, sk_local.hashes, sizes, shard_sizes, group, while_in_flight=hook)
    else:
        hashes, off, off_host = allgather_sketches(
            torch, dist, sk_local.hashes[: max(1, sk_local.total)].cpu(), sizes.cpu(), shard_sizes, group, while_in_flight=hook
        )
    sk = engine.sketches_from_gathered(hashes, off, off_host)
    assert sk.n == n_total
    if s_range[0] == s_range[1] or q_range[0] == q_range[1]:  # a rank without columns (fewer genomes than ranks)
        return sk, None
    counts = engine.pair_counts(sk, tuple(q_range), tuple(s_range))
    return sk, counts
