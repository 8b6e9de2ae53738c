"""Seeded synthetic genome sets for benchmarks and size-independent parity checks.

Shape follows SURVEY.md section 8(d) config 2: ``n_species`` random root genomes
and, per genome, point substitutions at a rate cycled from ``RATES`` so that
sketches overlap (uniformly random genomes share no 31-mers and every pair
would be NULL).  Genomes are produced directly in the 2-bit arena layout of
``include/pyani_hip.h`` -- a uniformly random base string *is* a uniformly
random word string -- so 1 000 x 5 Mb never exists as 5 GB of ASCII.

The generators are plumbing (numpy / torch RNG), not part of the measured path.
"""

from __future__ import annotations

import numpy as np

from .engine import DeviceArena, HostArena

RATES = (0.001, 0.002, 0.005, 0.01, 0.02, 0.05, 0.1, 0.2)
SEED = 20260802


def _padded(length: int) -> int:
    """Arena footprint of one genome: at least one invalid position after the last base."""
    return (length // 64 + 1) * 64


def species_and_rate(g: int, n_species: int) -> tuple[int, float]:
    return g % n_species, RATES[(g // n_species) % len(RATES)]


def synth_arena_numpy(n_genomes: int, lengths, n_species: int = 4, seed: int = SEED) -> HostArena:
    """Small host-side generator (CPU tests, multi-process gloo tests, smoke)."""
    lengths = [int(lengths)] * n_genomes if np.isscalar(lengths) else [int(x) for x in lengths]
    rng = np.random.Generator(np.random.Philox(key=seed))
    max_len = max(lengths) if lengths else 0
    roots = rng.integers(0, 4, size=(n_species, max_len), dtype=np.uint8)
    starts = np.zeros(n_genomes + 1, dtype=np.uint64)
    np.cumsum([_padded(x) for x in lengths], out=starts[1:])
    total = int(starts[-1])
    codes = np.zeros(total, dtype=np.uint8)
    invalid = np.ones(total, dtype=np.uint8)
    for g, length in enumerate(lengths):
        sp, rate = species_and_rate(g, n_species)
        seq = roots[sp, :length].copy()
        hit = rng.random(length) < rate
        seq[hit] = (seq[hit] + rng.integers(1, 4, size=int(hit.sum()), dtype=np.uint8)) & 3
        s = int(starts[g])
        codes[s : s + length] = seq
        invalid[s : s + length] = 0
    shifts = (np.arange(16, dtype=np.uint32) * 2)[None, :]
    packed = (codes.reshape(-1, 16).astype(np.uint32) << shifts).sum(axis=1, dtype=np.uint64).astype(np.uint32)
    mshift = np.arange(32, dtype=np.uint64)[None, :]
    mask = (invalid.reshape(-1, 32).astype(np.uint64) << mshift).sum(axis=1, dtype=np.uint64).astype(np.uint32)
    return HostArena(packed, mask, starts, residues=lengths, records=[1] * n_genomes, invalid=[0] * n_genomes)


def arena_to_ascii(arena: HostArena, g: int) -> bytes:
    """Residues of genome ``g`` as upper-case ASCII (invalid positions -> 'N'), for the oracle."""
    s, e = int(arena.genome_start[g]), int(arena.genome_start[g + 1])
    words = arena.packed[s // 16 : e // 16]
    codes = ((words[:, None] >> (np.arange(16, dtype=np.uint32) * 2)[None, :]) & 3).astype(np.uint8).reshape(-1)
    mwords = arena.mask[s // 32 : e // 32]
    inv = ((mwords[:, None] >> np.arange(32, dtype=np.uint32)[None, :]) & 1).astype(bool).reshape(-1)
    out = np.frombuffer(b"ACGT", dtype=np.uint8)[codes]
    out[inv] = ord("N")
    if getattr(arena, "ambig_pos", None) is not None and len(arena.ambig_pos):  # the letters the mask bit does not tell
        sel = (arena.ambig_pos >= s) & (arena.ambig_pos < e)
        out[(arena.ambig_pos[sel] - np.uint64(s)).astype(np.int64)] = arena.ambig_byte[sel]
    length = e - s
    if arena.residues:  # residues + one separator between consecutive records
        length = arena.residues[g] + (max(arena.records[g] - 1, 0) if arena.records else 0)
    return out[:length].tobytes()


def mixed_lengths(n_genomes: int, lo: int = 100_000, hi: int = 10_000_000, seed: int = SEED) -> list[int]:
    """Log-uniform genome lengths (BASELINE configs[4]: 100 kb - 10 Mb)."""
    rng = np.random.Generator(np.random.Philox(key=seed + 1))
    return [int(x) for x in np.exp(rng.uniform(np.log(lo), np.log(hi), size=n_genomes))]


def synth_arena_torch(engine, n_genomes: int, length, n_species: int = 40, seed: int = SEED, *, genome_offset: int = 0,
                      genome_ids=None) -> DeviceArena:
    """Generate the arena on the GPU (torch RNG).  ``length`` is one length or a list with one
    entry per genome of THIS shard; ``genome_offset`` numbers the shard's genomes globally so
    that every rank of a multi-GPU run draws its own slice of one set; ``genome_ids`` (optional) names the global
    number of every genome of the arena instead (the same set in another order)."""
    t = engine.torch
    dev = engine.device
    lengths = [int(length)] * n_genomes if np.isscalar(length) else [int(x) for x in length]
    assert len(lengths) == n_genomes
    max_pad = _padded(max(lengths)) if lengths else 64
    gen = t.Generator(device=dev)
    shifts = (t.arange(16, device=dev, dtype=t.int64) * 2)[None, :]
    mshifts = t.arange(32, device=dev, dtype=t.int64)[None, :]
    # roots depend only on (seed, species) so every rank builds identical roots
    roots = {}
    pads = [_padded(x) for x in lengths]
    starts = np.zeros(n_genomes + 1, dtype=np.uint64)
    np.cumsum(pads, out=starts[1:])
    total = int(starts[-1])
    packed = t.empty(max(total // 16, 1), dtype=t.int32, device=dev)
    mask = t.empty(max(total // 32, 1), dtype=t.int32, device=dev)
    position = t.arange(max_pad, device=dev)
    for i in range(n_genomes):
        g = genome_offset + i if genome_ids is None else int(genome_ids[i])
        sp, rate = species_and_rate(g, n_species)
        if sp not in roots:
            gen.manual_seed(seed * 1000003 + sp)
            roots[sp] = t.randint(0, 4, (max_pad,), generator=gen, device=dev, dtype=t.int64)
        padded = pads[i]
        valid = position[:padded] < lengths[i]
        gen.manual_seed(seed * 7919 + 104729 * (g + 1))
        hit = t.rand(padded, generator=gen, device=dev) < rate
        delta = t.randint(1, 4, (padded,), generator=gen, device=dev, dtype=t.int64) * hit
        codes = ((roots[sp][:padded] + delta) & 3) * valid
        words = (codes.view(-1, 16) << shifts).sum(1)
        words = t.where(words >= 2**31, words - 2**32, words).to(t.int32)
        s0 = int(starts[i])
        packed[s0 // 16 : s0 // 16 + padded // 16] = words
        inv = ((~valid).to(t.int64).view(-1, 32) << mshifts).sum(1)
        mask[s0 // 32 : s0 // 32 + padded // 32] = t.where(inv >= 2**31, inv - 2**32, inv).to(t.int32)
    t.cuda.synchronize(dev)
    return DeviceArena(packed, mask, starts)


def device_arena_to_host(arena: DeviceArena, genomes: list[int], length) -> HostArena:
    """Copy a few genomes of a device arena back to the host (oracle sample)."""
    starts = np.zeros(len(genomes) + 1, dtype=np.uint64)
    packed, mask = [], []
    for i, g in enumerate(genomes):
        s, e = int(arena.genome_start[g]), int(arena.genome_start[g + 1])
        packed.append(arena.packed[s // 16 : e // 16].cpu().numpy().view(np.uint32))
        mask.append(arena.mask[s // 32 : e // 32].cpu().numpy().view(np.uint32))
        starts[i + 1] = starts[i] + np.uint64(e - s)
    residues = [int(length)] * len(genomes) if np.isscalar(length) else [int(x) for x in length]
    return HostArena(np.concatenate(packed), np.concatenate(mask), starts, residues=residues,
                     records=[1] * len(genomes), invalid=[0] * len(genomes))
