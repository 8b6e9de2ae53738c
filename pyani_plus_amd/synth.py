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


# ---- a second generator mode: genomes that differ from their species' root by more than substitutions -----------------
# (fragment ANI, BASELINE configs[3]).  The substitution-only sets above are the friendliest input a fragment mapper can
# get: every fragment of a genome has ONE locus in every genome of its species, on one contig, at the same offset.  Real
# assemblies of one species differ by indels, by a handful of inversions and translocations, carry repeat families
# (IS elements, rRNA operons: seed hits at several loci) and come as tens to hundreds of contigs -- the regime of the
# reference's own bacterial fixtures (/root/reference/tests/fixtures/bacterial_example/intermediates/fastANI/*.fastani).
# A genome here is a list of PIECES of its species' root (which already holds the species' repeat copies): forward or
# reverse-complemented stretches, random insertions, contig separators -- drawn on the host from (seed, genome), expanded
# on the device by one gather.  Plumbing, not part of any measured path.
REARRANGED_CONTIGS = (30, 200)      # contigs per genome
REARRANGED_EVENTS = (3, 5)          # inversions / translocations per genome
REARRANGED_FAMILIES = 3             # repeat families per species,
REARRANGED_COPIES = (5, 20)         # copies of each in the root,
REARRANGED_ELEMENT = (1000, 2000)   # residues per copy
_P_ROOT, _P_REVCOMP, _P_RANDOM, _P_SEPARATOR = 0, 1, 2, 3


def _species_root_pieces(rng, length: int) -> list[tuple[int, int, int]]:
    """The species' root as pieces over a plain random string of ``length`` residues: copies of the species' repeat
    elements (pieces of the string's tail, where the elements live) spliced in at random places.  (kind, start, n)."""
    families = [(int(rng.integers(*REARRANGED_ELEMENT)), int(rng.integers(REARRANGED_COPIES[0], REARRANGED_COPIES[1] + 1))) for _ in range(REARRANGED_FAMILIES)]
    elements, tail = [], length
    for n_el, _copies in families:  # the elements themselves sit past the root's own residues
        elements.append((tail, n_el))
        tail += n_el
    cuts = sorted((int(rng.integers(0, length)), f) for f, (_n, copies) in enumerate(families) for _ in range(copies))
    pieces, at = [], 0
    for pos, f in cuts:
        if pos > at:
            pieces.append((_P_ROOT, at, pos - at))
        pieces.append((_P_ROOT if rng.random() < 0.5 else _P_REVCOMP, elements[f][0], elements[f][1]))
        at = pos
    if at < length:
        pieces.append((_P_ROOT, at, length - at))
    return pieces, tail


def _slice_pieces(pieces, a: int, b: int):
    """Pieces covering positions [a, b) of the sequence the pieces spell."""
    out, at = [], 0
    for kind, start, n in pieces:
        lo, hi = max(a, at), min(b, at + n)
        if lo < hi:
            off, m = lo - at, hi - lo
            if kind == _P_REVCOMP:  # a reverse-complemented stretch: the slice counts from its far end
                out.append((kind, start + n - off - m, m))
            elif kind == _P_ROOT:
                out.append((kind, start + off, m))
            else:
                out.append((kind, start, m))
        at += n
        if at >= b:
            break
    return out


def _revcomp_pieces(pieces):
    flip = {_P_ROOT: _P_REVCOMP, _P_REVCOMP: _P_ROOT}
    return [(flip.get(kind, kind), start, n) for kind, start, n in reversed(pieces)]


def rearranged_genome_pieces(g: int, length: int, n_species: int, seed: int = SEED, contigs: tuple[int, int] = REARRANGED_CONTIGS):
    """Pieces of genome ``g`` -- arrays (kind, start in the species' substituted root, residues) -- and its contig lengths."""
    sp, rate = species_and_rate(g, n_species)
    root_pieces, root_len = _species_root_pieces(np.random.Generator(np.random.Philox(key=seed * 31 + 7 * sp + 3)), length)
    total = sum(n for _k, _s, n in root_pieces)
    rng = np.random.Generator(np.random.Philox(key=seed * 131 + 977 * (g + 1)))
    pieces = root_pieces
    # inversions and translocations of 20 - 300 kb (as far as the genome allows)
    for _ in range(int(rng.integers(REARRANGED_EVENTS[0], REARRANGED_EVENTS[1] + 1))):
        seg = int(min(total // 4, rng.integers(20_000, 300_000))) if total >= 8 else 0
        if seg < 2:
            break
        a = int(rng.integers(0, total - seg))
        left, mid, right = _slice_pieces(pieces, 0, a), _slice_pieces(pieces, a, a + seg), _slice_pieces(pieces, a + seg, total)
        if rng.random() < 0.5:
            pieces = left + _revcomp_pieces(mid) + right  # inversion in place
        else:  # translocation: cut out, put back somewhere else (in either orientation)
            rest = left + right
            to = int(rng.integers(0, total - seg + 1))
            moved = mid if rng.random() < 0.5 else _revcomp_pieces(mid)
            pieces = _slice_pieces(rest, 0, to) + moved + _slice_pieces(rest, to, total - seg)
    # indels with geometric lengths (one per ~8 substitutions, at most one per 200 residues) and the contig breaks (cut points
    # anywhere: some contigs come out shorter than a fragment, as in real drafts), in one vectorised pass over the pieces:
    # the sequence is cut at every piece boundary, indel position, deletion end and contig break; what a deletion covers goes,
    # random insertions and one-position separators come in where they belong
    kinds = np.array([k for k, _s, _n in pieces], dtype=np.int64)
    p_start = np.array([st for _k, st, _n in pieces], dtype=np.int64)
    p_len = np.array([n for _k, _s, n in pieces], dtype=np.int64)
    bounds = np.concatenate(([0], np.cumsum(p_len)))
    n_indel = int(min(total / 200, rate * total / 8))
    at = np.sort(rng.integers(0, total, size=n_indel))
    lens = np.minimum(rng.geometric(0.4, size=n_indel), 50).astype(np.int64)
    insert = rng.random(n_indel) < 0.5
    n_contigs = int(rng.integers(contigs[0], contigs[1] + 1))
    n_contigs = max(1, min(n_contigs, total // 64))
    breaks = np.unique(rng.integers(1, max(total, 2), size=n_contigs - 1)) if n_contigs > 1 else np.zeros(0, dtype=np.int64)
    d_start = at[~insert]
    d_end = np.minimum(d_start + lens[~insert], total)
    if len(d_start):  # merge deletions that run into each other
        d_end = np.maximum.accumulate(d_end)
    cuts = np.unique(np.concatenate((bounds, at[insert], d_start, d_end, breaks)))
    cuts = cuts[(cuts >= 0) & (cuts <= total)]
    i_start, i_end = cuts[:-1], cuts[1:]
    owner = np.searchsorted(bounds, i_start, side="right") - 1
    if len(d_start):
        j = np.searchsorted(d_start, i_start, side="right") - 1
        deleted = (j >= 0) & (i_start < d_end[np.maximum(j, 0)])
    else:
        deleted = np.zeros(len(i_start), dtype=bool)
    keep = ~deleted
    off = i_start - bounds[owner]
    n = i_end - i_start
    k_kind = kinds[owner]
    k_start = np.where(k_kind == _P_REVCOMP, p_start[owner] + p_len[owner] - off - n, p_start[owner] + off)
    # everything that ends up in the genome, in order: (position, what comes first at a position: separator, insertion, then the residues)
    ev_pos = np.concatenate((breaks, at[insert], i_start[keep]))
    ev_rank = np.concatenate((np.zeros(len(breaks), np.int64), np.ones(int(insert.sum()), np.int64), np.full(int(keep.sum()), 2, np.int64)))
    ev_kind = np.concatenate((np.full(len(breaks), _P_SEPARATOR, np.int64), np.full(int(insert.sum()), _P_RANDOM, np.int64), k_kind[keep]))
    ev_start = np.concatenate((np.zeros(len(breaks) + int(insert.sum()), np.int64), k_start[keep]))
    ev_len = np.concatenate((np.ones(len(breaks), np.int64), lens[insert], n[keep]))
    order = np.lexsort((ev_rank, ev_pos))
    ev_kind, ev_start, ev_len = ev_kind[order], ev_start[order], ev_len[order]
    # a separator at the very start, two in a row or one at the very end would make an empty contig: dropped
    is_sep = ev_kind == _P_SEPARATOR
    prev_sep = np.concatenate(([True], is_sep[:-1]))
    drop = is_sep & prev_sep
    if len(is_sep) and is_sep[-1]:
        drop[-1] = True
    ev_kind, ev_start, ev_len = ev_kind[~drop], ev_start[~drop], ev_len[~drop]
    sep_at = np.flatnonzero(ev_kind == _P_SEPARATOR)
    csum = np.concatenate(([0], np.cumsum(ev_len)))
    edges = np.concatenate(([0], csum[sep_at + 1], [csum[-1] + 1]))  # a contig runs from after a separator up to the next one
    contig_lens = [int(x) for x in (edges[1:] - edges[:-1] - 1)]
    return (ev_kind, ev_start, ev_len), contig_lens, root_len, sp, rate


def synth_rearranged_arena_torch(engine, n_genomes: int, length: int, n_species: int = 40, seed: int = SEED, *, genome_ids=None, device=None,
                                 contigs: tuple[int, int] = REARRANGED_CONTIGS):
    """The rearranged set on the device: (DeviceArena, contig_start, contig_len, contig_genome).  Contigs of a genome are
    separated by one invalid position, as the FASTA packers leave them (include/pyani_hip.h); ``device``: for host-side tests
    (torch CPU tensors) -- by default the engine's."""
    t = engine.torch
    dev = engine.device if device is None else device
    plans = [rearranged_genome_pieces(i if genome_ids is None else int(genome_ids[i]), int(length), n_species, seed, contigs) for i in range(n_genomes)]
    lengths = [int(pl[0][2].sum()) for pl in plans]  # residues + separators
    pads = [_padded(x) for x in lengths]
    starts = np.zeros(n_genomes + 1, dtype=np.uint64)
    np.cumsum(pads, out=starts[1:])
    total = int(starts[-1])
    packed = t.empty(max(total // 16, 1), dtype=t.int32, device=dev)
    mask = t.empty(max(total // 32, 1), dtype=t.int32, device=dev)
    shifts = (t.arange(16, device=dev, dtype=t.int64) * 2)[None, :]
    mshifts = t.arange(32, device=dev, dtype=t.int64)[None, :]
    gen = t.Generator(device=dev)
    roots: dict[int, object] = {}
    c_start, c_len, c_genome = [], [], []
    for i, (pieces, contig_lens, root_len, sp, rate) in enumerate(plans):
        g = i if genome_ids is None else int(genome_ids[i])
        if sp not in roots:
            gen.manual_seed(seed * 1000003 + sp)
            roots[sp] = t.randint(0, 4, (root_len,), generator=gen, device=dev, dtype=t.int64)
        gen.manual_seed(seed * 7919 + 104729 * (g + 1))
        hit = t.rand(root_len, generator=gen, device=dev) < rate
        delta = t.randint(1, 4, (root_len,), generator=gen, device=dev, dtype=t.int64) * hit
        subbed = (roots[sp] + delta) & 3
        kinds = t.from_numpy(pieces[0]).to(dev)
        p_start = t.from_numpy(pieces[1]).to(dev)
        p_len = t.from_numpy(pieces[2]).to(dev)
        out_off = t.cumsum(p_len, 0) - p_len
        pid = t.repeat_interleave(t.arange(len(pieces[0]), device=dev), p_len)
        j = t.arange(lengths[i], device=dev) - out_off[pid]
        kind = kinds[pid]
        src = t.where(kind == _P_REVCOMP, p_start[pid] + p_len[pid] - 1 - j, p_start[pid] + j)
        src = t.where(kind >= _P_RANDOM, t.zeros_like(src), src)
        codes = subbed[src]
        codes = t.where(kind == _P_REVCOMP, 3 - codes, codes)
        codes = t.where(kind == _P_RANDOM, t.randint(0, 4, (lengths[i],), generator=gen, device=dev, dtype=t.int64), codes)
        valid = kind != _P_SEPARATOR
        padded = pads[i]
        full = t.zeros(padded, dtype=t.int64, device=dev)
        full[: lengths[i]] = codes * valid
        inv = t.ones(padded, dtype=t.int64, device=dev)
        inv[: lengths[i]] = (~valid).to(t.int64)
        words = (full.view(-1, 16) << shifts).sum(1)
        words = t.where(words >= 2**31, words - 2**32, words).to(t.int32)
        s0 = int(starts[i])
        packed[s0 // 16 : s0 // 16 + padded // 16] = words
        iw = (inv.view(-1, 32) << mshifts).sum(1)
        mask[s0 // 32 : s0 // 32 + padded // 32] = t.where(iw >= 2**31, iw - 2**32, iw).to(t.int32)
        at = s0
        for n in contig_lens:
            c_start.append(at)
            c_len.append(n)
            c_genome.append(i)
            at += n + 1
    if dev != "cpu" and str(dev) != "cpu":
        t.cuda.synchronize(dev)
    return (DeviceArena(packed, mask, starts), np.array(c_start, dtype=np.uint64), np.array(c_len, dtype=np.uint32),
            np.array(c_genome, dtype=np.uint32))


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
