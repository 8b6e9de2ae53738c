"""CPU: the generator of the rearranged genome sets (pyani_plus_amd/synth.py) -- the workload of
``tests/test_gpu_rearranged.py`` and of ``also.fragment_ani_rearranged`` in bench.py -- with torch on the CPU: the arena
layout the kernels rely on (contigs separated by one invalid position, genomes padded to 64), the contig tables, determinism,
and that the oracle maps genomes of one species onto each other and not onto another species'."""

from __future__ import annotations

import numpy as np
import pytest

import oracle
from pyani_plus_amd import synth

torch = pytest.importorskip("torch")


class _CpuEngine:
    torch = torch
    device = "cpu"


def _decode(arena, g):
    s, e = int(arena.genome_start[g]), int(arena.genome_start[g + 1])
    words = arena.packed[s // 16 : e // 16].numpy().view(np.uint32)
    codes = ((words[:, None] >> (np.arange(16, dtype=np.uint32) * 2)[None, :]) & 3).astype(np.uint8).reshape(-1)
    mw = arena.mask[s // 32 : e // 32].numpy().view(np.uint32)
    invalid = ((mw[:, None] >> np.arange(32, dtype=np.uint32)[None, :]) & 1).astype(bool).reshape(-1)
    return codes, invalid, s


def test_rearranged_arena_layout_and_contig_tables():
    n, length = 5, 150_000
    arena, c_start, c_len, c_genome = synth.synth_rearranged_arena_torch(_CpuEngine, n, length, n_species=2, contigs=(3, 9))
    again = synth.synth_rearranged_arena_torch(_CpuEngine, n, length, n_species=2, contigs=(3, 9))
    assert torch.equal(arena.packed, again[0].packed) and torch.equal(arena.mask, again[0].mask) and np.array_equal(c_start, again[1])
    assert np.all(np.diff(c_genome.astype(np.int64)) >= 0) and set(c_genome.tolist()) == set(range(n))  # contigs listed genome by genome
    assert np.all(arena.genome_start % 64 == 0)
    for g in range(n):
        codes, invalid, s = _decode(arena, g)
        sel = c_genome == g
        assert 3 <= int(sel.sum()) <= 9
        for a, m in zip(c_start[sel], c_len[sel]):
            a = int(a) - s
            assert m > 0 and not invalid[a : a + m].any() and invalid[a + m]  # valid inside, one invalid position right after
        assert int((~invalid).sum()) == int(c_len[sel].sum())  # nothing valid outside the contigs
        assert invalid[-1]  # the genome ends with an invalid position (no window runs into the next genome)
        # the root's residues plus the species' repeat copies, give or take the indels
        assert 0.95 * length < int(c_len[sel].sum()) < length + synth.REARRANGED_FAMILIES * synth.REARRANGED_COPIES[1] * synth.REARRANGED_ELEMENT[1]
    # pieces: every kind occurs somewhere in a genome with a high substitution rate (many indels)
    pieces, contig_lens, _root_len, _sp, rate = synth.rearranged_genome_pieces(2 * 7, 400_000, 2, contigs=(3, 9))  # genome 14: species 0, rate 0.2
    kinds = set(pieces[0].tolist())
    assert rate == 0.2 and {0, 1, 2, 3} <= kinds and len(contig_lens) == int((pieces[0] == 3).sum()) + 1
    assert sum(contig_lens) + len(contig_lens) - 1 == int(pieces[2].sum())


def test_rearranged_genomes_map_within_their_species():
    n, length = 4, 150_000
    arena, c_start, c_len, c_genome = synth.synth_rearranged_arena_torch(_CpuEngine, n, length, n_species=2, contigs=(3, 6))
    contigs = []
    for g in range(n):
        codes, _invalid, s = _decode(arena, g)
        text = np.frombuffer(b"ACGT", dtype=np.uint8)[codes]
        sel = c_genome == g
        contigs.append([text[int(a) - s : int(a) - s + int(m)].tobytes() for a, m in zip(c_start[sel], c_len[sel])])
    ani_self, kept_self, total_self = oracle.fragani_pair(contigs[0], contigs[0], 16, 3000, 0.0)
    assert ani_self == 100.0 and kept_self >= 0.9 * total_self > 0
    ani_mate, kept_mate, total = oracle.fragani_pair(contigs[0], contigs[2], 16, 3000, 0.0)  # genomes 0 and 2: species 0, rates 0.001 and 0.002
    assert 98.0 < ani_mate < 100.0 and kept_mate >= 0.8 * total
    _ani, kept_other, total = oracle.fragani_pair(contigs[0], contigs[1], 16, 3000, 0.0)  # another species
    assert kept_other <= 0.05 * total
