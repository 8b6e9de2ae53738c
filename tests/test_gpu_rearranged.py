"""GPU, full size: fragment ANI on genomes that differ from their species' root by MORE than substitutions.

The benchmark's sets (``synth_arena_torch``) are point substitutions on one contig: every fragment has one locus in
every genome of its species and half of the mapping kernel's segments are "one run of hits".  ``synth_rearranged_arena_torch``
adds what real assemblies of one species differ by -- indels with geometric lengths, 3-5 inversions / translocations,
repeat families of 5-20 copies of 1-2 kb elements, 30-200 contigs per genome -- the regime of the reference's own
bacterial fixtures (/root/reference/tests/fixtures/bacterial_example/intermediates/fastANI/*.fastani: four real genomes,
83 % pairs, kept fragments well below the totals).  Here 200 such genomes of 5 Mb go through ``pa_fragani`` all against
all, and every integer and the float mean of 56 sampled ordered pairs -- every genome of two species, and some of
others, against one reference each -- must equal the oracle's (tuned form on all of them, the checking form, which is the
one pinned to the reference's 25 fastANI rows, on eight).
"""

from __future__ import annotations

import os

import numpy as np
import pytest

import oracle
from pyani_plus_amd.synth import species_and_rate

pytestmark = pytest.mark.gpu
N, LENGTH, SPECIES = 200, 5_000_000, 8
K, FRAG = 16, 3000


def _contigs_of(arena, c_start, c_len, c_genome, g: int) -> list[bytes]:
    """The contigs of genome ``g`` as ASCII, from the device arena (two bits per residue; the set holds no N)."""
    s, e = int(arena.genome_start[g]), int(arena.genome_start[g + 1])
    words = arena.packed[s // 16 : e // 16].cpu().numpy().view(np.uint32)
    codes = ((words[:, None] >> (np.arange(16, dtype=np.uint32) * 2)[None, :]) & 3).astype(np.uint8).reshape(-1)
    text = np.frombuffer(b"ACGT", dtype=np.uint8)[codes]
    sel = c_genome == g
    return [text[int(a) - s : int(a) - s + int(n)].tobytes() for a, n in zip(c_start[sel], c_len[sel])]


def test_rearranged_genomes_against_the_oracle_at_full_size():
    from oracle import pyoracle
    from pyani_plus_amd import _capi
    from pyani_plus_amd.engine import HipEngine
    from pyani_plus_amd.methods.fastani_hip import fastani_mean
    from pyani_plus_amd.synth import synth_rearranged_arena_torch

    engine = HipEngine(0)
    try:
        arena, c_start, c_len, c_genome = synth_rearranged_arena_torch(engine, N, LENGTH, n_species=SPECIES)
        contigs_per_genome = np.bincount(c_genome, minlength=N)
        assert contigs_per_genome.min() >= 20 and contigs_per_genome.max() <= 200
        total, matched, ident_sum = engine.fragani(arena, c_start, c_len, c_genome, K, FRAG)
        ani = fastani_mean(ident_sum, matched)
        # fragments: per contig floor(len / fragLen), the remainder dropped (pyani_plus/methods/fastani.py:100-116: the last column)
        want_total = np.bincount(c_genome, weights=(c_len // FRAG), minlength=N).astype(np.uint32)
        assert np.array_equal(total, want_total)
        sp = np.array([species_and_rate(g, SPECIES)[0] for g in range(N)])
        same = sp[:, None] == sp[None, :]
        # a genome keeps (nearly) all its fragments against itself -- repeat copies compete for reference bins, as in the
        # reference's own self rows (1820 of 1825, ...) --, species mates map, strangers do not reach minFraction
        assert np.all(np.diag(matched) >= 0.97 * total) and np.all(np.diag(ani) > 99.9)
        # species mates whose substitutions alone leave them above 90 % identity (the generator's rates go up to 20 % per genome: two
        # such genomes are 65 % identical and below fastANI's 80 % floor) map most of their fragments; strangers stay below minFraction
        rate = np.array([species_and_rate(g, SPECIES)[1] for g in range(N)])
        p_true = (1 - rate[:, None]) * (1 - rate[None, :]) + rate[:, None] * rate[None, :] / 3.0
        close = same & (p_true >= 0.9)
        assert int(close.sum()) > 1000 and np.all(matched[close] > 0.5 * np.minimum(total[:, None], total[None, :])[close])
        assert np.all(matched[~same] < 0.2 * total[:, None].repeat(N, 1)[~same])
        # ---- the oracle on sampled ordered pairs: two references, every genome of their species + four strangers each
        cores = max(1, min(len(os.sched_getaffinity(0)), int(_capi.load_library().pa_host_cpu_budget())))
        checked = 0
        for ref in (0, 9):
            queries = [g for g in range(N) if sp[g] == sp[ref]] + [g for g in range(N) if sp[g] != sp[ref]][:4]
            ref_contigs = _contigs_of(arena, c_start, c_len, c_genome, ref)
            q_contigs = [_contigs_of(arena, c_start, c_len, c_genome, g) for g in queries]
            pyoracle.fragani_set_fast(True)
            try:
                o_ani, o_m, o_t = oracle.fragani_many(q_contigs, ref_contigs, K, FRAG, 0.0, threads=cores)
            finally:
                pyoracle.fragani_set_fast(False)
            s_ani, s_m, s_t = oracle.fragani_many(q_contigs[:4], ref_contigs, K, FRAG, 0.0, threads=cores)  # the checking form
            for i, q in enumerate(queries):
                assert (int(total[q]), int(matched[q, ref])) == (int(o_t[i]), int(o_m[i])), (q, ref)
                assert o_m[i] == 0 or float(ani[q, ref]) == float(o_ani[i]), (q, ref, ani[q, ref], o_ani[i])
                if i < 4:
                    assert (int(s_t[i]), int(s_m[i])) == (int(o_t[i]), int(o_m[i])) and (s_m[i] == 0 or float(s_ani[i]) == float(o_ani[i]))
                checked += 1
        assert checked >= 40
    finally:
        engine.close()
