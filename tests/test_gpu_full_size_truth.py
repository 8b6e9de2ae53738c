"""GPU, full size: both methods on the 1 000-genome synthetic set of BASELINE configs[1] / configs[3] against the
GROUND TRUTH the generator knows -- not against another implementation.

``synth.py`` makes genome g from the root of species ``g % 40`` by point substitutions at rate ``RATES[(g // 40) % 8]``
(every hit changes the base), independently per genome.  Two genomes of one species with rates a and b therefore
agree at a position with probability ``P = (1 - a)(1 - b) + a b / 3``: that is their true identity, and the
probability that a k-mer survives in both is ``P^k`` -- exactly the model behind both estimators.

* sourmash path: containment ANI ``(I / |S|)^(1/31)`` must track P within the sampling error of a sketch of |S|
  hashes (binomial: sigma_ANI = P * sqrt((1 - C) / (C |S|)) / k with C = P^k), stated band = 5 sigma + 2e-4;
  different species share no 31-mer: NULL, every one of the 975 000 ordered pairs.
* fragment ANI: ``total_frags`` exact (floor(5 000 000 / 3000) = 1666), a genome maps every fragment onto itself at
  exactly 100 %, same-species ANI tracks 100 P within a stated band per identity class down to 85 %, different
  species keep a handful of chance fragments near the 80 % floor and are never reported (minFraction).
"""

from __future__ import annotations

import numpy as np
import pytest

from pyani_plus_amd.synth import RATES, species_and_rate

pytestmark = pytest.mark.gpu
N, LENGTH, SPECIES = 1000, 5_000_000, 40
# fragment ANI by true identity class: (P from, P below, largest |ANI - P| allowed, smallest mapped fraction allowed).
# Measured on the MI355X (profiles/r03_truth_full_size.txt): 0.0022 / 0.9976, 0.0054 / 0.9934, 0.0119 / 0.9748, 0.0213 / 0.9424.
BANDS = ((0.99, 1.01, 0.003, 0.995), (0.95, 0.99, 0.007, 0.99), (0.90, 0.95, 0.015, 0.965), (0.85, 0.90, 0.026, 0.93))


@pytest.fixture(scope="module")
def engine():
    from pyani_plus_amd.engine import HipEngine

    eng = HipEngine(0)
    yield eng
    eng.close()


@pytest.fixture(scope="module")
def arena(engine):
    from pyani_plus_amd.synth import synth_arena_torch

    return synth_arena_torch(engine, N, LENGTH, n_species=SPECIES)


def _truth():
    sp = np.array([species_and_rate(g, SPECIES)[0] for g in range(N)])
    rate = np.array([species_and_rate(g, SPECIES)[1] for g in range(N)])
    same = sp[:, None] == sp[None, :]
    p = (1 - rate[:, None]) * (1 - rate[None, :]) + rate[:, None] * rate[None, :] / 3.0
    np.fill_diagonal(p, 1.0)
    return same, p, rate


def test_containment_ani_tracks_the_generators_identity(engine, arena):
    from pyani_plus_amd.engine import ani_host

    k, scaled = 31, 1000
    sk = engine.sketch(arena, k, scaled)
    counts = engine.pair_counts(sk).cpu().numpy().view(np.uint32)
    sizes = sk.sizes()
    ident, cov, null = ani_host(counts, sizes, sizes, k, symmetric=True)
    same, p, _rate = _truth()
    # different species: no shared 31-mer among 5 000 sampled ones, all NULL (the reference's NULL, not 0.0)
    assert np.all(null[~same]) and int((~same).sum()) == 975_000
    assert np.all(np.diag(ident) == 1.0) and np.all(np.diag(cov) == 1.0)
    assert np.array_equal(counts, counts.T)
    c_true = p**k
    expected_shared = c_true * sizes[None, :].astype(float)
    sure = same & (expected_shared >= 30)
    assert not np.any(null[sure])  # 30 expected shared hashes never come out as zero
    s_mean = float(sizes.mean())
    sigma = p * np.sqrt((1 - c_true) / (c_true * s_mean)) / k
    band = 5 * sigma + 2e-4
    dev_cov = np.abs(cov - p)
    assert np.all(dev_cov[sure] <= band[sure]), float((dev_cov[sure] / band[sure]).max())
    # identity = max containment: the same estimate up to the two sketch sizes (within 3 %)
    assert np.all(np.abs(ident[sure] - p[sure]) <= band[sure] + 0.03 / k)
    # what is NULL among same-species pairs is what the model says cannot be seen: fewer than one hash expected
    assert np.all(expected_shared[same & null] < 12)
    print(f"containment ANI vs truth: {int(sure.sum())} same-species pairs, max |dev| {dev_cov[sure].max():.2e}, "
          f"max dev/band {float((dev_cov[sure] / band[sure]).max()):.2f}")


def test_fragment_ani_tracks_the_generators_identity(engine, arena):
    k, frag = 16, 3000
    starts = np.ascontiguousarray(arena.genome_start[:-1])
    lens = np.full(N, LENGTH, dtype=np.uint32)
    total, matched, ident_sum = engine.fragani(arena, starts, lens, np.arange(N, dtype=np.uint32), k, frag)
    same, p, rate = _truth()
    assert np.all(total == LENGTH // frag)  # 1666, exactly
    # self: every fragment maps at 100 % -- up to the one or two per genome that lose their reference bin of
    # fragLen - 20 positions to a neighbour (fastANI's own self rows read 1820/1825, 1346/1347, ...)
    assert np.all(np.diag(matched) >= total - 2) and int((np.diag(matched) == total).sum()) >= 0.8 * N
    self_ani = np.diag(ident_sum) / np.diag(matched)
    print(f"self pairs: {int((np.diag(matched) == total).sum())} of {N} genomes keep all {LENGTH // frag} fragments, the others {int(np.diag(matched).min())}+; "
          f"mean identity {self_ani.min():.6f}..{self_ani.max():.6f} %")
    assert self_ani.min() >= 100.0 - 1e-9 and self_ani.max() <= 100.0 + 1e-9  # exactly 100 % since the exact slide of round 4 (99.993 before)
    # different species: unrelated 5 Mb genomes give a handful of chance mappings near the 80 % floor (as fastANI's
    # own statistics allow: its p-value bounds false fragments per reference, not across a million pairs) -- never
    # enough to pass minFraction, so every such pair is NULL in the database
    from pyani_plus_amd.methods.fastani_hip import is_reported

    stray = matched[~same]
    assert stray.max() <= 0.01 * (LENGTH // frag), int(stray.max())
    assert not is_reported(int(stray.max()), LENGTH // frag, frag, 0.2, LENGTH, LENGTH)
    with np.errstate(invalid="ignore", divide="ignore"):
        stray_ani = (ident_sum[~same] / stray)[stray > 0]
    print(f"different species: {int((stray > 0).sum())} of {stray.size} ordered pairs keep 1..{int(stray.max())} chance fragments, "
          f"their identities {stray_ani.min():.2f}..{stray_ani.max():.2f} %")
    assert stray_ani.max() < 86.0
    with np.errstate(invalid="ignore", divide="ignore"):
        ani = ident_sum / matched
    # same species: the mapped fraction and the mean identity of the mapped fragments against the truth, by identity
    # class.  Bands = measured maxima on this set (profiles/r03_truth_full_size.txt) plus a margin; the estimate is a
    # winnowed-MinHash Jaccard of ~240 minimizers per 3 kb fragment, and below 90 % only the fragments that still
    # pass the identity floor are averaged, which biases the mean upwards.
    off_diag = same & ~np.eye(N, dtype=bool)
    lines = []
    for lo, hi, max_dev, min_frac in BANDS:
        sel = off_diag & (p >= lo) & (p < hi)
        frac = matched[sel] / float(LENGTH // frag)
        dev = np.abs(ani[sel] / 100.0 - p[sel])
        lines.append(f"P in [{lo}, {hi}): {int(sel.sum())} pairs, mapped fraction {frac.min():.4f}..{frac.max():.4f}, |ANI - P| max {dev.max():.5f} mean {dev.mean():.5f}")
        print(lines[-1])
    for (lo, hi, max_dev, min_frac), line in zip(BANDS, lines):
        sel = off_diag & (p >= lo) & (p < hi)
        assert (matched[sel] / float(LENGTH // frag)).min() >= min_frac, line
        assert np.abs(ani[sel] / 100.0 - p[sel]).max() <= max_dev, line
    # pairs the generator puts far below fastANI's 80 % floor are not reported as relatives
    far = off_diag & (p < 0.70)
    reported = matched[far] * frag >= 0.2 * LENGTH
    print(f"P < 0.70: {int(far.sum())} pairs, {int(reported.sum())} pass minFraction")
    assert not np.any(reported)
    del rate
