"""GPU parity: HIP path (through the C ABI) vs the reference's own fixtures and the oracle."""

from __future__ import annotations

import numpy as np
import pytest

import oracle
from tests.helpers import FIXTURE_SETS, GOLDEN, load_manysearch, read_fasta_bytes, sig_mins

pytestmark = pytest.mark.gpu
K = 31


@pytest.fixture(scope="module")
def engine():
    from pyani_plus_amd.engine import HipEngine

    eng = HipEngine(0)
    yield eng
    eng.close()


@pytest.mark.parametrize("name", list(FIXTURE_SETS))
def test_sketch_equals_sig_fixtures(engine, name):
    from pyani_plus_amd.engine import pack_genomes

    scaled, genomes = FIXTURE_SETS[name]
    md5s = sorted(genomes)
    texts = [read_fasta_bytes(GOLDEN / name / genomes[m]) for m in md5s]
    arena = pack_genomes(texts)
    sk = engine.sketch(engine.upload(arena), K, scaled)
    got = sk.to_host()
    for m, mins in zip(md5s, got):
        want = sig_mins(GOLDEN / name / "sourmash" / f"{m}.sig")
        assert mins.dtype == np.uint64
        assert np.array_equal(mins, want), f"{name}/{m}: {len(mins)} vs {len(want)} hashes"


@pytest.mark.parametrize("algo", [1, 2, 3])
@pytest.mark.parametrize("name", list(FIXTURE_SETS))
def test_pairs_equal_manysearch(engine, name, algo):
    from pyani_plus_amd.engine import ani_host

    scaled, genomes = FIXTURE_SETS[name]
    md5s = sorted(genomes)
    sketches = [sig_mins(GOLDEN / name / "sourmash" / f"{m}.sig") for m in md5s]
    sk = engine.sketches_from_host(sketches)
    counts_t = engine.pair_counts(sk, algo=algo)
    counts = counts_t.cpu().numpy().view(np.uint32)
    assert np.array_equal(counts, oracle.pair_counts(sketches))
    sizes = [len(s) for s in sketches]
    ident, cov, null = ani_host(counts, sizes, sizes, K)
    d_ident, d_cov = engine.ani(counts_t, sk, K)
    d_ident, d_cov = d_ident.cpu().numpy(), d_cov.cpu().numpy()
    rows = load_manysearch(GOLDEN / name / "sourmash" / "manysearch.csv")
    seen = set()
    for row in rows:
        q, s = md5s.index(row["query_name"]), md5s.index(row["match_name"])
        seen.add((q, s))
        assert int(row["intersect_hashes"]) == counts[q, s]
        # strict (host libm) transform: bit-identical to the reference CSV
        assert float(row["query_containment_ani"]) == cov[q, s]
        assert float(row["max_containment_ani"]) == ident[q, s]
        # device transform: within 1 ulp (2.3e-16 relative)
        assert abs(d_cov[q, s] - cov[q, s]) <= 2.3e-16 * cov[q, s]
        assert abs(d_ident[q, s] - ident[q, s]) <= 2.3e-16 * ident[q, s]
        if q == s:
            assert d_ident[q, s] == 1.0 and d_cov[q, s] == 1.0
    for q in range(len(md5s)):
        for s in range(len(md5s)):
            assert ((q, s) in seen) == (not null[q, s])
            assert null[q, s] == bool(np.isnan(d_ident[q, s])) == bool(np.isnan(d_cov[q, s]))
