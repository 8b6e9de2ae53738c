"""Time the fastANI-style fragment-ANI path (BASELINE configs[3]) on synthetic 5 Mb genomes.

    python tools/bench_fragani.py [n_genomes] [cpu_sample_pairs]
Prints pairs/s for the all-vs-all device pipeline, and the oracle's CPU time on a few pairs.
"""
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pyani_plus_amd.engine import HipEngine  # noqa: E402
from pyani_plus_amd.synth import arena_to_ascii, device_arena_to_host, synth_arena_torch  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
n_cpu = int(sys.argv[2]) if len(sys.argv) > 2 else 4
length, k, frag = 5_000_000, 16, 3000
eng = HipEngine(0)
arena = synth_arena_torch(eng, n, length)
starts = arena.genome_start[:-1].copy()
lens = np.full(n, length, dtype=np.uint32)
genome = np.arange(n, dtype=np.uint32)
t = eng.torch
eng.prof_enable(True)
for rep in range(2):
    eng.prof_reset()
    t.cuda.synchronize()
    t0 = time.perf_counter()
    total, matched, ident_sum = eng.fragani(arena, starts, lens, genome, k, frag)
    dt = time.perf_counter() - t0
    print(f"rep {rep}: {n}x{n} pairs in {dt:.3f} s -> {n * n / dt:.3e} pairs/s", {k: round(v[0], 1) for k, v in eng.prof_get().items() if k.startswith("frag")}, flush=True)
ani = np.where(matched > 0, ident_sum / np.maximum(matched, 1), np.nan)
related = ~np.isnan(ani)
print("fragments per genome", int(total[0]), "pairs with mappings", int(related.sum()), "ANI range", float(np.nanmin(ani)), float(np.nanmax(ani)))
assert np.all(np.diag(matched) >= 0.99 * total)
out = {"n": n, "seconds": dt, "pairs_per_s": n * n / dt}
if n_cpu:
    import oracle

    host = device_arena_to_host(arena, list(range(min(n, 41))), length)
    seqs = {g: arena_to_ascii(host, g) for g in (0, min(n - 1, 40))}
    g1 = min(n - 1, 40)
    t0 = time.perf_counter()
    res = [oracle.fragani_pair([seqs[a]], [seqs[b]], k, frag, 0.0) for a, b in ((0, g1), (g1, 0), (0, 0), (g1, g1))[:n_cpu]]
    cpu = (time.perf_counter() - t0) / len(res)
    for (a, b), (o_ani, o_m, o_t) in zip(((0, g1), (g1, 0), (0, 0), (g1, g1)), res):
        assert matched[a, b] == o_m and total[a] == o_t and abs(ani[a, b] - o_ani) < 1e-7, (a, b, matched[a, b], o_m)
    print(f"oracle: {cpu:.2f} s per pair on one core; parity on {len(res)} pairs ok")
    out["cpu_s_per_pair_1core"] = cpu
print(json.dumps(out))
