"""Does the seeding of one batch hide behind the mapping of another?  Two contexts on two HIP streams map one half of the
query genomes each, at the same time, against the same index parameters (each builds its own index: counted), and the wall
time is set against one context mapping all of them.

    python tools/overlap_probe.py [n_genomes=1000]
"""
import sys
import threading
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pyani_plus_amd.engine import HipEngine  # noqa: E402
from pyani_plus_amd.synth import synth_arena_torch  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
length, k, frag = 5_000_000, 16, 3000
engines = [HipEngine(0), HipEngine(0)]
t = engines[0].torch
streams = [t.cuda.Stream(device=0), t.cuda.Stream(device=0)]
for e, s in zip(engines, streams):
    with t.cuda.stream(s):
        e.use_torch_stream()
    e.prof_enable(True)
arena = synth_arena_torch(engines[0], n, length)
t.cuda.synchronize()
starts = arena.genome_start[:-1].copy()
lens = np.full(n, length, dtype=np.uint32)
genome = np.arange(n, dtype=np.uint32)


def run(e, q0, q1, out):
    t0 = time.perf_counter()
    out.append(e.fragani(arena, starts, lens, genome, k, frag, query_range=(q0, q1)))
    out.append(time.perf_counter() - t0)


for rep in range(3):
    for e in engines:
        e.prof_reset()
    t0 = time.perf_counter()
    res = []
    run(engines[0], 0, n, res)
    one = time.perf_counter() - t0
    phases = {k_: round(v[0], 1) for k_, v in engines[0].prof_get().items() if k_.startswith("frag")}
    for e in engines:
        e.prof_reset()
    outs = [[], []]
    th = [threading.Thread(target=run, args=(engines[i], i * n // 2, (i + 1) * n // 2, outs[i])) for i in range(2)]
    t0 = time.perf_counter()
    for x in th:
        x.start()
    for x in th:
        x.join()
    two = time.perf_counter() - t0
    ph2 = [{k_: round(v[0], 1) for k_, v in e.prof_get().items() if k_.startswith("frag")} for e in engines]
    print(f"rep {rep}: one context {one:.3f} s {phases}; two contexts at once {two:.3f} s {ph2}", flush=True)
    same = all(np.array_equal(np.concatenate([outs[0][0][j][: n // 2], outs[1][0][j][n // 2 :]]), res[0][j]) for j in (1, 2)) if res[0][1].shape[0] == n else None
    print("   results equal:", same)
