"""Time the pair phase at the per-GPU size of BASELINE configs[2] (N=10^4, one rank's 1250 subject columns)."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pyani_plus_amd.engine import DeviceSketches, HipEngine

n, size, cols = int(sys.argv[1]) if len(sys.argv) > 1 else 10000, 5000, 1250
eng = HipEngine(0); t = eng.torch
# 400 species-like pools so that hashes are shared (U << P)
g = t.Generator(device=eng.device); g.manual_seed(1)
pools = t.randint(0, 2**54, (400, 6000), generator=g, device=eng.device, dtype=t.int64)
rows = []
for i in range(n):
    perm = t.randperm(6000, generator=g, device=eng.device)[:size]
    rows.append(t.sort(pools[i % 400][perm]).values)
hashes = t.cat(rows)
# make lists strictly ascending & unique per genome (pool draws can collide rarely): fix by host check on a sample
off = t.arange(0, (n + 1) * size, size, dtype=t.int64, device=eng.device)
sk = DeviceSketches(hashes, off, n, n * size)
eng.prof_enable(True)
for rep in range(3):
    eng.prof_reset(); t.cuda.synchronize(); t0 = time.perf_counter()
    counts = eng.pair_counts(sk, (0, n), (0, cols))
    ident, cov = eng.ani(counts, sk, 31, (0, n), (0, cols))
    t.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"rep {rep}: {dt*1e3:.2f} ms for {n}x{cols} pairs", {k: round(v[0], 3) for k, v in eng.prof_get().items() if v[1]})
print("diag ok:", bool((counts[:cols, :cols].diagonal() == size).all()), "mem GB", t.cuda.max_memory_allocated() / 1e9)
