"""Time the pair phase of BASELINE configs[2] on one GPU: N = 10^4 sketches of 5 000 hashes, all-vs-all over five
2 048-column tiles (tile pairs on and above the diagonal evaluated, the rest mirrored).

    python tools/pair_tiles.py [n=10000] [species=40]
Sketches are made per species so that genomes share hashes the way related genomes do (no genomes are hashed)."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pyani_plus_amd.engine import DeviceSketches, HipEngine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
species = int(sys.argv[2]) if len(sys.argv) > 2 else 40
size = 5000
eng = HipEngine(0)
t = eng.torch
g = t.Generator(device=eng.device)
g.manual_seed(1)
# a genome keeps a hash of its species' root with probability (1 - d)^31, d = its substitution rate (0.1 % ... 20 %,
# as the synthetic genomes of bench.py), and has a hash of its own otherwise
roots = t.randint(0, 2**54, (species, size), generator=g, device=eng.device, dtype=t.int64)
rates = [0.001, 0.002, 0.005, 0.01, 0.02, 0.03, 0.05, 0.08, 0.12, 0.2]
idx = t.arange(n, device=eng.device)
p_keep = (1.0 - t.tensor(rates, device=eng.device)[(idx // species) % len(rates)]) ** 31
keep = t.rand((n, size), generator=g, device=eng.device) < p_keep[:, None]
own = t.randint(2**54, 2**62, (n, size), generator=g, device=eng.device, dtype=t.int64)
hashes = t.sort(t.where(keep, roots[idx % species], own), dim=1).values.reshape(-1).contiguous()
off = t.arange(0, (n + 1) * size, size, dtype=t.int64, device=eng.device)
sk = DeviceSketches(hashes, off, n, n * size)
eng.prof_enable(True)
for rep in range(4):
    eng.prof_reset()
    t.cuda.synchronize()
    t0 = time.perf_counter()
    counts = eng.pair_counts(sk)
    t.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"rep {rep}: {dt * 1e3:.2f} ms for {n}x{n} pairs", {k: round(v[0], 3) for k, v in eng.prof_get().items() if v[1]}, flush=True)
c = counts.view(t.int32) if counts.dtype != t.int32 else counts
print("diagonal ok:", bool((c.diagonal() == size).all()), "symmetric:", bool((c[:512, -512:] == c[-512:, :512].T).all()))
