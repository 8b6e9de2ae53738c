"""Where the time of the batched front-end goes: the steps of sourmash_hip.sketch_fasta_batches,
timed one by one on synthetic files, without the overlap.

    python tools/front_end_pipeline.py [n_files=1000] [length=5000000]
"""
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from pyani_plus_amd.engine import HipEngine, load_fasta_files, max_hash_for_scaled  # noqa: E402
from pyani_plus_amd.methods import sourmash_hip  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
length = int(sys.argv[2]) if len(sys.argv) > 2 else 5_000_000
rng = np.random.default_rng(7)
acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
eng = HipEngine(0)
with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
    base = acgt[rng.integers(0, 4, size=length, dtype=np.uint8)]
    paths = []
    for g in range(n):
        seq = np.roll(base, g * 997)
        path = Path(tmp) / f"genome_{g:05d}.fasta"
        path.write_bytes(b">genome_%d synthetic\n" % g + b"\n".join(seq[i : i + 100_000].tobytes() for i in range(0, length, 100_000)) + b"\n")
        paths.append(path)
    per_batch = max(1, sourmash_hip.PREPARE_BATCH_BASES // length)
    for rep in range(2):
        t_all = time.perf_counter()
        steps = {"load": 0.0, "pin": 0.0, "sketch": 0.0, "to_host": 0.0}
        for b0 in range(0, n, per_batch):
            t0 = time.perf_counter()
            infos, arena = load_fasta_files(paths[b0 : b0 + per_batch], pinned=True)
            t1 = time.perf_counter()
            pinned = eng.pin_arena(arena)
            t2 = time.perf_counter()
            _dev, sk = eng.sketch_streamed(pinned, 31, 1000, max_hash=max_hash_for_scaled(1000))
            eng.sync()
            t3 = time.perf_counter()
            host = sk.to_host()
            t4 = time.perf_counter()
            for key, dt in zip(steps, (t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
                steps[key] += dt
        print(f"rep {rep}: {n} files in batches of {per_batch}: " + ", ".join(f"{k} {v:.3f}" for k, v in steps.items())
              + f"; total {time.perf_counter() - t_all:.3f} s (no overlap)", flush=True)
    t0 = time.perf_counter()
    import logging

    out = list(sourmash_hip.sketch_fasta_batches(logging.getLogger("x"), paths, kmersize=31, scaled=1000, engine=eng))
    print(f"sketch_fasta_batches (overlapped): {time.perf_counter() - t0:.3f} s")
