"""Lean harness for rocprofv3 --pmc passes: a handful of launches, no torch-heavy generator.

Random packed words are random genomes (HBM traffic of the hash kernel does not depend on
content).  Usage on the GPU box:
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- python3 tools/pmc_hash.py [n_genomes] [k]
"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pyani_plus_amd.engine import DeviceArena, HipEngine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 31
length = 5_000_000
padded = (length // 64 + 1) * 64
eng = HipEngine(0)
t = eng.torch
packed = t.randint(-(2**31), 2**31 - 1, (n * padded // 16,), dtype=t.int32, device=eng.device)
mask = t.zeros(n * padded // 32, dtype=t.int32, device=eng.device)
mask.view(n, padded // 32)[:, -2:] = -1  # the last 64 positions of every genome are padding
starts = (np.arange(n + 1, dtype=np.uint64) * np.uint64(padded)).astype(np.uint64)
arena = DeviceArena(packed, mask, starts)
for _ in range(2):
    sk = eng.sketch(arena, k, 1000)
t.cuda.synchronize()
print("genomes", n, "k", k, "hashes", sk.total)
eng.close()
