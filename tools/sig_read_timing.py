#!/usr/bin/env python3
"""How long a column worker needs to load N cached signature files: native threaded reader
against the Python one, cold (first call of the process) and warm.  Host only."""
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from pyani_plus_amd import sig  # noqa: E402
from pyani_plus_amd.engine import max_hash_for_scaled  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
src = (ROOT / "tests/golden/bacterial_example/sourmash/073194224aa8c13bebc1d14a3e74a3e7.sig").read_bytes()
d = Path(tempfile.mkdtemp())
paths = []
for i in range(n):
    p = d / f"{i}.sig"
    p.write_bytes(src)
    paths.append(p)
mh = max_hash_for_scaled(1000)
for label in ("cold", "warm", "warm"):
    t = time.perf_counter()
    got = sig.read_sigs(paths, ksize=31, max_hash=mh)
    print(f"native reader, {n} files of {len(got[0])} hashes, {label}: {time.perf_counter() - t:.3f} s")
t = time.perf_counter()
for p in paths[: max(1, n // 10)]:
    sig.read_sig(p, ksize=31, max_hash=mh)
print(f"python reader: {(time.perf_counter() - t) / max(1, n // 10) * n:.3f} s for {n} files (from {max(1, n // 10)})")
