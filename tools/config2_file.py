"""Whole path on files at a size between BASELINE configs[0] and configs[1]: N synthetic 5 Mb genomes written
as FASTA files, then FASTA directory -> database with N^2 comparisons through rundb.run_sourmash_hip.

    python tools/config2_file.py [n_genomes=200] [length=5000000] [ingest=json|direct|fastani|fastani-direct] [reps=2] [gz]
(`fastani`: the same files through rundb.run_fastani_hip -- BASELINE configs[3] on files.)
Prints the wall time of the run and of its parts (threaded FASTA front-end, device work, column files, SQLite, matrix cache).
"""
import logging
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from pyani_plus_amd import rundb  # noqa: E402
from pyani_plus_amd.synth import RATES  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
length = int(sys.argv[2]) if len(sys.argv) > 2 else 5_000_000
ingest = sys.argv[3] if len(sys.argv) > 3 else "json"
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
gz = len(sys.argv) > 5 and sys.argv[5] == "gz"  # 80-column lines, zlib level 6: the form NCBI ships
rng = np.random.default_rng(20260802)
acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
roots = rng.integers(0, 4, size=(8, length), dtype=np.uint8)
logging.basicConfig(level=logging.WARNING)
with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
    fasta = Path(tmp) / "genomes"
    fasta.mkdir()
    t0 = time.perf_counter()
    for g in range(n):
        seq = roots[g % 8].copy()
        hit = rng.random(length) < RATES[(g // 8) % len(RATES)]
        seq[hit] = (seq[hit] + rng.integers(1, 4, size=int(hit.sum()), dtype=np.uint8)) & 3
        text = acgt[seq]
        width = 80 if gz else 100_000
        body = b">genome_%d synthetic\n" % g + b"\n".join(text[i : i + width].tobytes() for i in range(0, length, width)) + b"\n"
        if gz:
            import zlib

            comp = zlib.compressobj(6, zlib.DEFLATED, 31)
            body = comp.compress(body) + comp.flush()
        (fasta / (f"genome_{g:05d}.fasta" + (".gz" if gz else ""))).write_bytes(body)
    print(f"wrote {n} {'gzip ' if gz else ''}FASTA files ({n * length / 1e9:.2f} Gb) in {time.perf_counter() - t0:.1f} s", flush=True)
    for rep in range(reps):
        db = Path(tmp) / f"run{rep}.sqlite"
        t0 = time.perf_counter()
        timings = {}
        if ingest.startswith("fastani"):  # "fastani" (JSON column files imported) or "fastani-direct"
            run = rundb.run_fastani_hip(fasta, db, timings=timings, ingest="direct" if ingest.endswith("direct") else "json")
        else:
            run = rundb.run_sourmash_hip(fasta, db, ingest=ingest, timings=timings)
        dt = time.perf_counter() - t0
        print(f"rep {rep} ({ingest}): files -> database in {dt:.2f} s for {n} genomes = {n * n / dt:.3e} pairs/s end to end; "
              + ", ".join(f"{k} {v:.2f}" for k, v in timings.items()) + f"; database {db.stat().st_size / 1e6:.0f} MB", flush=True)
