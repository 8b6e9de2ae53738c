"""BASELINE configs[0] through this build's own driver: FASTA files on disk -> SQLite database with all N^2
comparisons and cached matrices (the counterpart of `pyani-plus sourmash <dir> -d <db> --create-db`).

    python tools/config1_file.py            # the 4 gz bacteria of tests/golden/bacterial_example
Prints T_file; that the matrices equal the reference's is asserted by tests/test_gpu_method.py.
"""
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from pyani_plus_amd import rundb  # noqa: E402
from tests.helpers import GOLDEN  # noqa: E402

fasta = GOLDEN / "bacterial_example"
with tempfile.TemporaryDirectory() as tmp:
    for rep in range(2):
        db = Path(tmp) / f"run{rep}.sqlite"
        t0 = time.perf_counter()
        run = rundb.run_sourmash_hip(fasta, db)
        dt = time.perf_counter() - t0
        print(f"rep {rep}: T_file = {dt:.3f} s for {len(run.fasta_hashes)} genomes ({len(run.fasta_hashes) ** 2} comparisons)")
