"""Child of tests/test_host_logic.py::test_environment_selectors_pick_equivalent_implementations: loads FASTA files
through the threaded front-end twice (the second batch takes the first one's slabs when the slab cache is on), packs a
text with the C packer, inserts a small matrix through the native SQLite route, and prints one JSON line of digests.
The selectors under test (PA_GUNZIP, PA_PACK_SCALAR, PA_HOST_SLAB_CACHE, PA_SQLITE_SYNCHRONOUS) are read from the
environment by libpyani_hip.so, three of them once per process: hence a process per setting.  Host code only."""
import hashlib
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from pyani_plus_amd import rundb  # noqa: E402
from pyani_plus_amd.engine import load_fasta_files, pack_genomes  # noqa: E402

work = Path(sys.argv[1])
paths = [Path(p) for p in sys.argv[2:]]
out = {}
for round_no in (0, 1):  # the second load reuses the first one's host slabs (unless PA_HOST_SLAB_CACHE=0)
    infos, arena = load_fasta_files(paths, threads=3)
    assert all(i.status == 0 for i in infos), [i.message for i in infos]
    digest = hashlib.sha256(arena.packed.tobytes() + arena.mask.tobytes() + arena.genome_start.tobytes()).hexdigest()
    out[f"load_{round_no}"] = {"md5": [i.md5 for i in infos], "length": [i.length for i in infos], "records": [i.records for i in infos],
                               "invalid": [i.invalid for i in infos], "arena": digest,
                               "ambiguous": hashlib.sha256(arena.ambig_pos.tobytes() + arena.ambig_byte.tobytes()).hexdigest()}
    del arena
# the packer on a text with clean runs, lower case, N runs, IUPAC letters, blanks and CR (vector and scalar forms)
rng = np.random.default_rng(11)
text = b">one\n" + bytes(rng.choice(list(b"ACGTacgt"), 5000).tolist()) + b"\nNNNNNNNNNNRYKM acgt\r\n" + bytes(rng.choice(list(b"ACGTN"), 777).tolist()) + b"\n>two x\nACGT\n"
host = pack_genomes([text, text[:301]])
out["pack"] = hashlib.sha256(host.packed.tobytes() + host.mask.tobytes() + host.ambig_pos.tobytes() + host.ambig_byte.tobytes()).hexdigest()
# native row insert
n = 23
hashes = sorted(hashlib.md5(str(i).encode()).hexdigest() for i in range(n))
ident, cov, null = rng.random((n, n)), rng.random((n, n)), rng.random((n, n)) < 0.2


class RunStub:
    configuration_id = 0


conn = rundb.connect_to_db(work / "rows.sqlite")
RunStub.configuration_id = rundb.db_configuration(conn, "sourmash-hip", "libpyani_hip", "0.1.0", kmersize=31, extra="scaled=1000").configuration_id
conn.commit()
assert rundb.ingest_matrices(conn, RunStub, hashes, hashes, ident, cov, null, native=True) == n * n
rows = conn.execute("SELECT comparison_id, query_hash, subject_hash, configuration_id, identity, cov_query FROM comparisons ORDER BY comparison_id").fetchall()
out["rows"] = hashlib.sha256(repr(rows).encode()).hexdigest()
out["n_rows"] = len(rows)
print(json.dumps(out))
