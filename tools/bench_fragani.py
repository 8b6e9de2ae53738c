"""Time the fastANI-style fragment-ANI path (BASELINE configs[3]) on synthetic 5 Mb genomes.

    python tools/bench_fragani.py [n_genomes] [0] [interleaved|grouped] [query genomes] [reference genomes] [contigs per genome]
Prints pairs/s for the all-vs-all device pipeline (the CPU figure and the parity check against the oracle are
bench.py's `also.fragment_ani` leg and tests/test_gpu_fragani.py).  With a fourth argument only that many query
genomes are mapped (against the index of all n): 78 of them are one batch of 2^17 fragments -- the form the counter
passes of rocprofv3 take at the benchmark's 1 000 genomes (one dispatch of every kernel per repetition).  With a fifth,
the queries are mapped against the first that many genomes only, results as columns: 1 = what the reference's worker asks
for, one subject column (the first repetition includes the allocation of the workspace, as a worker process's only call does).
"""
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import os  # noqa: E402

from pyani_plus_amd import _capi  # noqa: E402

if os.environ.get("PA_AB_LIB"):  # tools/ab_fragani.sh: time another build of the library (A/B on one box)
    _capi.LIB_PATH = Path(os.environ["PA_AB_LIB"]).resolve()
from pyani_plus_amd.engine import HipEngine  # noqa: E402
from pyani_plus_amd.synth import synth_arena_torch  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
order = sys.argv[3] if len(sys.argv) > 3 else "interleaved"  # "grouped": the genomes of one species next to each other
n_query = int(sys.argv[4]) if len(sys.argv) > 4 else n
n_ref = int(sys.argv[5]) if len(sys.argv) > 5 else n
n_contigs = int(sys.argv[6]) if len(sys.argv) > 6 else 1
length, k, frag = 5_000_000, 16, 3000
eng = HipEngine(0)
ids = None
if order == "grouped":
    ids = sorted(range(n), key=lambda g: (g % 40, g))
rearranged = os.environ.get("PA_SYNTH") == "rearranged"  # the set with indels, inversions, repeat families and 30-200 contigs per genome
if rearranged:
    from pyani_plus_amd.synth import synth_rearranged_arena_torch  # noqa: E402

    arena, starts, lens, genome = synth_rearranged_arena_torch(eng, n, length, genome_ids=ids)
else:
    arena = synth_arena_torch(eng, n, length, genome_ids=ids)
    starts = arena.genome_start[:-1].copy()
    lens = np.full(n, length, dtype=np.uint32)
    genome = np.arange(n, dtype=np.uint32)
if n_contigs > 1 and not rearranged:
    piece = length // n_contigs
    starts = (starts[:, None] + (np.arange(n_contigs, dtype=np.uint64) * np.uint64(piece))[None, :]).reshape(-1)
    lens = np.full(n * n_contigs, piece, dtype=np.uint32)
    genome = np.repeat(genome, n_contigs)
t = eng.torch
eng.prof_enable(True)
for rep in range(2):
    eng.prof_reset()
    t.cuda.synchronize()
    t0 = time.perf_counter()
    total, matched, ident_sum = eng.fragani(arena, starts, lens, genome, k, frag, query_range=(0, n_query),
                                            ref_range=None if n_ref == n else (0, n_ref), columns_only=n_ref != n)
    dt = time.perf_counter() - t0
    print(f"rep {rep}: {n_query}x{n_ref} pairs in {dt:.3f} s -> {n_query * n_ref / dt:.3e} pairs/s", {k: round(v[0], 1) for k, v in eng.prof_get().items() if k.startswith("frag")}, flush=True)
from pyani_plus_amd.methods.fastani_hip import fastani_mean  # noqa: E402

ani = fastani_mean(ident_sum, matched)
related = ~np.isnan(ani)
print("fragments per genome", int(total[0]) if not rearranged else (int(total.min()), int(total.max())), "pairs with mappings", int(related.sum()), "ANI range", float(np.nanmin(ani)), float(np.nanmax(ani)))
# (a draft of 10 kb contigs keeps 69 of its own 72 fragments per 24 contigs -- the slide's end rule at every contig's end, as the oracle: tests/test_gpu_fragani.py)
assert np.all(np.diag(matched)[: min(n_query, n_ref)] >= (0.99 if n_contigs == 1 and not rearranged else 0.9) * total[: min(n_query, n_ref)])
import hashlib  # noqa: E402

# every integer and every float sum of the run in one line: two builds of the library agree on it or they differ somewhere
digest = hashlib.sha256(total.tobytes() + matched.tobytes() + ident_sum.tobytes()).hexdigest()[:16]
print("results sha256/16", digest, "kept fragments", int(matched.sum()))
out = {"n": n, "query_genomes": n_query, "reference_genomes": n_ref, "seconds": dt, "pairs_per_s": n_query * n_ref / dt, "results_sha16": digest}
print(json.dumps(out))
