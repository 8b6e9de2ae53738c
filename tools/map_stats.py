"""Event counts of map_segments_kernel (segments, candidates, groups, rounds, windows, what the tight bound drops).

    make -C pyani_plus_amd/csrc stats          # libpyani_hip_stats.so: the tools build with -DPA_MAP_STATS
    python tools/map_stats.py [n_genomes=1000] [query genomes=78]

One batch of 2^17 query fragments (78 genomes of 5 Mb) against the index of all n, as the counter passes of rocprofv3
take it; the counts are printed by the library (PA_FRAGANI_TRACE) on stderr.
"""
import os
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pyani_plus_amd import _capi  # noqa: E402

_capi.TOOLS_LIB_PATH = _capi.TOOLS_LIB_PATH.with_name("libpyani_hip_stats.so")
from pyani_plus_amd.engine import HipEngine  # noqa: E402
from pyani_plus_amd.synth import synth_arena_torch  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n_query = int(sys.argv[2]) if len(sys.argv) > 2 else 78
length, k, frag = 5_000_000, 16, 3000
os.environ["PA_FRAGANI_TRACE"] = "1"
eng = HipEngine(0, tools=True)
if os.environ.get("PA_SYNTH") == "rearranged":  # the set with indels, inversions, repeat families and 30-200 contigs per genome
    from pyani_plus_amd.synth import synth_rearranged_arena_torch  # noqa: E402

    arena, starts, lens, genome = synth_rearranged_arena_torch(eng, n, length)
else:
    arena = synth_arena_torch(eng, n, length)
    starts = arena.genome_start[:-1].copy()
    lens = np.full(n, length, dtype=np.uint32)
    genome = np.arange(n, dtype=np.uint32)
eng.fragani(arena, starts, lens, genome, k, frag, query_range=(0, min(n_query, n)))
