"""Where the time of map_segments_kernel goes: the kernel cut short after each phase (PA_MAP_CUT), timed by difference.

    python tools/map_cut.py [n_genomes=1000] [cuts, e.g. 9,9,4,9]
"""
import os
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pyani_plus_amd.engine import HipEngine  # noqa: E402
from pyani_plus_amd.synth import synth_arena_torch  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
length, k, frag = 5_000_000, 16, 3000
eng = HipEngine(0, tools=True)  # PA_MAP_CUT exists in the -DPA_TOOLS build of the library only
arena = synth_arena_torch(eng, n, length)
starts = arena.genome_start[:-1].copy()
lens = np.full(n, length, dtype=np.uint32)
genome = np.arange(n, dtype=np.uint32)
eng.prof_enable(True)
# the phases and their numbers come from the kernel's own list (enum MapCut in csrc/fragani_map.inc)
import re  # noqa: E402

_src = (Path(__file__).resolve().parent.parent / "pyani_plus_amd" / "csrc" / "fragani_map.inc").read_text()
_enum = _src[_src.index("enum MapCut"):]
_enum = _enum[: _enum.index("};")]
names = {int(m.group(2)): f"{m.group(1)}: {m.group(3).strip()}" for m in re.finditer(r"(kCut\w+) = (\d+),\s*// (.*)", _enum)}
assert {9, 10, 11, 1, 2, 3, 4}.issubset(names), names
prev = 0.0
cuts = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [9, 9, 10, 11, 1, 2, 3, 4, 5, 6, 7, 8, 9]
for cut in cuts:
    os.environ["PA_MAP_CUT"] = str(cut)
    eng.prof_reset()
    t0 = time.perf_counter()
    eng.fragani(arena, starts, lens, genome, k, frag)
    dt = time.perf_counter() - t0
    ms = eng.prof_get()["frag_map"][0]
    print(f"cut {cut}: frag_map {ms:8.1f} ms  (+{ms - prev:7.1f})  {names[cut]}   [run {dt:.3f} s]", flush=True)
    prev = ms if cut != 9 or prev else prev
    if cut == 9 or cut >= 20:
        prev = 0.0
