"""Fragment ANI under another mix of species: what a run costs follows the RELATED pairs (a fragment maps where it has seed
hits), not the pairs.

    python tools/species_mix.py <n_genomes> <n_species>
1000 40 is the benchmark (25 000 related ordered pairs); 200 1 and 400 4 hold 40 000 each; 1000 1000 holds the diagonal only."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pyani_plus_amd.engine import HipEngine  # noqa: E402
from pyani_plus_amd.synth import synth_arena_torch  # noqa: E402

n = int(sys.argv[1])
sp = int(sys.argv[2])
length, k, frag = 5_000_000, 16, 3000
eng = HipEngine(0)
arena = synth_arena_torch(eng, n, length, n_species=sp)
starts = arena.genome_start[:-1].copy()
lens = np.full(n, length, dtype=np.uint32)
genome = np.arange(n, dtype=np.uint32)
eng.prof_enable(True)
for rep in range(2):
    eng.prof_reset()
    eng.torch.cuda.synchronize()
    t0 = time.perf_counter()
    total, matched, ident_sum = eng.fragani(arena, starts, lens, genome, k, frag)
    dt = time.perf_counter() - t0
    related = int((matched > 0.5 * total[:, None]).sum())
    print(f"rep {rep}: {n} genomes of {sp} species: {dt:.3f} s -> {n * n / dt:.3e} pairs/s, {related} related ordered pairs -> {related / dt:.3e} of them per s",
          {k_: round(v[0], 1) for k_, v in eng.prof_get().items() if k_.startswith("frag")}, flush=True)
