"""An oracle-backed stand-in for HipEngine, for CPU tests of the HOST logic only.

Lives under tests/ on purpose: the product (pyani_plus_amd/) never routes through the
oracle; these shims let the plugin / driver / distributed code paths run without a GPU.
"""

from __future__ import annotations

import numpy as np

import oracle
from pyani_plus_amd.engine import HostArena
from pyani_plus_amd.synth import arena_to_ascii


class _HostTensor:
    """Looks enough like a torch tensor for ``.cpu().numpy().view(np.uint32)``."""

    def __init__(self, arr: np.ndarray):
        self.arr = arr

    def cpu(self):
        return self

    def numpy(self):
        return self.arr.view(np.int32) if self.arr.dtype == np.uint32 else self.arr


class _Sketches:
    def __init__(self, sketches: list[np.ndarray]):
        self.sketches = [np.asarray(s, dtype=np.uint64) for s in sketches]
        self.n = len(sketches)
        self.total = int(sum(len(s) for s in sketches))

    def to_host(self):
        return [s.copy() for s in self.sketches]

    def sizes(self):
        return np.array([len(s) for s in self.sketches], dtype=np.uint64)


class OracleEngine:
    def upload(self, arena: HostArena):
        return arena

    def sketch(self, arena: HostArena, k: int, scaled: int, *, max_hash=None):
        out = []
        for g in range(arena.n_genomes):
            s, e = int(arena.genome_start[g]), int(arena.genome_start[g + 1])
            tmp = HostArena(arena.packed[s // 16 : e // 16], arena.mask[s // 32 : e // 32], np.array([0, e - s], dtype=np.uint64), residues=[e - s])
            out.append(oracle.sketch_seq(arena_to_ascii(tmp, 0), k, scaled))
        return _Sketches(out)

    def sketches_from_host(self, sketches):
        return _Sketches(sketches)

    def pair_counts(self, sk: _Sketches, q_range=None, s_range=None, algo=0):
        return _HostTensor(oracle.pair_counts(sk.sketches, q_range, s_range))
