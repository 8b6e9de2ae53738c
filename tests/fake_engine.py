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
    """Host stand-in for ``DeviceSketches``: same attributes (``hashes``/``off`` are CPU tensors built on demand)."""

    def __init__(self, sketches: list[np.ndarray]):
        self.sketches = [np.asarray(s, dtype=np.uint64) for s in sketches]
        self.n = len(sketches)
        self.total = int(sum(len(s) for s in sketches))

    def to_host(self):
        return [s.copy() for s in self.sketches]

    def sizes(self):
        return np.array([len(s) for s in self.sketches], dtype=np.uint64)

    @property
    def hashes(self):
        import torch

        flat = np.concatenate(self.sketches) if self.total else np.zeros(1, dtype=np.uint64)
        return torch.from_numpy(flat.view(np.int64).copy())

    @property
    def off(self):
        import torch

        off = np.zeros(self.n + 1, dtype=np.int64)
        np.cumsum([len(s) for s in self.sketches], out=off[1:])
        return torch.from_numpy(off)


class OracleEngine:
    device = "cpu"  # where ``distributed.sharded_pair_step`` moves the gathered tensors

    def upload(self, arena: HostArena):
        return arena

    def sketches_from_gathered(self, hashes, off, off_host):
        flat = hashes.cpu().numpy().view(np.uint64)
        off_host = np.asarray(off_host, dtype=np.int64)
        return _Sketches([flat[off_host[g] : off_host[g + 1]] for g in range(len(off_host) - 1)])

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


    def fragani(self, arena: HostArena, contig_start, contig_len, contig_genome, k: int = 16, frag_len: int = 3000, ref_range=None,
                query_range=None, reuse_index: bool = False, out=None, columns_only: bool = False):
        """Fragment ANI of every ordered pair through the oracle: (total[n], matched[n, n], ident_sum[n, n])."""
        self.fragani_calls = getattr(self, "fragani_calls", [])
        self.fragani_calls.append({"ref_range": ref_range, "query_range": query_range, "reuse_index": reuse_index})
        n = arena.n_genomes
        contigs = [[] for _ in range(n)]
        for start, length, g in zip(contig_start, contig_len, contig_genome):
            g0 = int(arena.genome_start[int(g)])
            text = arena_to_ascii(HostArena(arena.packed, arena.mask, arena.genome_start), int(g))
            contigs[int(g)].append(text[int(start) - g0 : int(start) - g0 + int(length)])
        r0, r1 = (0, n) if ref_range is None else ref_range
        q0, q1 = (0, n) if query_range is None else query_range
        c0, width = (r0, r1 - r0) if columns_only else (0, n)
        if out is None:
            out = (np.zeros(n, dtype=np.uint32), np.zeros((n, width), dtype=np.uint32), np.zeros((n, width), dtype=np.float64))
        total, matched, ident_sum = out
        for q in range(n):
            total[q] = sum(len(c) // frag_len for c in contigs[q])
        matched[q0:q1] = 0
        ident_sum[q0:q1] = 0.0
        for q in range(q0, q1):
            for r in range(r0, r1):
                ani, m, t = oracle.fragani_pair(contigs[q], contigs[r], k, frag_len, 0.0)
                assert t == total[q]
                if m:
                    matched[q, r - c0] = m
                    ident_sum[q, r - c0] = _float_sum_with_mean(ani, m)
        return total, matched, ident_sum


def _float_sum_with_mean(ani: float, m: int) -> float:
    """A float whose float quotient by ``m`` is the oracle's (float) mean: what the library's ``ident_sum`` holds."""
    want, mf = np.float32(ani), np.float32(m)
    guess = want * mf
    up = down = guess
    for _ in range(4):  # (the oracle's mean IS such a quotient: the sum is within a few float steps of mean x count)
        if up / mf == want:
            return float(up)
        if down / mf == want:
            return float(down)
        up, down = np.nextafter(up, np.float32(np.inf)), np.nextafter(down, np.float32(-np.inf))
    return float(guess)


class SlowOracleEngine(OracleEngine):
    """``OracleEngine`` whose fragment-ANI calls take at least ``PYANI_TEST_FRAGANI_SLEEP`` seconds each (default 1.5):
    the interrupt tests need query batches that last long enough to be interrupted between them."""

    def fragani(self, *args, **kwargs):
        import os
        import time

        out = super().fragani(*args, **kwargs)
        time.sleep(float(os.environ.get("PYANI_TEST_FRAGANI_SLEEP", "1.5")))
        return out
