#!/usr/bin/env python3
"""Randomised differential test of the fragment-ANI path: device against oracle on small genome sets with everything that has
broken it before -- repeats (tandem and dispersed), runs of N and single N, IUPAC codes and other bytes (hashed as the
characters they are), contigs shorter than a fragment, contigs that end with their last fragment, lower case, several k
and fragment lengths.

    python tests/tools/fragani_stress.py [cases=200] [seed=1] [big]

`big`: genomes of 1.4 to 3 Mb with arrays of short repeats and homopolymer runs, so that Mashmap's frequency cut of the seeds
is active (it needs 100 000 distinct minimizers before it ignores one), and a small mutated excerpt as the second genome.

Prints one line per failing case (and stops after ten); exit code 1 if any.  Needs a GPU; the oracle is the checker (test
infrastructure, like the rest of tests/)."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import oracle  # noqa: E402
from pyani_plus_amd.engine import HipEngine, pack_genomes  # noqa: E402
from pyani_plus_amd.methods.fastani_hip import fastani_mean  # noqa: E402

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
IUPAC = np.frombuffer(b"RYKMSWBDHVrykmX*", dtype=np.uint8)


def make_case(rng):
    k = int(rng.choice([16, 16, 15, 14, 12]))
    frag = int(rng.choice([3000, 3000, 2000, 1000, 500]))
    root = rng.choice(ACGT, size=int(rng.integers(8, 30)) * frag + int(rng.integers(0, frag)))
    genomes = []
    for _ in range(int(rng.integers(2, 5))):
        g = root.copy()
        rate = float(rng.choice([0.0, 0.001, 0.01, 0.03, 0.08, 0.15]))
        hit = rng.random(g.size) < rate
        g[hit] = ACGT[rng.integers(0, 4, size=int(hit.sum()))]
        g = bytearray(g.tobytes())
        for _ in range(int(rng.integers(0, 4))):  # tandem repeats
            unit = rng.choice(ACGT, size=int(rng.integers(1, 60))).tobytes()
            at = int(rng.integers(0, len(g)))
            g[at:at] = unit * int(rng.integers(2, 80))
        for _ in range(int(rng.integers(0, 3))):  # dispersed copies
            n = int(rng.integers(200, 3 * frag))
            src = int(rng.integers(0, max(1, len(g) - n)))
            at = int(rng.integers(0, len(g)))
            g[at:at] = g[src : src + n]
        for _ in range(int(rng.integers(0, 4))):  # unknown residues
            n = int(rng.choice([1, 1, 2, 17, 40, 300, frag + 7, 2 * frag + 100]))
            at = int(rng.integers(0, max(1, len(g) - n)))
            g[at : at + n] = b"N" * n
        if rng.random() < 0.5:  # IUPAC codes and other bytes, hashed as the characters they are: single ones and short runs
            for _ in range(int(rng.integers(1, 40))):
                n = int(rng.choice([1, 1, 1, 2, 5, 20]))
                at = int(rng.integers(0, max(1, len(g) - n)))
                g[at : at + n] = bytes(rng.choice(IUPAC, size=n).tolist())
        if rng.random() < 0.3:
            at = int(rng.integers(0, len(g) - 100))
            g[at : at + 100] = bytes(g[at : at + 100]).lower()
        cuts = sorted(set(int(x) for x in rng.integers(1, len(g), size=int(rng.integers(0, 4)))))
        if rng.random() < 0.3:  # a contig that ends with its last fragment
            cuts = sorted(set(cuts + [int(rng.integers(1, 4)) * frag]))
        contigs = [bytes(g[a:b]) for a, b in zip([0] + cuts, cuts + [len(g)]) if b > a]
        genomes.append(contigs)
    if rng.random() < 0.5:
        genomes.append([rng.choice(ACGT, size=int(rng.integers(1, 6)) * frag).tobytes()])
    return k, frag, genomes


def make_big_case(rng):
    n = int(rng.integers(1_400_000, 3_000_000))
    g = bytearray(rng.choice(ACGT, size=n).tobytes())
    arrays = []
    for _ in range(int(rng.integers(1, 6))):
        unit = rng.choice(ACGT, size=int(rng.integers(1, 5))).tobytes()
        copies = int(rng.integers(60, 400))
        at = int(rng.integers(10_000, len(g) - 10_000))
        g[at:at] = unit * copies
        arrays.append(at)
    for _ in range(int(rng.integers(0, 3))):  # a dispersed family: one 1.2 kb element at 30 to 60 places
        el = rng.choice(ACGT, size=1_200).tobytes()
        for _ in range(int(rng.integers(30, 60))):
            at = int(rng.integers(0, len(g)))
            g[at:at] = el
    big = bytes(g)
    around = arrays[int(rng.integers(0, len(arrays)))]
    lo = max(0, around - int(rng.integers(6_000, 30_000)))
    part = np.frombuffer(big[lo : lo + int(rng.integers(30_000, 90_000))], dtype=np.uint8).copy()
    hit = rng.random(part.size) < float(rng.choice([0.0, 0.01, 0.05, 0.1]))
    part[hit] = ACGT[rng.integers(0, 4, size=int(hit.sum()))]
    return 16, 3000, [[big], [part.tobytes()]]


def main() -> int:
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    big = len(sys.argv) > 3 and sys.argv[3] == "big"
    rng = np.random.default_rng(seed)
    eng = HipEngine(0)
    bad = 0
    t0 = time.time()
    for case in range(cases):
        k, frag, genomes = make_big_case(rng) if big else make_case(rng)
        if oracle.fragani_window_size(k, frag) > 64:
            continue
        texts = [b"".join(b">c%d\n" % i + c + b"\n" for i, c in enumerate(contigs)) for contigs in genomes]
        arena = pack_genomes(texts)
        total, matched, ident_sum = eng.fragani(eng.upload(arena), arena.contig_start, arena.contig_len, arena.contig_genome, k, frag)
        n = len(genomes)
        for q in range(n):
            for r in range(n):
                ani, m, t = oracle.fragani_pair(genomes[q], genomes[r], k, frag, 0.0)
                got = float(fastani_mean(ident_sum[q, r], matched[q, r])) if matched[q, r] else float("nan")
                if (int(total[q]), int(matched[q, r])) != (t, m) or (m and got != ani):
                    print(f"case {case} (k={k} frag={frag}) pair ({q},{r}): device {int(total[q])} {int(matched[q, r])} {got} oracle {t} {m} {ani}", flush=True)
                    bad += 1
        if n > 1:  # one subject column, and a range of them: the dictionary of those genomes' minimizers only
            r0 = int(rng.integers(0, n))
            r1 = int(rng.integers(r0 + 1, n + 1))
            for a, b in ((r0, r0 + 1), (r0, r1)):
                t2, m2, s2 = eng.fragani(eng.upload(arena), arena.contig_start, arena.contig_len, arena.contig_genome, k, frag,
                                         ref_range=(a, b), columns_only=True)
                if not (np.array_equal(t2, total) and np.array_equal(m2, matched[:, a:b]) and np.array_equal(s2, ident_sum[:, a:b])):
                    print(f"case {case} (k={k} frag={frag}): columns [{a}, {b}) differ from the all-against-all run", flush=True)
                    bad += 1
        if bad >= 10:
            break
    print(f"{cases} cases, {bad} differing pairs, {time.time() - t0:.0f} s")
    return 1 if bad else 0


if __name__ == "__main__":
    raise SystemExit(main())
