import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import oracle
from pyani_plus_amd.engine import HipEngine, pack_genomes
from tests.test_gpu_fragani import _random_genomes

k, frag = 15, 1000
texts, contig_lists = _random_genomes(7)
eng = HipEngine(0)
bad = 0
for qg in range(len(texts)):
  for rg in range(len(texts)):
    ani, m, t = oracle.fragani_pair(contig_lists[qg], contig_lists[rg], k, frag, 0.0)
    maps, total = oracle.fragani_map(contig_lists[qg], contig_lists[rg], k, frag)
    # single-fragment genomes
    frags = []
    for c in contig_lists[qg]:
        for f in range(len(c) // frag):
            frags.append(c[f * frag:(f + 1) * frag])
    ref_text = texts[rg]
    arena = pack_genomes([ref_text] + [b">f\n" + f + b"\n" for f in frags])
    tot, matched, isum = eng.fragani(eng.upload(arena), arena.contig_start, arena.contig_len, arena.contig_genome, k, frag)
    omap = {int(f): (int(sh), int(s), int(rs), int(rp)) for f, sh, s, rs, rp in zip(maps["frag"], maps["shared"], maps["s"], maps["ref_seq"], maps["ref_pos"])}
    for i in range(len(frags)):
        gm = int(matched[i + 1, 0])
        o = omap.get(i)
        if (o is None) != (gm == 0):
            print("pair", qg, rg, "frag", i, "oracle", o, "gpu matched", gm); bad += 1
        elif o is not None:
            want = oracle.fragani_identity(o[0], o[1], k)
            if abs(isum[i + 1, 0] - want) > 1e-9:
                print("pair", qg, rg, "frag", i, "oracle shared/s", o, "ident", want, "gpu ident", isum[i + 1, 0]); bad += 1
print("mismatches", bad)
