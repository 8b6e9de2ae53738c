#!/bin/bash
# Vector / scalar / LDS instructions of map_segments_kernel<320, true> PER PHASE: the kernel cut short after each phase
# (PA_MAP_CUT, tools build; enum MapCut of csrc/fragani_map.inc) under rocprofv3 --pmc, one batch of 2^17 query fragments
# against the 1 000-genome index; a phase's instructions are the difference of two cuts.  (Time by difference is
# tools/map_cut.py; this is the instruction counter per phase that reconciles the work model with the phase cuts.)
#   bash tools/map_cut_valu.sh <tag>      -> gpurun_out/<tag>_map_cut_valu.txt
TAG=${1:-r06}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/${TAG}_map_cut_valu.txt
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
: > $OUT
for CUT in 10 11 1 2 3 4 5 7 8 24 23 9; do
  D=/tmp/mcv_$$_$CUT
  PA_MAP_CUT=$CUT PA_AB_LIB=$ROOT/pyani_plus_amd/_lib/libpyani_hip_tools.so rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES \
    --kernel-include-regex "map_segments_kernel<320u, true>" --kernel-trace --output-format csv -d $D -- python3 $ROOT/tools/bench_fragani.py 1000 0 interleaved 78 > $D.log 2>&1
  python3 - $D $CUT >> $OUT <<'PY'
import csv, sys
from collections import defaultdict
from pathlib import Path
root, cut = Path(sys.argv[1]), sys.argv[2]
per = defaultdict(lambda: defaultdict(float))
for f in root.rglob("*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if "map_segments_kernel<320u, true>" in row["Kernel_Name"]:
            per[row["Dispatch_Id"]][row["Counter_Name"]] += float(row["Counter_Value"])
big = [d for d in per.values() if d.get("SQ_WAVES", 0) >= 0.1 * max(x.get("SQ_WAVES", 0) for x in per.values())]
use = big[1:] if len(big) > 1 else big
mean = {k: sum(d[k] for d in use) / len(use) for k in use[0]}
print(f"cut {cut:>2s}: " + "  ".join(f"{k} {v:.6g}" for k, v in sorted(mean.items())) + f"  ({len(use)} full dispatches averaged)")
PY
  rm -rf $D $D.log
done
cat $OUT
