#!/bin/bash
# Per-kernel time of one kernel under several builds of the library on ONE box (rocprofv3 --kernel-trace --stats of one batch
# of queries against the 1 000-genome index):   bash tools/ab_kernel.sh <kernel name filter> <libA.so> <libB.so> ...
F=$1; shift
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  D=/tmp/abk_$$_$(basename $L .so)
  PA_AB_LIB=$ROOT/$L rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $ROOT/tools/bench_fragani.py 1000 0 interleaved 78 > $D.log 2>&1
  f=$(find $D -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$F" "$L" <<'PY'
import csv, sys
f, flt, lib = sys.argv[1:4]
for r in csv.DictReader(open(f)):
    if flt in r["Name"]:
        print(f"== {lib}: {r['Name'][:60]} calls {r['Calls']} avg_ms {float(r['AverageNs'])/1e6:.3f}")
PY
  grep "^rep 1" $D.log
  rm -rf $D $D.log
done
