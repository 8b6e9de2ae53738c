#!/bin/bash
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r03h_stats -- python3 $ROOT/tools/bench_fragani.py 1000 0 > $ROOT/gpurun_out/r03h_fragani.log 2>&1
f=$(find $ROOT/gpurun_out/r03h_stats -name "*kernel_stats.csv" | head -1)
{ head -1 "$f"; grep -v "at::native\|rocclr\|hiprand" "$f" | tail -n +2; } > $ROOT/gpurun_out/r03h_fragani.kernel_stats.csv
rm -rf $ROOT/gpurun_out/r03h_stats
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$ROOT/gpurun_out/r03h_fragani.kernel_stats.csv")))
for r in rows[:16]:
    print(r["Name"].replace("(anonymous namespace)::","")[:48].ljust(48), r["Calls"].rjust(5), f'{float(r["TotalDurationNs"])/1e6:9.1f} ms', f'{float(r["AverageNs"])/1e6:8.3f} ms avg', r["Percentage"])
PY
