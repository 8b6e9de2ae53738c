#!/bin/bash
# FETCH_SIZE calibrated for the seeding kernel's access pattern: runs tools/fetch_calib (hipcc --offload-arch=gfx950 -O3 -o
# tools/fetch_calib tools/fetch_calib.hip, built in the build container) under rocprofv3 --pmc FETCH_SIZE and divides what the
# counter says by what the host knows was read.   bash tools/fetch_calib.sh <tag>  ->  gpurun_out/<tag>_fetch_calibration.txt
TAG=${1:-r06}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/${TAG}_fetch_calibration.txt
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
D=/tmp/fcal_$$
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $D -- $ROOT/tools/fetch_calib > $D.log 2>&1
python3 - $D $D.log > $OUT <<'PY'
import csv, sys
from collections import defaultdict
from pathlib import Path
root, log = Path(sys.argv[1]), Path(sys.argv[2])
known = {}
for line in log.read_text().splitlines():
    if line.startswith("calib_"):
        name, rest = line.split(" bytes_asked ")
        parts = rest.split()
        known[name] = (float(parts[0]), float(parts[2]))
per = defaultdict(lambda: defaultdict(float))
dur = defaultdict(list)
for f in root.rglob("*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] == "FETCH_SIZE":
            per[row["Kernel_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
for f in root.rglob("*kernel_trace.csv"):
    for row in csv.DictReader(open(f)):
        dur[row["Kernel_Name"]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
print("FETCH_SIZE (rocprofv3 --pmc, KiB per dispatch, summed over the XCDs; first dispatch of each kernel dropped) against what tools/fetch_calib read:")
for name, (asked, lines) in known.items():
    kern = [k for k in per if name in k]
    if not kern:
        print(f"{name}: no counter rows"); continue
    vals = list(per[kern[0]].values())[1:] or list(per[kern[0]].values())
    counted = sum(vals) / len(vals) * 1024.0
    ms = dur[kern[0]][1:] or dur[kern[0]]
    print(f"{name}: bytes asked {asked:.4g}, 64-byte lines touched x 64 = {lines:.4g}, FETCH_SIZE counted {counted:.4g} bytes -> "
          f"counted / asked = {counted / asked:.3f}, counted / line bytes = {counted / lines:.3f}   ({sum(ms) / len(ms):.3f} ms per dispatch, {lines / (sum(ms) / len(ms) * 1e-3) / 1e9:.0f} GB/s of lines)")
PY
cat $OUT
rm -rf $D $D.log
