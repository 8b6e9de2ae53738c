"""Print a compact per-kernel table from a rocprofv3 *_kernel_stats.csv (our kernels only)."""
import csv
import sys

for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "at::native" in n:
        continue
    short = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    print(f"{short[:40]:40s} calls {r['Calls']:>6s}  total_ms {int(r['TotalDurationNs']) / 1e6:10.2f}  avg_us {float(r['AverageNs']) / 1e3:11.1f}  {float(r['Percentage']):6.2f}%")
