"""Our own kernels from a rocprofv3 kernel_stats.csv (torch's generators and copies left out), as an aligned table."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
skip = ("at::native", "rocclr", "hiprand", "rocprim", "at::cuda")
print(f"{'kernel':60s} {'calls':>6s} {'avg us':>10s} {'total ms':>10s}")
for r in rows:
    name = r["Name"]
    if any(x in name for x in skip):
        continue
    short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    print(f"{short[:60]:60s} {int(r['Calls']):6d} {float(r['AverageNs']) / 1e3:10.1f} {float(r['TotalDurationNs']) / 1e6:10.2f}")
