"""Summarise the rocprofv3 counter passes of tools/pmc_passes.sh: per kernel, the mean of every counter
over its dispatches (first dispatch of each kernel dropped as warm-up when there are several)."""
import csv
import sys
from collections import defaultdict
from pathlib import Path

root = Path(sys.argv[1])
want = sys.argv[2] if len(sys.argv) > 2 else "kmer_hash"
for pass_dir in sorted(p for p in root.iterdir() if p.is_dir()):
    counters = defaultdict(list)
    durations = defaultdict(list)
    for f in pass_dir.rglob("*counter_collection.csv"):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if want in row.get("Kernel_Name", ""):
                    counters[(row["Counter_Name"], row["Dispatch_Id"])].append(float(row["Counter_Value"]))
    for f in pass_dir.rglob("*kernel_trace.csv"):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if want in row.get("Kernel_Name", ""):
                    durations[row["Kernel_Name"][:60]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
    per_counter = defaultdict(list)
    for (name, _disp), vals in counters.items():
        per_counter[name].append(sum(vals))  # sum over XCDs / instances of one dispatch
    # A kernel may be launched twice per batch -- the mapping kernel takes the sparse kernel's overflow list in a second, tiny
    # launch --: only the dispatches within a factor of ten of the largest count (FULL dispatches), the first of them dropped
    # as warm-up when there are several.
    def full(vals):
        top = max(vals) if vals else 0.0
        big = [v for v in vals if v >= 0.1 * top]
        return big[1:] if len(big) > 1 else big

    print(f"== {pass_dir.name}")
    for name, vals in sorted(per_counter.items()):
        use = full(vals)
        print(f"  {name:28s} mean per dispatch {sum(use) / len(use):.6g}  ({len(vals)} dispatches, {len(use)} full ones averaged)")
    for name, vals in durations.items():
        use = full(vals)
        print(f"  duration_ms {name}: mean {sum(use) / len(use):.4f} ({len(vals)} dispatches, {len(use)} full ones averaged)")
