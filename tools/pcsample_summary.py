"""Aggregate a rocprofv3 PC-sampling run (csv output) by source line and by instruction.

    python tools/pcsample_summary.py <output dir of rocprofv3> [source file filter] > summary.txt
The library has to be built with -gline-tables-only for the source lines to be there (Instruction_Comment column)."""
import collections
import csv
import glob
import re
import sys

root = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else "fragani.hip"
files = glob.glob(root + "/**/*pc_sampling*.csv", recursive=True)
by_line = collections.Counter()
by_inst = collections.Counter()
total = 0
cols = None
for f in files:
    with open(f, newline="") as h:
        rd = csv.DictReader(h)
        cols = rd.fieldnames
        for row in rd:
            total += 1
            comment = row.get("Instruction_Comment") or ""
            inst = (row.get("Instruction") or "").split()[0] if row.get("Instruction") else "?"
            m = re.findall(re.escape(flt) + r":(\d+)", comment)
            key = int(m[-1]) if m else -1  # outermost frame of the inlining chain
            by_line[key] += 1
            by_inst[(key, inst)] += 1
print("files", files)
print("columns", cols)
print("samples", total)
for line, n in by_line.most_common(120):
    tops = [(i, c) for (l, i), c in by_inst.items() if l == line]
    tops.sort(key=lambda x: -x[1])
    print(f"{line:6d} {n:9d} {100.0 * n / max(total, 1):6.2f} %  " + " ".join(f"{i}:{c}" for i, c in tops[:6]))
