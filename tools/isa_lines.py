"""Static instruction counts per source line of one kernel in a -S -gline-tables-only listing.
    hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S -gline-tables-only X.hip -o X.s
    python tools/isa_lines.py X.s <kernel symbol prefix> <source> [first line] [last line]
Inlined code is attributed to the outermost line of the source file (the `@[ file:line ]` chain of the .loc comment)."""
import collections
import re
import sys

import os
INNER = os.environ.get("ISA_INNER", "1") == "1"
asm, prefix, source = sys.argv[1:4]
first = int(sys.argv[4]) if len(sys.argv) > 4 else 0
last = int(sys.argv[5]) if len(sys.argv) > 5 else 10**9
lines = open(asm).read().splitlines()
name = source.rsplit("/", 1)[-1]
start = [i for i, l in enumerate(lines) if l.startswith(prefix)][0]
end = [i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm")][0]
cur = 0
hist = collections.defaultdict(collections.Counter)
for l in lines[start:end]:
    if ".loc" in l:
        refs = re.findall(re.escape(name) + r":(\d+)", l)
        if refs:
            cur = int(refs[0 if INNER else -1])  # innermost or outermost frame of the inlining chain
        continue
    t = l.strip()
    if not t or t.startswith((".", ";")) or t.endswith(":"):
        continue
    op = t.split()[0]
    kind = "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "mem"
    hist[cur][kind] += 1
src = open(source).read().splitlines()
total = collections.Counter()
for ln in sorted(hist):
    if first <= ln <= last:
        total.update(hist[ln])
        print(f"{ln:5d} {sum(hist[ln].values()):4d} {dict(hist[ln])!s:55s} | {src[ln - 1].strip()[:70] if 0 < ln <= len(src) else ''}")
print("total in range", sum(total.values()), dict(total))
