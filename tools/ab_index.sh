#!/bin/bash
# A/B of builds of the library on ONE box, index phase: alternates them, three runs each (one batch of queries: the index dominates).
#   bash tools/ab_index.sh <libA.so> <libB.so> ...
for i in 1 2 3; do
  for L in "$@"; do
    echo "== $L: $(PA_AB_LIB=$L python3 tools/bench_fragani.py 1000 0 interleaved 78 2>/dev/null | grep -E '^rep 1|sha256' | tr '\n' ' ')"
  done
done
