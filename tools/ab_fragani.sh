#!/bin/bash
# A/B of two builds of the library on ONE box (boxes differ by a few per cent): alternates them, three runs each.
#   bash tools/ab_fragani.sh <libA.so> <libB.so> [n_genomes=1000]
A=$1; B=$2; N=${3:-1000}
for i in 1 2 3; do
  for L in "$A" "$B"; do
    echo "== $L"
    PA_AB_LIB=$L python3 tools/bench_fragani.py $N 2>/dev/null | grep "^rep 1"
  done
done
