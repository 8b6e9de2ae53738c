#!/bin/bash
# several builds alternated on one box, two rounds
N=1000
for i in 1 2; do
  for L in "$@"; do
    echo "== $L $(PA_AB_LIB=pyani_plus_amd/_lib/libpyani_hip_$L.so python3 tools/bench_fragani.py $N 2>/dev/null | grep '^rep 1' | sed "s/.*pairs\/s//")"
  done
done
