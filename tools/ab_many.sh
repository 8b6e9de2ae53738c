#!/bin/bash
# Several builds of the library alternated on ONE box (boxes differ by a few per cent), two rounds; the phase timers of the
# second repetition of tools/bench_fragani.py 1000.  The builds are pyani_plus_amd/_lib/libpyani_hip_<name>.so, as
# tools/build_variant.sh makes them (copy the product library to such a name to have it in the comparison):
#   bash tools/ab_many.sh <name> [<name> ...]
N=1000
for i in 1 2; do
  for L in "$@"; do
    echo "== $L $(PA_AB_LIB=pyani_plus_amd/_lib/libpyani_hip_$L.so python3 tools/bench_fragani.py $N 2>/dev/null | grep '^rep 1' | sed "s/.*pairs\/s//")"
  done
done
