#!/bin/bash
# The product build and compile-time variants of the mapping kernel on the benchmark's 1 000 x 5 Mb genomes: the bounds, the
# order in which states are taken up, how the hits are ordered and whether the L1 scan runs decide what is evaluated and how,
# never what comes out -- one digest over every result of the 10^6 pairs (tools/bench_fragani.py prints it) must be the same
# for all of them.  Build the variants first:
#   bash tools/build_variant.sh scan -DPA_MAP_L1_ONE_RUN=0;      bash tools/build_variant.sh noend -DPA_MAP_BOUND_IN_ROUND=0
#   bash tools/build_variant.sh two -DPA_MAP_ROUND_ITEMS=128 -DPA_MAP_CENTRE_LANE=32
#   bash tools/build_variant.sh nocount -DPA_MAP_COUNT_SORT=0;   bash tools/build_variant.sh nostray -DPA_MAP_L1_STRAYS=0
for L in "" scan noend two nocount nostray; do
  if [ -z "$L" ]; then lib=pyani_plus_amd/_lib/libpyani_hip.so; else lib=pyani_plus_amd/_lib/libpyani_hip_$L.so; fi
  [ -f "$lib" ] || continue
  echo "== ${L:-product}: $(PA_AB_LIB=$lib python3 tools/bench_fragani.py 1000 2>/dev/null | grep -E 'sha256|^rep 1' | tr '\n' ' ')"
done
