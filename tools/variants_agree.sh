for L in "" scan noend two; do
  if [ -z "$L" ]; then lib=pyani_plus_amd/_lib/libpyani_hip.so; else lib=pyani_plus_amd/_lib/libpyani_hip_$L.so; fi
  echo "== ${L:-product}: $(PA_AB_LIB=$lib python3 tools/bench_fragani.py 1000 2>/dev/null | grep -E 'sha256|^rep 1' | tr '\n' ' ')"
done
