#!/bin/bash
# A variant of the product library for an A/B on one box (tools/ab_fragani.sh): fragani.hip compiled with extra -D flags,
# linked with the product's other objects.
#   bash tools/build_variant.sh <name> -DPA_MAP_ROUND_ITEMS=128 -DPA_MAP_CENTRE_LANE=32   ->  pyani_plus_amd/_lib/libpyani_hip_<name>.so
# (the variants are git-ignored like every built library; remove them when the comparison is done)
set -e
cd "$(dirname "$0")/../pyani_plus_amd/csrc"
name=$1; shift
make -s ../_lib/libpyani_hip.so
mkdir -p ../_build/variant
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden "$@" -c fragani.hip -o ../_build/variant/fragani_$name.o
objs=$(ls ../_build/*.o | grep -v '/fragani.o')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../_lib/libpyani_hip_$name.so ../_build/variant/fragani_$name.o $objs -Wl,-Bsymbolic -lz -lpthread -ldl
echo ../_lib/libpyani_hip_$name.so
