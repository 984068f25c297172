#!/bin/bash
# Builds the library from the current sources into radiative-transfer_amd/csrc/variants/<name>.so (for same-session
# A/B timing with RT_AMD_LIB=..., tools/ab.sh): bash tools/build_variant.sh <name> [extra hipcc flags]
set -e
NAME=$1; shift
cd "$(dirname "$0")/../radiative-transfer_amd/csrc"
mkdir -p variants build_$NAME
for f in *.hip; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC "$@" -c $f -o build_$NAME/$f.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC build_$NAME/*.o -o variants/$NAME.so
rm -rf build_$NAME
echo built variants/$NAME.so
