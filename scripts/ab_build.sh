#!/bin/bash
# scripts/ab_build.sh <name> <sed-expr applied to raster_bwd.hip> : builds scripts/ablate/libd4gs_<name>.so (A/B timing only)
set -e
cd "$(dirname "$0")/.."
mkdir -p scripts/ablate
name=$1; shift
src=${AB_SRC:-raster_bwd}
sed "$@" deblur4dgs_amd/csrc/$src.hip > deblur4dgs_amd/csrc/_ab_$name.hip
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -w -c deblur4dgs_amd/csrc/_ab_$name.hip -o scripts/ablate/ab_$name.o
rm deblur4dgs_amd/csrc/_ab_$name.hip
objs=""
for f in deblur4dgs_amd/csrc/*.hip; do b=$(basename $f .hip); [ $b = $src ] && objs="$objs scripts/ablate/ab_$name.o" || objs="$objs deblur4dgs_amd/csrc/_obj/$b.o"; done
hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/ablate/libd4gs_$name.so $objs
