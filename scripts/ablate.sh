#!/bin/bash
# Cost attribution of k_raster_bwd_q by ablation: builds libd4gs variants with -DD4GS_ABL=n (see raster_bwd.hip) into
# scripts/ablate/ and, on the GPU box, times the kernel for each (results of the ablated kernels are garbage).
#   scripts/ablate.sh build      (build container)        scripts/ablate.sh run > gpurun_out/ablate.txt   (GPU box)
set -e
cd "$(dirname "$0")/.."
mkdir -p scripts/ablate
if [ "$1" = build ]; then
  for n in 1 2 3 4; do
    objs=""
    for f in deblur4dgs_amd/csrc/*.hip; do
      b=$(basename $f .hip)
      if [ $b = raster_bwd ]; then
        hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -w -DD4GS_ABL=$n -c $f -o scripts/ablate/rb_$n.o
        objs="$objs scripts/ablate/rb_$n.o"
      else
        objs="$objs deblur4dgs_amd/csrc/_obj/$b.o"
      fi
    done
    hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/ablate/libd4gs_abl$n.so $objs
  done
  exit 0
fi
for n in 0 1 2 3 4; do
  lib=""; [ $n != 0 ] && lib="$PWD/scripts/ablate/libd4gs_abl$n.so"
  D4GS_LIB_PATH=$lib python bench.py --no-cpu-baseline --steps 20 ${BENCH_ARGS} | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']
print('ABL=$n', 'bwd_q %.1f us' % (1e3*k.get('k_raster_bwd_q',0)), 'frame %.3f ms' % d['ms_per_step'])"
done
