#!/bin/bash
# usage: exp_nb.sh  -> rebuild raster_bwd with different NB and report kernel times (run on the GPU box; hipcc is there)
set -e
for nb in 32 64; do
  sed -i "0,/constexpr int NB = [0-9]*;  \/\/ splats per batch/s//constexpr int NB = $nb;  \/\/ splats per batch/" deblur4dgs_amd/csrc/raster_bwd.hip
  python -m deblur4dgs_amd.build > /dev/null 2>&1
  echo "NB=$nb $(python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep metric | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), round(d['kernels_ms_per_step']['k_raster_bwd_q'],3))")"
done
