#!/bin/bash
set -e
for fb in 64 128 256; do
  sed -i "s/constexpr int FB = [0-9]*;  \/\/ splats per batch (forward)/constexpr int FB = $fb;  \/\/ splats per batch (forward)/" deblur4dgs_amd/csrc/raster_fwd.hip
  python -m deblur4dgs_amd.build > /dev/null 2>&1
  echo "FB=$fb $(python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep metric | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), round(d['kernels_ms_per_step']['k_raster_fwd_q'],3), round(d['kernels_ms_per_step']['k_raster_bwd_q'],3))")"
done
