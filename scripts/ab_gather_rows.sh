#!/bin/bash
# scripts/ab_gather_rows.sh : LDS stage rows of k_gather's sparse instantiation (A/B libraries: python -m deblur4dgs_amd.build --ab gs96 raster_bwd.hip -DD4GS_GATHER_ROWS_SPARSE=96)
cd "$(dirname "$0")/.."
for lib in "" scripts/ablate/libd4gs_gs128.so scripts/ablate/libd4gs_gs96.so; do for c in "--config cfg2 --scale-mul 2" "--config cfg5 --steps 10" "--config refdefault" "--config refdefault --scale-mul 4" "--config cfg2 --scale-mul 4"; do
  D4GS_LIB_PATH=$lib python bench.py $c --no-cpu-baseline --no-peaks 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']
print('[$lib $c]', '%.3f ms' % d['ms_per_step'], 'k_gather', round(1e3*k.get('k_gather',0),1))"
done; done
