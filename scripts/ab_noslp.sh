#!/bin/bash
# scripts/ab_noslp.sh : the composite kernels compiled with -fno-slp-vectorize (no v_pk_fma_f32 / v_pk_mul_f32: a packed fp32 instruction
# issues at 0.41x the plain rate on gfx950 - d4gs_measure_peaks: 102 vs 124 TFLOP/s - so two plain ones are faster than one packed)
cd "$(dirname "$0")/.."
for lib in "" scripts/ablate/libd4gs_noslp_fwd.so scripts/ablate/libd4gs_noslp_bwd.so; do for c in "--config cfg2" "--config refdefault" "--config cfg3"; do
  D4GS_LIB_PATH=$lib python bench.py $c --no-cpu-baseline --no-peaks 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']
print('[$lib $c]', '%.3f ms' % d['ms_per_step'], {n: round(1e3*t,1) for n,t in list(k.items())[:2]})"
done; done
