#!/bin/bash
# scripts/ab_blend_bwd.sh : the exposure blend's adjoint in the composite backward's prologue (default) against the k_blend_bwd launch (D4GS_FUSE_BLEND_BWD=0)
cd "$(dirname "$0")/.."
for c in "--config cfg2" "--config refdefault" "--config cfg3" "--config cfg5 --steps 10" "--share 8"; do for t in 0 1; do
  D4GS_FUSE_BLEND_BWD=$t python bench.py $c --no-cpu-baseline --no-peaks 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']
print('[$c fused=$t]', '%.4f ms' % d['ms_per_step'], {n: round(1e3*k.get(n,0),1) for n in ('k_raster_bwd_q','k_blend_bwd','k_blend_fwd')})"
done; done
