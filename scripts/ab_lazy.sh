#!/bin/bash
# scripts/ab_lazy.sh : D4GS_LAZY_SORT against the full sort on the large-footprint / occluded workloads (and cfg2, where it should not pay)
cd "$(dirname "$0")/.."
for c in "--config cfg2 --scale-mul 4" "--config cfg2 --scale-mul 2" "--config cfg5 --steps 10" "--config refdefault --scale-mul 4" "--config cfg3" "--config cfg2"; do for lz in "" "--lazy-sort"; do
  python bench.py $c $lz --no-cpu-baseline --no-peaks 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']
print('[$c $lz]', '%.3f ms' % d['ms_per_step'], {n: round(1e3*t) for n,t in list(k.items())[:10]})"
done; done
