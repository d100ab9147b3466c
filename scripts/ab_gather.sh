#!/bin/bash
# scripts/ab_gather.sh : k_gather (and the frame) on the workloads where it matters
cd "$(dirname "$0")/.."
for c in "--config cfg2" "--config cfg3" "--config cfg5 --steps 10" "--config refdefault" "--config cfg2 --scale-mul 4" "--config cfg2 --channels 16" "--share 8"; do
  python bench.py $c --no-cpu-baseline --no-peaks 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']
print('[$c]', '%.3f ms' % d['ms_per_step'], 'k_gather', round(1e3*k.get('k_gather',0),1), {n: round(1e3*t) for n,t in list(k.items())[:5]})"
done
