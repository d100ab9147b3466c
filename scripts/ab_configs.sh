#!/bin/bash
# scripts/ab_configs.sh : frame time + top kernels of the product build on the configs the round tracks
cd "$(dirname "$0")/.."
for c in "--config cfg2" "--config cfg2 --channels 16" "--config cfg3" "--config cfg5 --steps 10" "--config refdefault" "--config cfg2 --scale-mul 4" "--config refdefault --scale-mul 4"; do
  python bench.py $c --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']
print('[$c]', '%.3f ms' % d['ms_per_step'], '%.1f M/s' % (d['value']/1e6), {n: round(1e3*t) for n,t in list(k.items())[:7]})"
done
