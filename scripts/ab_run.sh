#!/bin/bash
# scripts/ab_run.sh "<bench args>" name1 name2 ... : times the raster kernels with scripts/ablate/libd4gs_<name>.so ("base" = product lib)
cd "$(dirname "$0")/.."
args=$1; shift
for n in "$@"; do
  lib=""; envs=""
  case $n in base) ;; v1) envs="D4GS_BWD_QUADS_V1=1";; *) lib="$PWD/scripts/ablate/libd4gs_$n.so";; esac
  env $envs D4GS_LIB_PATH=$lib python bench.py --no-cpu-baseline --sustain 0 $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']
print('[$args] $n', {n: round(1e3*t,1) for n,t in list(k.items())[:8]}, 'frame %.3f ms' % d['ms_per_step'])"
done
