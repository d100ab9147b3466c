cd $GRAFT_REPO_ROOT
for sh in 8 4; do
  for n in base seg0 seg_u512 seg_u128 seg_m4; do
    lib=""; envs="X=1"
    case $n in base) ;; seg0) envs="D4GS_SEG=0";; *) lib="$PWD/scripts/ablate/libd4gs_$n.so";; esac
    env $envs D4GS_LIB_PATH=$lib python bench.py --no-cpu-baseline --share $sh --steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']
print('[share $sh] $n', {n: round(1e3*t,1) for n,t in list(k.items())[:6]}, 'sum %.0f' % (1e3*sum(k.values())), 'frame %.3f ms' % d['ms_per_step'])"
  done
done
