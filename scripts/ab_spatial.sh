cd $GRAFT_REPO_ROOT
for c in "--config cfg2" "--config cfg5 --steps 10" "--config cfg3" "--config refdefault" "--config cfg2 --scale-mul 4"; do for sp in "" "--spatial-order view"; do
  python bench.py $c $sp --no-cpu-baseline --no-peaks 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']
print('[$c $sp]', '%.3f ms' % d['ms_per_step'], {n: round(1e3*t) for n,t in list(k.items())[:9]})"
done; done
