set -x
run() { # label, env..., -- bench args
  lab=$1; shift
  envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  env $envs python bench.py --no-cpu-baseline --sustain 0 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']
print('[$lab | $*]', {n: round(1e3*t,1) for n,t in list(k.items())[:6]}, 'frame %.3f ms' % d['ms_per_step'])"
}
{
for rep in 1 2; do
run q8 -- --config cfg2
run bwd7 D4GS_BWD_Q=7 -- --config cfg2
run bwd6 D4GS_BWD_Q=6 -- --config cfg2
run bwd5 D4GS_BWD_Q=5 -- --config cfg2
run fwd7 D4GS_FWD_Q=7 -- --config cfg2
run fwd6 D4GS_FWD_Q=6 -- --config cfg2
run fwd5 D4GS_FWD_Q=5 -- --config cfg2
run both6 D4GS_FWD_Q=6 D4GS_BWD_Q=6 -- --config cfg2
done
run q8 -- --config cfg3
run both6 D4GS_FWD_Q=6 D4GS_BWD_Q=6 -- --config cfg3
run both7 D4GS_FWD_Q=7 D4GS_BWD_Q=7 -- --config cfg3
run q8 -- --config cfg2 --share 2
run both6 D4GS_FWD_Q=6 D4GS_BWD_Q=6 -- --config cfg2 --share 2
run both5 D4GS_FWD_Q=5 D4GS_BWD_Q=5 -- --config cfg2 --share 2
run both7 D4GS_FWD_Q=7 D4GS_BWD_Q=7 -- --config cfg2 --share 2
} 2>&1 | grep -v "^+" | tee gpurun_out/r6d_ab_wgs_per_cu.txt
timeout 2400 python -m pytest tests/test_gpu_refdefault_fullsize.py -q -m gpu -p no:cacheprovider -s 2>&1 | tail -30 | tee gpurun_out/r6d_pytest_refdefault.txt
