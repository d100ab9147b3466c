#!/bin/bash
# round-5 batch E: exact tiles after the 2x2 / early-store changes (product = 5 waves, xt6 = 6 waves + 16 B scratch), full suites
mkdir -p gpurun_out
python scripts/mock_upstream_fixture.py gpurun_out/mock_upstream > gpurun_out/r5e_mock.txt 2>&1
D4GS_UPSTREAM_DIR=$PWD/gpurun_out/mock_upstream timeout 900 python -m pytest tests/test_gpu_upstream_fixture.py -q --tb=short 2>&1 | tail -12 > gpurun_out/r5e_pytest_upstream_mock.txt
{
for c in "--config cfg3" "--config cfg5 --steps 10" "--config cfg2 --scale-mul 4" "--config refdefault720 --steps 10" "--config cfg2 --scale-mul 2"; do
  for v in "0:" "1:" "1:$PWD/scripts/ablate/libd4gs_xt6.so" "auto:"; do
    xt=${v%%:*}; lib=${v#*:}
    D4GS_LIB_PATH=$lib D4GS_EXACT_TILES=$xt python bench.py $c --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']
print('[$c] EXACT_TILES=$xt lib=${lib##*/}', '%.3f ms' % d['ms_per_step'], d.get('n_isect_per_step'), {n: round(1e3*t) for n,t in list(k.items())[:9]})"
  done
done
bash scripts/ab_run.sh "--config cfg2" base base
bash scripts/ab_run.sh "--config refdefault" base
} 2>&1 | tee gpurun_out/r5e_ab.txt
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -12 | tee gpurun_out/r5e_pytest_gpu.txt
