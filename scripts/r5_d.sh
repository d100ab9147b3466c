#!/bin/bash
# round-5 batch D: exact tiles (tests, then A/B over the tracked configs), upstream consumer on mock data, PB_PAIR A/B, forward table rule
mkdir -p gpurun_out
python scripts/mock_upstream_fixture.py gpurun_out/mock_upstream > gpurun_out/r5d_mock.txt 2>&1
D4GS_UPSTREAM_DIR=$PWD/gpurun_out/mock_upstream timeout 900 python -m pytest tests/test_gpu_upstream_fixture.py -q --tb=short 2>&1 | tail -30 > gpurun_out/r5d_pytest_upstream_mock.txt
timeout 1500 python -m pytest tests/test_gpu_rasterization.py tests/test_gpu_frame.py -x -q -m gpu -k "exact_tiles or exact_cull or lazy" --tb=short 2>&1 | tail -40 | tee gpurun_out/r5d_pytest_xt.txt
{
for c in "--config cfg2" "--config cfg3" "--config cfg5 --steps 10" "--config refdefault" "--config cfg2 --scale-mul 4" "--config refdefault720 --steps 10"; do
  for xt in 0 1 auto; do
    D4GS_EXACT_TILES=$xt python bench.py $c --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']
print('[$c] EXACT_TILES=$xt', '%.3f ms' % d['ms_per_step'], d.get('n_isect_per_step'), {n: round(1e3*t) for n,t in list(k.items())[:9]})"
  done
done
bash scripts/ab_run.sh "--config cfg2" base pbpair base pbpair
} 2>&1 | tee gpurun_out/r5d_ab.txt
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -12 | tee gpurun_out/r5d_pytest_gpu.txt
D4GS_EXACT_TILES=1 timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -25 | tee gpurun_out/r5d_pytest_gpu_xt_forced.txt
