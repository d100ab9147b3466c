#!/bin/bash
# round-5 batch H: blend kernels with their loads in flight (fwd: 8 sub-samples at a time; bwd: winner map / batched search); full suite
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -6 | tee gpurun_out/r5h_pytest_gpu.txt
{
for c in "--config cfg2" "--config refdefault" "--config cfg3" "--config refdefault720 --steps 10" "--config cfg5 --steps 10" "--config cfg2 --channels 16"; do
  python bench.py $c --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']
print('[$c]', '%.3f ms' % d['ms_per_step'], {n: round(1e3*t,1) for n,t in list(k.items())[:14]})"
done
} 2>&1 | tee gpurun_out/r5h_ab.txt
