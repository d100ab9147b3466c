#!/bin/bash
# round-5 batch I: colour-table rows 16 bytes at a time in k_project_fwd / k_project_bwd; parity subset + timing
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_exposure.py tests/test_gpu_rasterization.py tests/test_gpu_frame.py tests/test_gpu_scene_model.py tests/test_gpu_baseline_configs.py -q -m gpu -x 2>&1 | tail -4 | tee gpurun_out/r5i_pytest.txt
{
for c in "--config cfg2" "--config refdefault" "--config cfg2 --channels 16" "--config cfg5 --steps 10"; do
  python bench.py $c --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']
print('[$c]', '%.3f ms' % d['ms_per_step'], {n: round(1e3*t,1) for n,t in list(k.items())[:12]})"
done
} 2>&1 | tee gpurun_out/r5i_ab.txt
