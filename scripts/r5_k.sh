#!/bin/bash
# round-5 batch K: 24-float gradient rows for 17-channel renders (k_raster_bwd_q<16> b128 row stores, k_gather<16> b128 LDS reads)
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -5 | tee gpurun_out/r5k_pytest.txt
D4GS_BWD_ROWS=sparse timeout 1500 python -m pytest tests/test_gpu_rasterization.py tests/test_gpu_frame.py tests/test_gpu_scene_model.py -q -m gpu -x 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -3 | tee -a gpurun_out/r5k_pytest.txt
{
for c in "--config refdefault" "--config cfg2 --channels 16" "--config refdefault720 --steps 10" "--config cfg2" "--config refdefault --scale-mul 4"; do
  python bench.py $c --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']
print('[$c]', '%.3f ms' % d['ms_per_step'], {n: round(1e3*t,1) for n,t in list(k.items())[:8]})"
done
} 2>&1 | tee gpurun_out/r5k_ab.txt
