#!/bin/bash
# round-5 batch S: backward - the staging / write-out wave per-block last contributors in the hit test
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_rasterization.py tests/test_gpu_exposure.py tests/test_gpu_known_answers.py tests/test_gpu_frame.py -q -m gpu -x 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -5 | tee gpurun_out/r5t_pytest.txt
{
for rep in 1 2; do
bash scripts/ab_run.sh "--config cfg2" base noblk
done
bash scripts/ab_run.sh "--config cfg3 --steps 10" base noblk
bash scripts/ab_run.sh "--config refdefault" base noblk
bash scripts/ab_run.sh "--config cfg2 --scale-mul 4" base noblk
} 2>&1 | tee gpurun_out/r5t_ab.txt
