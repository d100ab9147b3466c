#!/bin/bash
# round-5 batch W: 17-channel backward - parked hit slots in a 64-bit scalar
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_rasterization.py tests/test_gpu_exposure.py tests/test_gpu_scene_model.py tests/test_gpu_frame.py -q -m gpu -x 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -5 | tee gpurun_out/r5w_pytest.txt
{
for rep in 1 2; do
bash scripts/ab_run.sh "--config refdefault" base prehp
done
bash scripts/ab_run.sh "--config cfg2 --channels 16" base prehp
bash scripts/ab_run.sh "--config refdefault720 --steps 10" base prehp
} 2>&1 | tee gpurun_out/r5w_ab.txt
