#!/bin/bash
# round-5 batch V: 17-channel backward - the MFMA flush reads its B operand with plain loads (columns 8..15 repeat 0..7)
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_rasterization.py tests/test_gpu_exposure.py tests/test_gpu_scene_model.py tests/test_gpu_frame.py -q -m gpu -x 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -5 | tee gpurun_out/r5v_pytest.txt
{
for rep in 1 2; do
bash scripts/ab_run.sh "--config refdefault" base premfl
done
bash scripts/ab_run.sh "--config cfg2 --channels 16" base premfl
bash scripts/ab_run.sh "--config refdefault720 --steps 10" base premfl
} 2>&1 | tee gpurun_out/r5v_ab.txt
