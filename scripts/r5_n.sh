#!/bin/bash
# round-5 batch N: composite backward - colour record read in front of the early-out, va - bsum as one running value
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_rasterization.py tests/test_gpu_exposure.py tests/test_gpu_known_answers.py tests/test_gpu_frame.py -q -m gpu -x 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -5 | tee gpurun_out/r5n_pytest.txt
{
for rep in 1 2; do
bash scripts/ab_run.sh "--config cfg2" base noearly nopack
done
bash scripts/ab_run.sh "--config cfg3 --steps 10" base noearly
bash scripts/ab_run.sh "--config refdefault" base noearly
} 2>&1 | tee gpurun_out/r5n_ab.txt
