#!/bin/bash
# round-5 batch O: composite forward - four list bytes per LDS read (SDWA byte x 16), zero-filled lists instead of the per-step select,
# one saturation compare through explicit lane masks; A/B against the previous forward (fwdold) and the one-byte loop (nolist4)
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_rasterization.py tests/test_gpu_exposure.py tests/test_gpu_known_answers.py tests/test_gpu_frame.py tests/test_gpu_graph.py -q -m gpu -x 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -5 | tee gpurun_out/r5o_pytest.txt
{
for rep in 1 2; do
bash scripts/ab_run.sh "--config cfg2" base fwdold nolist4
done
bash scripts/ab_run.sh "--config cfg3 --steps 10" base fwdold
bash scripts/ab_run.sh "--config cfg5 --steps 10" base fwdold
bash scripts/ab_run.sh "--config refdefault" base fwdold
bash scripts/ab_run.sh "--config cfg2 --channels 16" base fwdold
bash scripts/ab_run.sh "--config cfg2 --scale-mul 4" base fwdold
} 2>&1 | tee gpurun_out/r5o_ab.txt
