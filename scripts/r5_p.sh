#!/bin/bash
# round-5 batch P: forward with the four-steps-per-read loop for one-record colours only; full suite + A/B on the wide shapes
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -8 | tee gpurun_out/r5p_pytest.txt
{
bash scripts/ab_run.sh "--config cfg2" base fwdold
bash scripts/ab_run.sh "--config refdefault" base fwdold
bash scripts/ab_run.sh "--config cfg2 --channels 16" base fwdold
bash scripts/ab_run.sh "--config refdefault720 --steps 10" base fwdold
} 2>&1 | tee gpurun_out/r5p_ab.txt
