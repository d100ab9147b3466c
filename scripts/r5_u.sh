#!/bin/bash
# round-5 batch U: 17-channel composite backward - its 6 / 7 VALU rows through the two-register packed ladder
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -8 | tee gpurun_out/r5u_pytest.txt
{
for rep in 1 2; do
bash scripts/ab_run.sh "--config refdefault" base nopack2
done
bash scripts/ab_run.sh "--config cfg2 --channels 16" base nopack2
bash scripts/ab_run.sh "--config refdefault720 --steps 10" base nopack2
bash scripts/ab_run.sh "--config cfg2" base
} 2>&1 | tee gpurun_out/r5u_ab.txt
