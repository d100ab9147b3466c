#!/bin/bash
# round-5 batch Q: forward - null record in slot 0 (no per-step row-count compare), T update as one FMA
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -8 | tee gpurun_out/r5q_pytest.txt
{
for rep in 1 2; do
bash scripts/ab_run.sh "--config cfg2" base nofma nonull fwdlist4
done
bash scripts/ab_run.sh "--config cfg3 --steps 10" base fwdlist4
bash scripts/ab_run.sh "--config cfg5 --steps 10" base fwdlist4
bash scripts/ab_run.sh "--config refdefault" base fwdlist4
bash scripts/ab_run.sh "--config cfg2 --scale-mul 4" base fwdlist4
} 2>&1 | tee gpurun_out/r5q_ab.txt
