#!/bin/bash
# round-5 batch R: backward - 16-byte slab zero-fill, 8-byte row stores; suite on the committed forward + this
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -8 | tee gpurun_out/r5r_pytest.txt
{
for rep in 1 2; do
bash scripts/ab_run.sh "--config cfg2" base nozero128 head
done
bash scripts/ab_run.sh "--config cfg3 --steps 10" base head
bash scripts/ab_run.sh "--config refdefault" base head
} 2>&1 | tee gpurun_out/r5r_ab.txt
