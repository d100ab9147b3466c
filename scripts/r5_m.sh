#!/bin/bash
# round-5 batch M: packed DPP ladder in wave_sum_store (R = 9 / 10 rows): parity tests, then A/B against the unpacked ladder
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -8 | tee gpurun_out/r5m_pytest.txt
{
for rep in 1 2; do
bash scripts/ab_run.sh "--config cfg2" base nopack
done
bash scripts/ab_run.sh "--config cfg3 --steps 10" base nopack
bash scripts/ab_run.sh "--config cfg5 --steps 10" base nopack
bash scripts/ab_run.sh "--config cfg2 --scale-mul 4" base nopack
bash scripts/ab_run.sh "--config refdefault" base nopack
} 2>&1 | tee gpurun_out/r5m_ab.txt
