#!/bin/bash
# round-5 batch Y: backward - the staged records' LDS offset pinned in one VGPR; 17 channels: g0 as one 16-byte read
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_rasterization.py tests/test_gpu_exposure.py -q -m gpu -x 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -5 | tee gpurun_out/r5y_pytest.txt
{
for rep in 1 2; do
bash scripts/ab_run.sh "--config cfg2" base prepin
bash scripts/ab_run.sh "--config refdefault" base prepin
done
bash scripts/ab_run.sh "--config cfg3 --steps 10" base prepin
} 2>&1 | tee gpurun_out/r5y_ab.txt
