#!/bin/bash
# round-5 batch X: k_emit / k_count_tiles issue all of a lane's loads up front (one wait) instead of instance by instance
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_rasterization.py tests/test_gpu_frame.py tests/test_gpu_graph.py -q -m gpu -x 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -5 | tee gpurun_out/r5x_pytest.txt
D4GS_LAZY_SORT=1 timeout 1500 python -m pytest tests/test_gpu_rasterization.py -q -m gpu -x 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -3 | tee -a gpurun_out/r5x_pytest.txt
D4GS_EXACT_TILES=1 timeout 1500 python -m pytest tests/test_gpu_rasterization.py -q -m gpu -x 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -3 | tee -a gpurun_out/r5x_pytest.txt
{
for rep in 1 2; do
bash scripts/ab_run.sh "--config cfg2" base prebatch
done
bash scripts/ab_run.sh "--config cfg3 --steps 10" base prebatch
bash scripts/ab_run.sh "--config cfg5 --steps 10" base prebatch
bash scripts/ab_run.sh "--config refdefault" base prebatch
bash scripts/ab_run.sh "--config cfg2 --scale-mul 4" base prebatch
} 2>&1 | tee gpurun_out/r5x_ab.txt
