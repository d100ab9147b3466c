#!/bin/bash
# round-5 batch C: the whole GPU suite on the adopted bases table, the upstream-fixture consumer on MOCK data (full traceback), forward table A/B
mkdir -p gpurun_out
python scripts/mock_upstream_fixture.py gpurun_out/mock_upstream > gpurun_out/r5c_mock.txt 2>&1
D4GS_UPSTREAM_DIR=$PWD/gpurun_out/mock_upstream timeout 900 python -m pytest tests/test_gpu_upstream_fixture.py -q --tb=long 2>&1 | tail -60 > gpurun_out/r5c_pytest_upstream_mock.txt
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/r5c_pytest_gpu.txt
{
bash scripts/ab_run.sh "--config cfg2" base pftab base pftab
bash scripts/ab_run.sh "--config cfg5 --steps 10" base pftab
bash scripts/ab_run.sh "--config refdefault" base pftab
} 2>&1 | tee gpurun_out/r5c_ab.txt
