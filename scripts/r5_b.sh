#!/bin/bash
# round-5 batch B: new tests (canonical pose, render_view values, upstream-fixture consumer on MOCK data, fused counts, single-block finish),
# A/B of the bases table in k_project_bwd and of the one-launch finish
mkdir -p gpurun_out
python scripts/mock_upstream_fixture.py gpurun_out/mock_upstream > gpurun_out/r5b_mock.txt 2>&1
D4GS_UPSTREAM_DIR=$PWD/gpurun_out/mock_upstream timeout 900 python -m pytest tests/test_gpu_upstream_fixture.py -x -q 2>&1 | tail -15 | tee gpurun_out/r5b_pytest_upstream_mock.txt
timeout 1800 python -m pytest tests/test_gpu_scene_model.py tests/test_gpu_frame.py tests/test_gpu_graph.py tests/test_gpu_exposure.py tests/test_gpu_poses.py tests/test_gpu_parallel.py tests/test_gpu_bench_line.py tests/test_c_abi_demo.py tests/test_gpu_baseline_configs.py -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/r5b_pytest.txt
{
bash scripts/ab_run.sh "--config cfg2" base btab base btab
D4GS_FINISH_ONE=0 bash scripts/ab_run.sh "--config cfg2" base
bash scripts/ab_run.sh "--config cfg5 --steps 10" base btab
bash scripts/ab_run.sh "--config refdefault" base btab
bash scripts/ab_run.sh "--config cfg3" base btab
} 2>&1 | tee gpurun_out/r5b_ab.txt
