#!/bin/bash
# round-5 batch Z: k_project_bwd reads its epilogue's pointers afresh from the kernarg segment (fewer SGPR spills)
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_exposure.py tests/test_gpu_poses.py tests/test_gpu_scene_model.py tests/test_gpu_frame.py -q -m gpu -x 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -5 | tee gpurun_out/r5z2_pytest.txt
{
for rep in 1 2; do
bash scripts/ab_run.sh "--config cfg2" base karg1 prekarg
done
bash scripts/ab_run.sh "--config cfg5 --steps 10" base karg1 prekarg
bash scripts/ab_run.sh "--config refdefault" base karg1 prekarg
bash scripts/ab_run.sh "--config cfg3 --steps 10" base karg1 prekarg
} 2>&1 | tee gpurun_out/r5z2_ab.txt
