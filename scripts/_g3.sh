timeout 1200 python -m pytest tests/test_gpu_rasterization.py tests/test_gpu_frame.py tests/test_gpu_scene_model.py tests/test_gpu_exposure.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -5 | tee gpurun_out/r6c_pytest.txt
{
for rep in 1 2; do
bash scripts/ab_run.sh "--config refdefault" bwd_nodm base
done
bash scripts/ab_run.sh "--config cfg2 --channels 16" bwd_nodm base
bash scripts/ab_run.sh "--config refdefault720 --steps 10" bwd_nodm base
} 2>&1 | tee gpurun_out/r6c_ab_bwd_dmfma.txt
