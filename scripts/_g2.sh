scripts/microbench/mfma4x4 2>&1 | tee gpurun_out/r6_mfma4x4.txt
timeout 1200 python -m pytest tests/test_gpu_rasterization.py tests/test_gpu_frame.py tests/test_gpu_scene_model.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -5 | tee gpurun_out/r6b_pytest.txt
{
for rep in 1 2; do
bash scripts/ab_run.sh "--config refdefault" fwd_base fwd_mf1 fwd_mf4w5 base
done
bash scripts/ab_run.sh "--config cfg2 --channels 16" fwd_base fwd_mf1 fwd_mf4w5 base
bash scripts/ab_run.sh "--config refdefault720 --steps 10" fwd_base base
} 2>&1 | tee gpurun_out/r6b_ab_fwd_mfma.txt
