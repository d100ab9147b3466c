set -x
D4GS_LIB_PATH=$PWD/scripts/ablate/libd4gs_trace.so python scripts/trace_wgs.py --share 1 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6e_trace_bwd_cfg2.txt
D4GS_LIB_PATH=$PWD/scripts/ablate/libd4gs_trace.so python scripts/trace_wgs.py --share 2 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6e_trace_bwd_cfg2_share2.txt
