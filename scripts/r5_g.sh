#!/bin/bash
# round-5 batch G: k_project_bwd with the view-matrix gradient through the per-pass wave reduction (PB_VIEWRED), and K <= 8 on the matrix pipe
mkdir -p gpurun_out
V=$PWD/scripts/ablate/libd4gs_viewred.so; V4=$PWD/scripts/ablate/libd4gs_viewred4.so
for lib in $V $V4; do
  for m in 0 1; do
    echo "== parity: $(basename $lib) MFMA_ALL=$m"
    D4GS_LIB_PATH=$lib D4GS_PB_MFMA_ALL=$m timeout 900 python -m pytest tests/test_gpu_exposure.py tests/test_gpu_poses.py tests/test_gpu_frame.py tests/test_gpu_baseline_configs.py -x -q -m gpu 2>&1 | tail -3
  done
done 2>&1 | tee gpurun_out/r5g_parity.txt
{
for c in "--config cfg2" "--config cfg5 --steps 10" "--config refdefault" "--share 8" "--share 4"; do
  for v in ":0" "$V:0" "$V:1" "$V4:0" "$V4:1"; do
    lib=${v%%:*}; m=${v#*:}
    D4GS_LIB_PATH=$lib D4GS_PB_MFMA_ALL=$m python bench.py $c --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']
print('[$c] lib=$(basename "$lib") MFMA_ALL=$m', '%.3f ms' % d['ms_per_step'], {n: round(1e3*t,1) for n,t in k.items() if n in ('k_project_bwd','k_project_fwd','k_reduce_partials','k_finish')})"
  done
done
} 2>&1 | tee gpurun_out/r5g_ab.txt
timeout 900 python -m pytest tests/test_gpu_frame.py tests/test_gpu_graph.py -q -m gpu -x 2>&1 | tail -4 | tee gpurun_out/r5g_pytest.txt
