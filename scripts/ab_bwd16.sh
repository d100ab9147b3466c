cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_frame.py -x -q -k no_grad 2>&1 | grep -E "Error|assert|launches" | head -8
export AB=$PWD/scripts/ablate/libd4gs_bwd16w4.so
D4GS_LIB_PATH=$AB python -m pytest tests/test_gpu_rasterization.py tests/test_gpu_exposure.py tests/test_gpu_scene_model.py -x -q -k "16 or 17 or segment or sparse or scene" 2>&1 | tail -3
for c in "--config refdefault" "--config cfg2 --channels 16" "--config refdefault --scale-mul 4" "--config refdefault720"; do
 for n in base w4; do
  lib=""; [ $n = w4 ] && lib=$AB
  D4GS_LIB_PATH=$lib python bench.py $c --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']
print('[$c] $n', '%.3f ms' % d['ms_per_step'], {n: round(1e3*t) for n,t in list(k.items())[:4]})"
 done
done
