run() { lab=$1; shift; envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  env $envs python bench.py --no-cpu-baseline --sustain 0 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']
print('[$lab | $*]', {n: round(1e3*k[n],1) for n in ('k_project_fwd','k_count_tiles','k_bases_table','k_project_bwd') if n in k}, 'frame %.4f ms' % d['ms_per_step'])"; }
{
for rep in 1 2; do
run sg1 D4GS_PROJ_SG=1 -- --config cfg2
run sg2 D4GS_PROJ_SG=2 -- --config cfg2
run sg4 D4GS_PROJ_SG=4 -- --config cfg2
run auto -- --config cfg2
done
for c in refdefault cfg3 cfg5 cfg1; do
run sg1 D4GS_PROJ_SG=1 -- --config $c
run sg2 D4GS_PROJ_SG=2 -- --config $c
run auto -- --config $c
done
run sg4 D4GS_PROJ_SG=4 -- --config refdefault
run sg1 D4GS_PROJ_SG=1 -- --config cfg2 --share 8
run auto -- --config cfg2 --share 8
run sg1 D4GS_PROJ_SG=1 -- --config cfg2 --channels 16
run sg2 D4GS_PROJ_SG=2 -- --config cfg2 --channels 16
} 2>&1 | tee gpurun_out/r6p_ab_proj_sg.txt
timeout 900 python -m pytest tests/test_gpu_rasterization.py tests/test_gpu_frame.py tests/test_gpu_exposure.py tests/test_gpu_poses.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -3
D4GS_PROJ_SG=3 timeout 900 python -m pytest tests/test_gpu_rasterization.py tests/test_gpu_frame.py tests/test_gpu_exposure.py tests/test_gpu_fullsize_properties.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -3
