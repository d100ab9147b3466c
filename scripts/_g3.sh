run() { lab=$1; shift; envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  env $envs python bench.py --no-cpu-baseline --sustain 0 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']
print('[$lab | $*]', {n: round(1e3*k[n],1) for n in ('k_emit','k_count_tiles','k_scan_single','k_tile_sort','k_project_fwd') if n in k}, 'frame %.4f ms' % d['ms_per_step'], d['config'].get('lazy_sort'))"; }
{
for rep in 1 2; do
run pt4 D4GS_CHUNK_PT=4 -- --config cfg3
run auto -- --config cfg3
done
run pt4 D4GS_CHUNK_PT=4 -- --config cfg2 --scale-mul 4
run auto -- --config cfg2 --scale-mul 4
run pt4 D4GS_CHUNK_PT=4 -- --config cfg2 --scale-mul 1.5
run auto -- --config cfg2 --scale-mul 1.5
run pt4 D4GS_CHUNK_PT=4 -- --config cfg2 --scale-mul 0.7
run auto -- --config cfg2 --scale-mul 0.7
} 2>&1 | tee gpurun_out/r6n_ab_chunk_pt.txt
