tag=r06s
for c in cfg2 refdefault cfg3 cfg5; do python bench.py --config $c --no-cpu-baseline 2>>gpurun_out/${tag}_bench3.err | tail -1 > gpurun_out/${tag}_bench_$c.json; done
python bench.py --steps 30 --warmup 5 2>>gpurun_out/${tag}_bench3.err | tail -1 > gpurun_out/${tag}_bench_cfg2.json
python bench.py --gpus 1 --steps 20 --warmup 5 2>>gpurun_out/${tag}_bench3.err | tail -1 > gpurun_out/${tag}_bench_cfg2_driver_flags.json
python bench.py 2>>gpurun_out/${tag}_bench3.err | tail -1 > gpurun_out/${tag}_bench_cfg2_default_flags.json
python bench.py --graph --steps 30 --warmup 5 --no-cpu-baseline 2>>gpurun_out/${tag}_bench3.err | tail -1 > gpurun_out/${tag}_bench_graph_cfg2.json
python bench.py --force-dist --graph --steps 30 --warmup 5 --no-cpu-baseline 2>>gpurun_out/${tag}_bench3.err | tail -1 > gpurun_out/${tag}_bench_graph_rccl_world1_cfg2.json
python bench.py --sync-size-check --steps 30 --warmup 5 --no-cpu-baseline 2>>gpurun_out/${tag}_bench3.err | tail -1 > gpurun_out/${tag}_bench_cfg2_sync_size_check.json
for i in 1 2 3 4 5 6; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --sustain 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); h=d['host_step_times']
print('driver flags run $i: %.4f ms' % d['ms_per_step'], 'host median %.3f max %.3f' % (h['host_ms_median'], h['host_ms_max']))"; done | tee gpurun_out/${tag}_bench_driver_flags_repeats.txt
timeout 900 python -m pytest tests/test_gpu_bench_line.py tests/test_gpu_graph.py tests/test_engine_host.py -q -m gpu -p no:cacheprovider 2>&1 | tail -3
