tag=r06s
python bench.py --steps 30 --warmup 5 2>>gpurun_out/${tag}_bench2.err | tail -1 > gpurun_out/${tag}_bench_cfg2.json
python bench.py --gpus 1 --steps 20 --warmup 5 2>>gpurun_out/${tag}_bench2.err | tail -1 > gpurun_out/${tag}_bench_cfg2_driver_flags.json
python bench.py 2>>gpurun_out/${tag}_bench2.err | tail -1 > gpurun_out/${tag}_bench_cfg2_default_flags.json
timeout 600 python -m pytest tests/test_gpu_bench_line.py -q -m gpu -p no:cacheprovider 2>&1 | tail -3
