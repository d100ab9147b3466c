#!/bin/bash
# round-5 first contact: the whole GPU suite, the bench line, and the N>1 bench flow at world size 1 over RCCL (--force-dist)
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/r5a_pytest_gpu.txt
timeout 600 python bench.py --steps 30 --warmup 5 2>gpurun_out/r5a_bench.err | tail -1 > gpurun_out/r5a_bench.json
timeout 300 python bench.py --steps 30 --warmup 5 --force-dist --graph --no-cpu-baseline 2>gpurun_out/r5a_bench_fd.err | tail -1 > gpurun_out/r5a_bench_forcedist_graph.json
python - <<'PY'
import json
for f in ("r5a_bench.json","r5a_bench_forcedist_graph.json"):
    try:
        d=json.load(open("gpurun_out/"+f)); print(f, d["value"], d["ms_per_step"], d["config"].get("launch"), d.get("kernels_ms_per_step"))
    except Exception as e: print(f, "FAILED", e)
PY
