#!/bin/bash
# One GPU round: gpu tests, smoke, bench (JSON), rocprofv3 kernel-trace stats of the same bench command.
set -x
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/pytest_gpu.txt
python __graft_entry__.py smoke 2>&1 | tail -2 | tee gpurun_out/smoke.txt
python bench.py --steps 30 --warmup 5 2>gpurun_out/bench.err | tail -1 | tee gpurun_out/bench.json
export TMPDIR=/tmp
R=$PWD
cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o r1 -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile > $R/gpurun_out/prof_bench.json 2>$R/gpurun_out/prof.err
cd $R
ls -R gpurun_out/prof | head -20
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -25 "$f" | tee gpurun_out/kernel_stats_head.csv
