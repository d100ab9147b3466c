#!/bin/bash
# One GPU round: gpu tests, smoke, bench (JSON), rocprofv3 kernel-trace stats of the same bench command.
# usage: scripts/gpu_round.sh [tag] [pytest-args...]   (outputs under gpurun_out/<tag>_*)
tag=${1:-r}; shift
set -x
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu "$@" 2>&1 | tail -15 | tee gpurun_out/${tag}_pytest_gpu.txt
python __graft_entry__.py smoke 2>&1 | tail -2 | tee gpurun_out/${tag}_smoke.txt
python bench.py --steps 30 --warmup 5 2>gpurun_out/${tag}_bench.err | tail -1 | tee gpurun_out/${tag}_bench.json
export TMPDIR=/tmp
R=$PWD
for cfg in cfg2 cfg5 refdefault; do
  cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${tag}_$cfg -o r1 -- python $R/bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline --no-profile --no-peaks > $R/gpurun_out/${tag}_prof_bench_$cfg.json 2>$R/gpurun_out/${tag}_prof_$cfg.err
  cd $R

  db=$(find gpurun_out/prof_${tag}_$cfg -name "*.db" | head -1)
  [ -n "$db" ] && python scripts/rocpd_summary.py $db > gpurun_out/${tag}_kernel_stats_$cfg.csv && head -12 gpurun_out/${tag}_kernel_stats_$cfg.csv
  rm -rf gpurun_out/prof_${tag}_$cfg
done
