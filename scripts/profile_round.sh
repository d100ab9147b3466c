#!/bin/bash
# scripts/profile_round.sh <tag> : the evidence bench.py's roofline object and DESIGN.md cite, for one round.
#   bench line (cfg2) + rocprofv3 --kernel-trace --stats of the SAME command (cfg2, cfg3, cfg5, refdefault),
#   rocprofv3 --pmc passes (each counter set in its own run, kernel-trace only): FETCH_SIZE, WRITE_SIZE, SQ_*,
#   and the offline lane statistics of the backward's replays.  Everything lands in gpurun_out/<tag>_*.
tag=${1:-r02p}
R=$PWD
export TMPDIR=/tmp
mkdir -p gpurun_out
python bench.py --steps 30 --warmup 5 2>gpurun_out/${tag}_bench.err | tail -1 > gpurun_out/${tag}_bench_cfg2.json
for cfg in cfg2 cfg3 cfg5 refdefault; do
  steps=10; [ $cfg = cfg5 ] && steps=5
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_prof_$cfg -o r -- python $R/bench.py --config $cfg --steps $steps --warmup 3 --no-cpu-baseline --no-profile --no-peaks > $R/gpurun_out/${tag}_bench_under_rocprof_$cfg.json 2>>$R/gpurun_out/${tag}_prof.err)
  db=$(find gpurun_out/${tag}_prof_$cfg -name "*.db" | head -1)
  [ -n "$db" ] && python scripts/rocpd_summary.py $db > gpurun_out/${tag}_kernel_stats_$cfg.csv
  csv=$(find gpurun_out/${tag}_prof_$cfg -name "*kernel_stats.csv" | head -1)
  [ -n "$csv" ] && cp $csv gpurun_out/${tag}_kernel_stats_$cfg.csv
  rm -rf gpurun_out/${tag}_prof_$cfg
done
# one rank's share of an exposure-sharded cfg2 frame (BASELINE config 4: S / P sub-samples, no collectives) under the same profiler
for sh in 4 8; do
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_prof_share$sh -o r -- python $R/bench.py --share $sh --steps 10 --warmup 3 --no-cpu-baseline --no-profile --no-peaks > $R/gpurun_out/${tag}_bench_under_rocprof_share$sh.json 2>>$R/gpurun_out/${tag}_prof.err)
  db=$(find gpurun_out/${tag}_prof_share$sh -name "*.db" | head -1)
  [ -n "$db" ] && python scripts/rocpd_summary.py $db > gpurun_out/${tag}_kernel_stats_share$sh.csv
  rm -rf gpurun_out/${tag}_prof_share$sh
done
python scripts/shard_floor.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_shard_floor.txt
python bench.py --graph --steps 30 --warmup 5 --no-cpu-baseline 2>>gpurun_out/${tag}_bench.err | tail -1 > gpurun_out/${tag}_bench_graph_cfg2.json
python bench.py --force-dist --graph --steps 30 --warmup 5 --no-cpu-baseline 2>>gpurun_out/${tag}_bench.err | tail -1 > gpurun_out/${tag}_bench_graph_rccl_world1_cfg2.json
scripts/pmc_run.sh ${tag}_fetch FETCH_SIZE > /dev/null
scripts/pmc_run.sh ${tag}_write WRITE_SIZE > /dev/null
scripts/pmc_run.sh ${tag}_sq SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_LDS SQ_WAIT_INST_ANY > /dev/null
rm -rf gpurun_out/pmc_${tag}_fetch gpurun_out/pmc_${tag}_write gpurun_out/pmc_${tag}_sq
LANE_STATS_OUT=gpurun_out/${tag}_lane_stats_cfg2.json python scripts/pair_stats.py > gpurun_out/${tag}_pair_stats.txt 2>&1
ls -la gpurun_out | grep $tag
