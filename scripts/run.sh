#!/bin/bash
# scripts/run.sh <what> ... : ONE parametrised runner for everything a gpurun call does in this repo (replaces the per-batch
# r5_[a-z].sh one-shots of round 5, which live on in git history).  Everything lands in gpurun_out/<tag>_*.
#
#   run.sh suite  <tag> [pytest args]            GPU suite (+ parity table) + smoke()
#   run.sh bench  <tag> [bench args]             one bench line -> <tag>_bench.json (stderr -> <tag>_bench.err)
#   run.sh frames <tag> "<bench args>" ...       frame time + top kernels of each argument string, one line each
#   run.sh prof   <tag> <cfgkey> [bench args]    rocprofv3 --kernel-trace --stats of bench.py -> <tag>_kernel_stats_<cfgkey>.csv
#   run.sh pmc    <tag> <cfgkey> [bench args]    three rocprofv3 --pmc passes (own runs, kernel-trace only): FETCH_SIZE, WRITE_SIZE, SQ_*
#   run.sh lanes  <tag> <cfg> [--channels D]     scripts/lane_stats.py -> <tag>_lane_stats_<cfg>.json
#   run.sh ab     "<bench args>" name ...        scripts/ab_run.sh (product build vs scripts/ablate/libd4gs_<name>.so)
#   run.sh forced <tag> [pytest args]            the rasterizer / frame suites under forced lazy / sparse / exact-tiles modes
#   run.sh sha    <tag>                          sha256 of the libd4gs.so the numbers describe
# <cfgkey> = a bench --config name; extra bench args (e.g. "--config cfg2 --channels 16") may follow for non-default workloads.
what=$1; tag=$2; shift 2
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R
export TMPDIR=/tmp
mkdir -p gpurun_out
SQ_SET="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_LDS SQ_WAIT_INST_ANY"
SQ_SET2="SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA"
line() {  # stdin: a bench JSON line -> one readable line
  python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernels_ms_per_step',{})
    print('[$1]', '%.3f ms' % d['ms_per_step'], '%.1f M/s' % (d['value']/1e6), 'n_isect', d.get('n_isect_per_step'), d['config'].get('lazy_sort'), {n: round(1e3*t) for n,t in list(k.items())[:${TOPK:-10}]})
except Exception as e:
    print('[$1] FAILED', repr(e))"
}
case $what in
suite)
  timeout ${SUITE_TIMEOUT:-1800} python -m pytest tests -q -m gpu -p no:cacheprovider "$@" 2>&1 | grep -E "passed|failed|FAILED|Error|error" | tail -12 | tee gpurun_out/${tag}_pytest_gpu.txt
  cp gpurun_out/parity_table.md gpurun_out/${tag}_parity_table.md 2>/dev/null; cp gpurun_out/parity_table.json gpurun_out/${tag}_parity_table.json 2>/dev/null
  python __graft_entry__.py smoke 2>&1 | tail -2 | tee gpurun_out/${tag}_smoke.txt ;;
bench)
  python bench.py "$@" 2>gpurun_out/${tag}_bench.err | tail -1 > gpurun_out/${tag}_bench.json; head -c 600 gpurun_out/${tag}_bench.json; echo ;;
frames)
  for c in "$@"; do python bench.py $c --no-cpu-baseline --sustain 0 2>/dev/null | line "$c"; done 2>&1 | tee -a gpurun_out/${tag}_frames.txt ;;
prof)
  cfg=$1; shift
  args="--config $cfg"; [ $# -gt 0 ] && args="$*"
  steps=10; case "$args" in *cfg5*) steps=5;; esac
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_prof_$cfg -o r -- python $R/bench.py $args --steps $steps --warmup 3 --no-cpu-baseline --no-profile --no-peaks --sustain 0 > $R/gpurun_out/${tag}_bench_under_rocprof_$cfg.json 2>>$R/gpurun_out/${tag}_prof.err)
  db=$(find gpurun_out/${tag}_prof_$cfg -name "*.db" | head -1)
  [ -n "$db" ] && python scripts/rocpd_summary.py $db > gpurun_out/${tag}_kernel_stats_$cfg.csv
  csv=$(find gpurun_out/${tag}_prof_$cfg -name "*kernel_stats.csv" | head -1)
  [ -n "$csv" ] && cp $csv gpurun_out/${tag}_kernel_stats_$cfg.csv
  rm -rf gpurun_out/${tag}_prof_$cfg
  head -8 gpurun_out/${tag}_kernel_stats_$cfg.csv ;;
pmc)
  cfg=$1; shift
  export BENCH_ARGS="--config $cfg"; [ $# -gt 0 ] && export BENCH_ARGS="$*"
  case "$BENCH_ARGS" in *cfg5*) export BENCH_ARGS="$BENCH_ARGS --pre-roll 10";; esac  # (counter passes serialise the launches)
  scripts/pmc_run.sh ${tag}_${cfg}_fetch FETCH_SIZE > /dev/null
  scripts/pmc_run.sh ${tag}_${cfg}_write WRITE_SIZE > /dev/null
  scripts/pmc_run.sh ${tag}_${cfg}_sq $SQ_SET > /dev/null
  [ -n "$PMC_MORE" ] && scripts/pmc_run.sh ${tag}_${cfg}_sq2 $SQ_SET2 > /dev/null
  rm -rf gpurun_out/pmc_${tag}_${cfg}_fetch gpurun_out/pmc_${tag}_${cfg}_write gpurun_out/pmc_${tag}_${cfg}_sq gpurun_out/pmc_${tag}_${cfg}_sq2
  grep -E "k_raster|k_gather|k_emit" gpurun_out/pmc_${tag}_${cfg}_sq.txt | head -40 ;;
lanes)
  cfg=$1; shift
  LANE_STATS_OUT=gpurun_out/${tag}_lane_stats_$cfg.json python scripts/lane_stats.py $cfg "$@" 2>&1 | grep -v amdgpu.ids | tail -1 ;;
ab)
  bash scripts/ab_run.sh "$tag" "$@" ;;
forced)
  for m in "D4GS_LAZY_SORT=1" "D4GS_BWD_ROWS=sparse" "D4GS_BWD_ROWS=dense" "D4GS_EXACT_TILES=1"; do
    echo "== $m"; env $m timeout 1500 python -m pytest tests/test_gpu_rasterization.py tests/test_gpu_frame.py tests/test_gpu_exposure.py -q -m gpu -p no:cacheprovider "$@" 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -4
  done 2>&1 | tee gpurun_out/${tag}_pytest_gpu_forced_modes.txt ;;
sha)
  sha256sum deblur4dgs_amd/libd4gs.so | tee gpurun_out/${tag}_lib_sha.txt ;;
*) echo "unknown: $what" >&2; exit 2 ;;
esac
