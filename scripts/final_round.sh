#!/bin/bash
# scripts/final_round.sh <tag> : the evidence of one round on ONE library, in one gpurun call (everything lands in gpurun_out/<tag>_*):
#   GPU suite + parity table + flip-cause tables + smoke, the suites under forced modes and with D4GS_SEG=0, randomized stress runs, the upstream-fixture consumer on
#   MOCK files, then per configuration (cfg2, refdefault, cfg3, cfg5): lane statistics, the PMC passes (FETCH / WRITE / two SQ sets, own runs), rocprofv3 --kernel-trace
#   --stats of the bench command, and the bench line with the roofline objects those counters feed; the graph / driver-flag / no-flag lines of cfg2, rank shares, other workloads.
tag=${1:-r06s}
rnd=${2:-$tag}
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
export TMPDIR=/tmp
mkdir -p gpurun_out
set -x
date
scripts/run.sh sha $tag
scripts/run.sh suite $tag --durations=10
cp gpurun_out/flip_cause.json gpurun_out/${tag}_flip_cause_cases.json; cp gpurun_out/flip_cause_refdefault.json gpurun_out/${tag}_flip_cause_refdefault.json
date
scripts/run.sh forced $tag
D4GS_SEG=0 timeout 1500 python -m pytest tests/test_gpu_rasterization.py tests/test_gpu_frame.py tests/test_gpu_exposure.py tests/test_gpu_fullsize_properties.py tests/test_gpu_scene_model.py -q -m gpu -p no:cacheprovider 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -4 | tee gpurun_out/${tag}_pytest_gpu_seg0.txt
python scripts/stress_rows.py 2>&1 | tail -2 | tee gpurun_out/${tag}_stress_rows.txt
python scripts/stress_parity.py 2>&1 | tail -2 | tee gpurun_out/${tag}_stress_parity.txt
python scripts/mock_upstream_fixture.py gpurun_out/mock_upstream > /dev/null 2>&1
D4GS_UPSTREAM_DIR=$PWD/gpurun_out/mock_upstream timeout 600 python -m pytest tests/test_gpu_upstream_fixture.py -q -p no:cacheprovider 2>&1 | tail -3 | tee gpurun_out/${tag}_pytest_upstream_mock.txt
rm -rf gpurun_out/mock_upstream
date
for c in cfg2 refdefault cfg3 cfg5; do
  scripts/run.sh lanes $tag $c
  PMC_MORE=1 scripts/run.sh pmc $tag $c
  python scripts/pmc_to_json.py ${tag}_$c $rnd $c   # (on the box: the bench lines below quote the counters of THIS library; redone from gpurun_out/ at home)
  scripts/run.sh prof $tag $c
  python bench.py --config $c --no-cpu-baseline 2>>gpurun_out/${tag}_bench.err | tail -1 > gpurun_out/${tag}_bench_$c.json
  date
done
python bench.py --steps 30 --warmup 5 2>>gpurun_out/${tag}_bench.err | tail -1 > gpurun_out/${tag}_bench_cfg2.json
python bench.py --gpus 1 --steps 20 --warmup 5 2>>gpurun_out/${tag}_bench.err | tail -1 > gpurun_out/${tag}_bench_cfg2_driver_flags.json
python bench.py 2>>gpurun_out/${tag}_bench.err | tail -1 > gpurun_out/${tag}_bench_cfg2_default_flags.json
python bench.py --graph --steps 30 --warmup 5 --no-cpu-baseline 2>>gpurun_out/${tag}_bench.err | tail -1 > gpurun_out/${tag}_bench_graph_cfg2.json
python bench.py --force-dist --graph --steps 30 --warmup 5 --no-cpu-baseline 2>>gpurun_out/${tag}_bench.err | tail -1 > gpurun_out/${tag}_bench_graph_rccl_world1_cfg2.json
python bench.py --sync-size-check --steps 30 --warmup 5 --no-cpu-baseline 2>>gpurun_out/${tag}_bench.err | tail -1 > gpurun_out/${tag}_bench_cfg2_sync_size_check.json
date
for sh in 2 4 8; do scripts/run.sh prof $tag share$sh --config cfg2 --share $sh; done
scripts/run.sh prof $tag ch16 --config cfg2 --channels 16
python scripts/shard_floor.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_shard_floor.txt
TOPK=10 scripts/run.sh frames ${tag}_other "--config cfg3" "--config cfg5" "--config refdefault" "--config cfg2 --channels 16" "--config cfg2 --scale-mul 4" "--config refdefault720 --steps 10" "--config cfg1"
mv gpurun_out/${tag}_other_frames.txt gpurun_out/${tag}_other_workloads.txt
date
ls gpurun_out | grep $tag | wc -l
