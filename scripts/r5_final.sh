#!/bin/bash
# round-5 final evidence: GPU suite + smoke + parity table, the profile round (bench line, rocprofv3 kernel stats, PMC passes, lane statistics,
# shard floor, graph benches), the 17-channel cfg2 profile, the upstream-fixture consumer on mock data
tag=r05s
mkdir -p gpurun_out
R=$PWD
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -6 | tee gpurun_out/${tag}_pytest_gpu.txt
cp gpurun_out/parity_table.md gpurun_out/${tag}_parity_table.md 2>/dev/null; cp gpurun_out/parity_table.json gpurun_out/${tag}_parity_table.json 2>/dev/null
python __graft_entry__.py smoke 2>&1 | tail -2 | tee gpurun_out/${tag}_smoke.txt
python scripts/mock_upstream_fixture.py gpurun_out/mock_upstream > /dev/null 2>&1
D4GS_UPSTREAM_DIR=$PWD/gpurun_out/mock_upstream timeout 600 python -m pytest tests/test_gpu_upstream_fixture.py -q 2>&1 | tail -3 | tee gpurun_out/${tag}_pytest_upstream_mock.txt
rm -rf gpurun_out/mock_upstream
bash scripts/profile_round.sh $tag > gpurun_out/${tag}_profile_round.log 2>&1
(cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_prof_ch16 -o r -- python $R/bench.py --config cfg2 --channels 16 --steps 10 --warmup 3 --no-cpu-baseline --no-profile --no-peaks > $R/gpurun_out/${tag}_bench_under_rocprof_ch16.json 2>>$R/gpurun_out/${tag}_prof.err)
db=$(find gpurun_out/${tag}_prof_ch16 -name "*.db" | head -1)
[ -n "$db" ] && python scripts/rocpd_summary.py $db > gpurun_out/${tag}_kernel_stats_ch16.csv
rm -rf gpurun_out/${tag}_prof_ch16
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_cfg2_driver_flags.json
for c in cfg3 cfg5 refdefault "cfg2 --channels 16" "cfg2 --scale-mul 4" refdefault720; do
  python bench.py --config $c --steps 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']
print('[$c]', '%.3f ms' % d['ms_per_step'], '%.1f M/s' % (d['value']/1e6), d['config'].get('lazy_sort'), {n: round(1e3*t) for n,t in list(k.items())[:10]})"
done 2>&1 | tee gpurun_out/${tag}_other_workloads.txt
sha256sum deblur4dgs_amd/libd4gs.so > gpurun_out/${tag}_lib_sha.txt
ls gpurun_out | grep $tag
