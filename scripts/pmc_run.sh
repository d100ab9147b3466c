#!/bin/bash
# usage: [BENCH_ARGS='--share 8'] pmc_run.sh <tag> <counters...>   -> gpurun_out/pmc_<tag>.txt  (per-kernel counter averages)
tag=$1; shift
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out
cd /tmp && rocprofv3 --pmc "$@" --kernel-trace -d $R/gpurun_out/pmc_$tag -o r -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-peaks --sustain 0 $BENCH_ARGS > /dev/null 2>$R/gpurun_out/pmc_$tag.err
cd $R
python - <<PY
import sqlite3, glob, collections
db = sqlite3.connect(glob.glob("gpurun_out/pmc_$tag/*.db")[0])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
rows = cur.execute("select * from counters_collection").fetchall()
ik, ic, iv = cols.index("kernel_name") if "kernel_name" in cols else cols.index("name"), cols.index("counter_name"), cols.index("value")
agg = collections.defaultdict(lambda: [0.0, 0])
for r in rows:
    k = r[ik].replace("(anonymous namespace)::", "")[:40]
    a = agg[(k, r[ic])]; a[0] += r[iv]; a[1] += 1
with open("gpurun_out/pmc_$tag.txt", "w") as f:
    for (k, c), (v, n) in sorted(agg.items()):
        if k.startswith(("k_", "void k_")):
            line = f"{k:42s} {c:28s} avg {v/n:16.1f}  n {n}"
            print(line); f.write(line + "\n")
PY
