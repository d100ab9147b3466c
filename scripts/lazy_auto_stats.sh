#!/bin/bash
# scripts/lazy_auto_stats.sh : what D4GS_LAZY_SORT=auto sees (live-row fraction, average keys per tile list) and does, per workload
cd "$(dirname "$0")/.."
configs=("--config cfg2" "--config cfg2 --scale-mul 2" "--config cfg2 --scale-mul 4" "--config cfg3" "--config cfg5 --steps 10" "--config refdefault" "--config refdefault --scale-mul 4")
[ -n "$ONLY" ] && configs=("$ONLY")
for c in "${configs[@]}"; do
  for mode in 0:0.3,2048 auto:0.3,2048 auto:0.3,1500 auto:0.5,1000; do
  D4GS_LAZY_AUTO=${mode#*:} D4GS_LAZY_SORT=${mode%%:*} python - $c <<'PY' 2>/dev/null
import sys, json, subprocess, os
sys.argv = ["bench.py"] + sys.argv[1:] + ["--no-cpu-baseline", "--no-peaks", "--no-profile"]
import io, contextlib, runpy
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    try:
        runpy.run_path("bench.py", run_name="__main__")
    except SystemExit:
        pass
d = json.loads(buf.getvalue().strip().splitlines()[-1])
from deblur4dgs_amd import engine
stats = []
for key, f in engine._LIVE_FRAC.items():
    g = engine._guess_get(key)
    stats.append((round(f, 3), g[0] if g else None))
print("[%s] D4GS_LAZY_SORT=%s: %.3f ms; (live fraction, capacity) per shape: %s" % (" ".join(sys.argv[1:-3]), os.environ["D4GS_LAZY_SORT"] + " " + os.environ.get("D4GS_LAZY_AUTO", ""), d["ms_per_step"], stats))
PY
  done
done
