import sys, os, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deblur4dgs_amd.engine as E
import bench
for flag in (False, True):
    E.TILE_ORDER = flag
    sys.argv = ["bench.py", "--steps", "20", "--warmup", "5", "--no-cpu-baseline"]
    print("TILE_ORDER", flag, flush=True)
    bench.main()
