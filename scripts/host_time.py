"""Host-side cost of one eager step: the frame of a scene so small that the GPU is never the bottleneck ("tiny"), timed
as the Python loop alone (no sync inside) - i.e. how long the host needs to ISSUE a forward + backward."""
import json, subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for cfg in ("tiny", "cfg2"):
    for extra in ([], ["--graph"]):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", cfg, "--no-cpu-baseline", "--no-profile",
                            "--steps", "200", "--warmup", "20"] + extra, capture_output=True, text=True)
        d = json.loads(r.stdout.strip().splitlines()[-1])
        print(cfg, extra, "%.3f ms/step" % d["ms_per_step"])
