"""Where does the HOST spend a bench step, and how far ahead of the device does it run?  Times the calls of one eager cfg2 step (no sync inside)
and, every step, how many of the step-end events recorded so far have completed (= how many steps the device is behind the host)."""
import os, sys, time, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from deblur4dgs_amd.exposure import render_exposure
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
N, G, K, S, W, H = bench.CONFIGS[name]
sc, d, leaves, wimg, wacc = bench.make_inputs(name, dev)
bg = torch.ones(3, device=dev)
T = {k: [] for k in ("zero", "render", "loss", "backward", "total", "behind")}
evs = []
def step(i, rec):
    t0 = time.perf_counter()
    for v in leaves.values():
        v.grad = None
    t1 = time.perf_counter()
    res = render_exposure(leaves["means"], leaves["quats"], leaves["scales"], leaves["opacities"], leaves["colors"], 3, leaves.get("motion_coefs"),
                          leaves.get("rots"), leaves.get("transls"), leaves.get("times"), leaves["RTs"], leaves["viewmat"], d["K"], W, H, background=bg,
                          return_depth=True, deferred_size_check=True, fused=True)
    t2 = time.perf_counter()
    loss = torch.dot(res["blended"].reshape(-1), wimg.reshape(-1)) + torch.dot(res["acc"].reshape(-1), wacc.reshape(-1))
    t3 = time.perf_counter()
    loss.backward()
    t4 = time.perf_counter()
    e = torch.cuda.Event(); e.record(); evs.append(e)
    if rec:
        done = sum(1 for x in evs if x.query())
        for k, v in (("zero", t1 - t0), ("render", t2 - t1), ("loss", t3 - t2), ("backward", t4 - t3), ("total", t4 - t0)):
            T[k].append(1e3 * v)
        T["behind"].append(len(evs) - done)
for i in range(40):
    step(i, False)
torch.cuda.synchronize()
evs.clear()
t0 = time.perf_counter()
for i in range(200):
    step(i, True)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(name, "host loop %.3f ms/step, with final sync %.3f ms/step" % (1e3 * t_host / 200, 1e3 * t_all / 200))
for k in ("zero", "render", "loss", "backward", "total"):
    v = T[k]
    print("  %-9s median %.3f  p90 %.3f  max %.3f ms" % (k, statistics.median(v), sorted(v)[int(0.9 * len(v))], max(v)))
b = T["behind"]
print("  steps the device is behind the host when a step's calls return: first 10", b[:10], "median", statistics.median(b), "max", max(b))
