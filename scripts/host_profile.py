"""cProfile of the host side of eager steps on a scene too small to keep the GPU busy (bench's "tiny")."""
import cProfile, pstats, os, sys, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from deblur4dgs_amd.exposure import render_exposure

dev = torch.device("cuda:0")
name = "tiny"
N, G, K, S, W, H = bench.CONFIGS[name]
sc, d, leaves, wimg, wacc = bench.make_inputs(name, dev, channels=3)
bg = torch.ones(3, device=dev)

def step():
    for v in leaves.values():
        v.grad = None
    res = render_exposure(leaves["means"], leaves["quats"], leaves["scales"], leaves["opacities"], leaves["colors"], 3,
                          leaves.get("motion_coefs"), leaves.get("rots"), leaves.get("transls"), leaves.get("times"),
                          leaves["RTs"], leaves["viewmat"], d["K"], W, H, background=bg, return_depth=True)
    loss = torch.dot(res["blended"].reshape(-1), wimg.reshape(-1)) + torch.dot(res["acc"].reshape(-1), wacc.reshape(-1))
    loss.backward()

for _ in range(20):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    step()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])

# ---- the backward runs on the autograd thread, invisible to cProfile above: time our three Python backward functions ----
import time
from deblur4dgs_amd import engine, exposure
acc = {}
def wrap(cls):
    orig = cls.backward
    def timed(ctx, *a):
        t0 = time.perf_counter()
        r = orig(ctx, *a)
        acc[cls.__name__] = acc.get(cls.__name__, 0.0) + time.perf_counter() - t0
        return r
    cls.backward = staticmethod(timed)
fns = [engine.ProjectFn, engine.RasterFn] + [v for v in vars(exposure).values() if isinstance(v, type) and issubclass(v, torch.autograd.Function) and v is not torch.autograd.Function]
for c in fns:
    wrap(c)
torch.cuda.synchronize()
t0 = time.perf_counter(); tb = 0.0
for _ in range(300):
    for v in leaves.values():
        v.grad = None
    res = render_exposure(leaves["means"], leaves["quats"], leaves["scales"], leaves["opacities"], leaves["colors"], 3,
                          leaves.get("motion_coefs"), leaves.get("rots"), leaves.get("transls"), leaves.get("times"),
                          leaves["RTs"], leaves["viewmat"], d["K"], W, H, background=bg, return_depth=True)
    loss = torch.dot(res["blended"].reshape(-1), wimg.reshape(-1)) + torch.dot(res["acc"].reshape(-1), wacc.reshape(-1))
    t1 = time.perf_counter()
    loss.backward()
    tb += time.perf_counter() - t1
torch.cuda.synchronize()
tot = time.perf_counter() - t0
print("per step: total %.3f ms, loss.backward() %.3f ms, of which in our Python backward functions:" % (tot / 0.3, tb / 0.3),
      {k: "%.3f ms" % (v / 0.3) for k, v in acc.items()})
