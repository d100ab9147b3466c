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
