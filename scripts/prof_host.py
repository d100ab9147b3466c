import sys, os, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from deblur4dgs_amd.exposure import render_exposure
dev = torch.device("cuda:0")
name = "tiny"
N, G, K, S, W, H = bench.CONFIGS[name]
sc, d, leaves, wimg, wacc = bench.make_inputs(name, dev)
bg = torch.ones(3, device=dev)
def step():
    for v in leaves.values(): v.grad = None
    res = render_exposure(leaves["means"], leaves["quats"], leaves["scales"], leaves["opacities"], leaves["colors"], 3, leaves["motion_coefs"], leaves["rots"], leaves["transls"], leaves["times"], leaves["RTs"], leaves["viewmat"], d["K"], W, H, background=bg, return_depth=True)
    loss = (res["blended"] * wimg).sum() + (res["acc"] * wacc).sum()
    loss.backward()
for _ in range(20): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(200): step()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:4500])
