import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.util import static_inputs
from deblur4dgs_amd.rasterization import rasterization
W, H, N, D = 128, 80, 2500, 3
mode = sys.argv[1] if len(sys.argv) > 1 else "RGB"
inp = static_inputs(N, W, H, seed=203, dtype=torch.float32, D=D)
dev = torch.device("cuda:0")
t = {k: v.to(dev) for k, v in inp.items()}
for k in ("means", "quats", "scales", "opac", "colors", "V"):
    t[k].requires_grad_()
rc, ra, info = rasterization(t["means"], t["quats"], t["scales"], t["opac"], t["colors"], t["V"][None], t["K"][None], W, H, backgrounds=torch.ones(1, D, device=dev), render_mode=mode)
torch.cuda.synchronize(); print("fwd ok", info["n_isect"], flush=True)
loss = rc.sum() + ra.sum()
gm = torch.autograd.grad(loss, info["means2d"], retain_graph=True)
torch.cuda.synchronize(); print("raster bwd ok", gm[0].abs().sum().item(), flush=True)
loss.backward()
torch.cuda.synchronize(); print("full bwd ok", t["means"].grad.abs().sum().item(), flush=True)
