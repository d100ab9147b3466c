import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.util import static_inputs
from deblur4dgs_amd.rasterization import rasterization
W, H, N = 256, 160, 20000
inp = static_inputs(N, W, H, seed=77, dtype=torch.float32, D=3, scale_mul=2.0)
t = {k: v.cuda() for k, v in inp.items()}
def run(cull):
    rc, ra, info = rasterization(t["means"], t["quats"], t["scales"], t["opac"], t["colors"], t["V"][None], t["K"][None], W, H,
                                 backgrounds=torch.tensor([[0.2, 0.5, 0.8]]).cuda(), render_mode="RGB+ED", exact_cull=cull)
    torch.cuda.synchronize()
    return rc.clone(), ra.clone(), info
a1, _, i1 = run(True); a2, _, i2 = run(True); b1, _, j1 = run(False); b2, _, _ = run(False)
print("cull twice equal:", torch.equal(a1, a2), " nocull twice equal:", torch.equal(b1, b2), " cull vs nocull:", torch.equal(a1, b1))
d = (a1 - b1).abs()
print("max diff", d.max().item(), "n diff px", (d.amax(-1) > 0).sum().item())
idx = (d.amax(-1) > 0).nonzero()
print(idx[:10].tolist())
if len(idx):
    _, y, x = idx[0].tolist()
    print("pixel", x, y, a1[0, y, x].tolist(), b1[0, y, x].tolist(), "last", i1["last_ids"][0, y, x].item(), j1["last_ids"][0, y, x].item())
