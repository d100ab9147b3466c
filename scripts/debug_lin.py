import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deblur4dgs_amd.synth import make_scene
from deblur4dgs_amd.exposure import render_exposure
N, G, K, S, W, H = [int(x) for x in (sys.argv[1:7] if len(sys.argv) > 6 else (300000, 300000, 6, 8, 512, 288))]
dev = torch.device("cuda:0")
sc = make_scene(N, G, K, S, W, H, seed=1001)
sc = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in sc.items()}
g = torch.Generator().manual_seed(1)
c = torch.rand(N, 3, generator=g).to(dev).requires_grad_()
d = torch.randn(N, 3, generator=g).to(dev)
w = torch.randn(H, W, 3, generator=g).to(dev)
def R(col, blend=True):
    return render_exposure(sc["means"], sc["quats"], sc["scales"], sc["opacities"], col, 0, sc["motion_coefs"], sc["rots"], sc["transls"], sc["times"], sc["RTs"], sc["viewmat"], sc["K"], W, H, background=torch.ones(3, device=dev), return_depth=True)
r1 = R(c)["blended"][..., :3]
(r1 * w).sum().backward()
with torch.no_grad():
    r2 = R(c + d)["blended"][..., :3]
lhs = (c.grad.double() * d.double()).sum().item()
rhs = ((r2.double() - r1.detach().double()) * w.double()).sum().item()
print("lhs", lhs, "rhs", rhs, "rel", abs(lhs - rhs) / abs(rhs))
# per-channel / positive-only direction
d2 = torch.ones_like(d)
with torch.no_grad():
    r3 = R(c + d2)["blended"][..., :3]
print("ones dir: lhs", (c.grad.double()).sum().item(), "rhs", ((r3.double() - r1.detach().double()) * w.double()).sum().item())
