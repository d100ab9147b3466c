import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deblur4dgs_amd.synth import make_scene
from deblur4dgs_amd.exposure import render_exposure
N, G, K, S, W, H = [int(x) for x in (sys.argv[2:8] if len(sys.argv) > 7 else (300000, 300000, 6, 8, 512, 288))]
leaf = sys.argv[1]
dev = torch.device("cuda:0")
sc = make_scene(N, G, K, S, W, H, seed=1001)
sc = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in sc.items()}
g = torch.Generator().manual_seed(11)
w = torch.randn(H, W, 3, generator=g).to(dev)
w = torch.nn.functional.avg_pool2d(w.permute(2, 0, 1)[None], 9, 1, 4)[0].permute(1, 2, 0).contiguous()
P = {k: sc[k].clone() for k in ("means", "quats", "scales", "opacities", "colors", "motion_coefs", "rots", "transls")}
def img(Q):
    return render_exposure(Q["means"], Q["quats"], Q["scales"], Q["opacities"], Q["colors"], 3, Q["motion_coefs"], Q["rots"], Q["transls"], sc["times"], sc["RTs"], sc["viewmat"], sc["K"], W, H, background=torch.ones(3, device=dev), return_depth=True)["blended"][..., :3]
P[leaf].requires_grad_()
(img(P) * w).sum().backward()
d = torch.randn(P[leaf].shape, generator=g).to(dev)
an = (P[leaf].grad.double() * d.double()).sum().item()
print(leaf, "analytic", an)
for eps in (1e-5, 1e-4, 1e-3, 1e-2):
    with torch.no_grad():
        Pp = dict(P); Pp[leaf] = P[leaf] + eps * d
        Pm = dict(P); Pm[leaf] = P[leaf] - eps * d
        fd = ((img(Pp).double() - img(Pm).double()) * w.double()).sum().item() / (2 * eps)
    print("  eps", eps, "fd", fd)
