"""Tile-list length distribution of a bench config (sizes the sort classes)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from deblur4dgs_amd.exposure import render_exposure
name = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
N, G, K, S, W, H = bench.CONFIGS[name]
sc, d, L, wimg, wacc = bench.make_inputs(name, "cuda:0")
with torch.no_grad():
    res = render_exposure(L["means"], L["quats"], L["scales"], L["opacities"], L["colors"], 3, L["motion_coefs"], L["rots"],
                          L["transls"], L["times"], L["RTs"], L["viewmat"], d["K"], W, H, return_depth=True)
st = res["state"]
tw, th = st.cfg.tiles
offs = st.proj_out["tile_offsets"][: S * tw * th + 1].long()
c = (offs[1:] - offs[:-1]).float()
q = torch.tensor([0.5, 0.9, 0.99, 0.999, 1.0], device=c.device)
print(name, "tiles", c.numel(), "n_isect", int(c.sum()), "mean", float(c.mean()), "quantiles 50/90/99/99.9/max", [int(v) for v in torch.quantile(c, q)])
for lo, hi in ((0, 512), (512, 1024), (1024, 2048), (2048, 4096), (4096, 8192), (8192, 16384), (16384, 1 << 30)):
    m = (c > lo) & (c <= hi)
    n = c[m]
    work = float((n * torch.log2(n.clamp(min=2)) ** 2).sum())
    print(f"  ({lo:6d}, {hi:6d}]: {int(m.sum()):6d} tiles, {int(n.sum()):9d} keys, n log^2 n work {work:.3e}")
