"""What fraction of the (tile, splat) intersections that survive the tight bounding-box test (D4GS_EXACT_CULL) would an
exact ellipse-vs-tile test remove?  (minimum of sigma over the tile's 16x16 rectangle of pixel centres vs ln(255 opacity))"""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from deblur4dgs_amd.exposure import render_exposure
out = []
for name, mul in (("cfg2", 1.0), ("cfg3", 1.0), ("cfg2", 4.0), ("refdefault", 1.0)):
    N, G, K, S, W, H = bench.CONFIGS[name]
    dev = "cuda:0"
    sc = bench.scene_of(name, channels=3, scale_mul=mul)
    L = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in sc.items()}
    res = render_exposure(L["means"], L["quats"], L["scales"], L["opacities"], L["colors"], 3, L.get("motion_coefs"), L.get("rots"),
                          L.get("transls"), L.get("times"), L["RTs"], L["viewmat"], L["K"], W, H, return_depth=True)
    st = res["state"]; po, iz = st.proj_out, st.isect; n = st.n_isect
    tw, th = st.cfg.tiles
    offs = po["tile_offsets"][: S * tw * th + 1].long()
    gid = iz["sorted_gid"][:n].long()
    tile_of = torch.repeat_interleave(torch.arange(S * tw * th, device=dev), (offs[1:] - offs[:-1]))
    s_of = tile_of // (tw * th); tl = tile_of % (tw * th); ty, tx = tl // tw, tl % tw
    g = po["geom"].view(S * N, -1)[s_of * N + gid]
    mx, my, op, a, b, c = g[:, 0], g[:, 1], g[:, 2], g[:, 4], g[:, 5], g[:, 6]
    tau = torch.log(255 * op)
    xl, xh, yl, yh = tx * 16 + 0.5, torch.clamp(tx * 16 + 15.5, max=W - 0.5), ty * 16 + 0.5, torch.clamp(ty * 16 + 15.5, max=H - 0.5)
    dxl, dxh, dyl, dyh = mx - xh, mx - xl, my - yh, my - yl
    inside = (dxl <= 0) & (dxh >= 0) & (dyl <= 0) & (dyh >= 0)
    f = lambda dx, dy: 0.5 * (a * dx * dx + c * dy * dy) + b * dx * dy
    cl = lambda v, lo, hi: torch.minimum(torch.maximum(v, lo), hi)
    m = torch.minimum(torch.minimum(f(dxl, cl(-b / c * dxl, dyl, dyh)), f(dxh, cl(-b / c * dxh, dyl, dyh))),
                      torch.minimum(f(cl(-b / a * dyl, dxl, dxh), dyl), f(cl(-b / a * dyh, dxl, dxh), dyh)))
    hit = inside | (m <= tau)
    out.append({"config": name, "scale_mul": mul, "n_isect": int(n), "removed_by_exact_tile_test": float((~hit).float().mean())})
    del res, st
    torch.cuda.empty_cache()
print(json.dumps(out))
