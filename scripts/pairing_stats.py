"""How many backward replay iterations would a 2-way split of the quadrant wave save?  A (quadrant, splat) replay
whose tight alpha >= 1/255 box stays in the top 4 or the bottom 4 pixel rows of the 8x8 quadrant (or left / right 4
columns) could share an iteration with a replay confined to the other half; replays that straddle the mid-line force
the pending singles out first (per-pixel order must be kept).  cfg2, same selection as k_raster_bwd_q (box & <= last)."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from deblur4dgs_amd.exposure import render_exposure

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
N, G, K, S, W, H = bench.CONFIGS[name]
dev = "cuda:0"
sc = bench.scene_of(name, channels=3, scale_mul=float(sys.argv[2]) if len(sys.argv) > 2 else 1.0)
L = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in sc.items()}
res = render_exposure(L["means"], L["quats"], L["scales"], L["opacities"], L["colors"], 3, L.get("motion_coefs"), L.get("rots"),
                      L.get("transls"), L.get("times"), L["RTs"], L["viewmat"], L["K"], W, H, return_depth=True)
st = res["state"]; po, iz = st.proj_out, st.isect; n = st.n_isect
tw, th = st.cfg.tiles
offs = po["tile_offsets"][: S * tw * th + 1].long()
gid = iz["sorted_gid"][:n].long()
tile_of = torch.repeat_interleave(torch.arange(S * tw * th, device=dev), (offs[1:] - offs[:-1]))
pos = torch.arange(n, device=dev)
s_of = tile_of // (tw * th); tl = tile_of % (tw * th); ty, tx = tl // tw, tl % tw
geom = po["geom"].view(S * N, -1)[s_of * N + gid]
mx, my, op, ca, cb, cc = geom[:, 0], geom[:, 1], geom[:, 2], geom[:, 4], geom[:, 5], geom[:, 6]
tau = torch.log(255 * op) * 1.01 + 0.02; det = ca * cc - cb * cb
ex = torch.sqrt(2 * tau * cc / det) + 1e-3; ey = torch.sqrt(2 * tau * ca / det) + 1e-3
ok = (tau > 0) & (det > 0)
last = st.raster["last_ids"].view(S, H, W).long()
pad = torch.nn.functional.pad(last, (0, tw * 16 - W, 0, th * 16 - H), value=-1)
lastq = pad.view(S, th, 2, 8, tw, 2, 8).amax(dim=(3, 6))  # [S,th,2,tw,2] per quadrant
out = {}
tot_now = 0; it_tb = 0; it_lr = 0; cls = torch.zeros(3, dtype=torch.long, device=dev); clsx = torch.zeros(3, dtype=torch.long, device=dev)
for q in range(4):
    qy, qx = q >> 1, q & 1
    x0 = (tx * 16 + qx * 8).float(); y0 = (ty * 16 + qy * 8).float()
    lq = lastq[s_of, ty, qy, tx, qx]
    hit = ok & (mx - ex <= x0 + 7.5) & (mx + ex >= x0 + 0.5) & (my - ey <= y0 + 7.5) & (my + ey >= y0 + 0.5) & (pos <= lq)
    idx = hit.nonzero()[:, 0]
    tot_now += idx.numel()
    for axis in ("tb", "lr"):
        if axis == "tb":
            lo, hi = (my - ey)[idx], (my + ey)[idx]; mid_lo, mid_hi = y0[idx] + 3.5, y0[idx] + 4.5
        else:
            lo, hi = (mx - ex)[idx], (mx + ex)[idx]; mid_lo, mid_hi = x0[idx] + 3.5, x0[idx] + 4.5
        first = hi < mid_hi   # touches no pixel centre of the second half (centres at +4.5 ...)
        second = lo > mid_lo  # touches no pixel centre of the first half
        both = ~(first | second)
        if axis == "tb":
            cls += torch.stack([first.sum(), second.sum(), both.sum()])
        else:
            clsx += torch.stack([first.sum(), second.sum(), both.sum()])
        # segments between straddling replays, per (tile): iterations = n_both + sum over segments of max(n_first, n_second)
        t = tile_of[idx]
        seg = torch.cumsum(both.long(), 0)  # global running id; combine with tile for uniqueness
        key = t * (int(seg.max()) + 2) + seg
        uk, inv = torch.unique(key, return_inverse=True)
        a = torch.zeros(uk.numel(), dtype=torch.long, device=dev).index_add_(0, inv, (first & ~both).long())
        b = torch.zeros(uk.numel(), dtype=torch.long, device=dev).index_add_(0, inv, (second & ~both).long())
        its = int(both.sum()) + int(torch.maximum(a, b).sum())
        if axis == "tb":
            it_tb += its
        else:
            it_lr += its
out = {"config": name, "n_isect": int(n), "replays_now": tot_now, "classes_top_bottom_both": cls.tolist(),
       "classes_left_right_both": clsx.tolist(), "iterations_paired_top_bottom": it_tb, "iterations_paired_left_right": it_lr}
print(json.dumps(out))
