"""Lane statistics of the composite backward's replays on ANY bench config (scripts/pair_stats.py is the cfg2-only, much more
detailed ancestor): per (tile, splat) intersection, which 8x8 quadrants k_raster_bwd_q replays (tight alpha >= 1/255 box reaches
a 4x4 block of the quadrant that still has a contributor at or behind the splat) and how many of their 64 lanes pass the alpha
test.  bench.py quotes the JSON (stamped with the library's sha256) as roofline.hardware of that config.
usage: lane_stats.py <config> [--channels D]   ->  $LANE_STATS_OUT or gpurun_out/lane_stats_<config>.json"""
import hashlib, json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from deblur4dgs_amd import _lib as _L
from deblur4dgs_amd.exposure import render_exposure

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
channels = int(sys.argv[sys.argv.index("--channels") + 1]) if "--channels" in sys.argv else (16 if name.startswith("refdefault") else 3)
N_, G_, K_, S_, W, H = bench.CONFIGS[name]
sc = bench.scene_of(name, channels=channels)  # the benched scene itself
dev = "cuda:0"
L = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in sc.items()}
for _ in range(3):  # the library's auto decisions (lazy far sort, exact tiles, row mode) settle within 3 renders, as in bench.py
    res = render_exposure(L["means"], L["quats"], L["scales"], L["opacities"], L["colors"], 3, L.get("motion_coefs"), L.get("rots"),
                          L.get("transls"), L["times"], L["RTs"], L["viewmat"], L["K"], W, H,
                          background=torch.ones(channels, device=dev), return_depth=True)
st = res["state"]
po, iz = st.proj_out, st.isect
n = st.n_isect
S, N = st.cfg.S, st.cfg.N
tw, th = st.cfg.tiles
assert tw * 16 == W and th * 16 == H, "statistics assume whole tiles"
offs = po["tile_offsets"][: S * tw * th + 1].long()
gid = iz["sorted_gid"][:n].long()
lens = offs[1:] - offs[:-1]
tile_of = torch.repeat_interleave(torch.arange(S * tw * th, device=dev), lens)
last = st.raster["last_ids"].view(S, H, W)
last_tile = last.view(S, th, 16, tw, 16).permute(0, 1, 3, 2, 4).reshape(S * th * tw, 256).max(-1).values.long()
# only positions at or before the tile's last contributor can be replayed (and, with the lazy far sort, only those are ordered at all)
pos = torch.arange(n, device=dev)
alive = pos <= last_tile[tile_of]
sel = alive.nonzero()[:, 0]
n_alive = int(sel.numel())
geom_all = po["geom"].view(S * N, -1)
K_hits = K_zero = K_valid = KQ_hits = 0
tot_q = tot_valid = 0
hist = torch.zeros(65, device=dev)
CH = 200_000
q_off = torch.tensor([[0, 0], [8, 0], [0, 8], [8, 8]], device=dev)  # (x, y) origin of quadrant q = qy * 2 + qx
for a0 in range(0, n_alive, CH):
    ii = sel[a0:a0 + CH]
    t = tile_of[ii]
    s_of, tl = t // (tw * th), t % (tw * th)
    ty, tx = tl // tw, tl % tw
    g = geom_all[s_of * N + gid[ii]]  # x, y, opacity, depth, conic a b c
    mx, my, op, ca, cb, cc = g[:, 0], g[:, 1], g[:, 2], g[:, 4], g[:, 5], g[:, 6]
    tau = torch.log(255 * op) * 1.01 + 0.02
    det = ca * cc - cb * cb
    ex = torch.sqrt(2 * tau * cc / det) + 1e-3
    ey = torch.sqrt(2 * tau * ca / det) + 1e-3
    ok_box = (tau > 0) & (det > 0)
    px = (tx * 16)[:, None, None] + torch.arange(16, device=dev)[None, None, :] + 0.5
    py = (ty * 16)[:, None, None] + torch.arange(16, device=dev)[None, :, None] + 0.5
    dx, dy = mx[:, None, None] - px, my[:, None, None] - py
    sig = 0.5 * (ca[:, None, None] * dx * dx + cc[:, None, None] * dy * dy) + cb[:, None, None] * dx * dy
    alpha = torch.clamp(op[:, None, None] * torch.exp(-sig), max=0.999)
    lastp = last[s_of[:, None, None], py.long().expand(-1, 16, 16), px.long().expand(-1, 16, 16)]
    valid = (sig >= 0) & (alpha >= 1 / 255) & (ii[:, None, None] <= lastp)
    vq = valid.view(-1, 2, 8, 2, 8).permute(0, 1, 3, 2, 4).reshape(-1, 4, 64).sum(-1)
    qx0 = (tx * 16)[:, None] + q_off[None, :, 0]
    qy0 = (ty * 16)[:, None] + q_off[None, :, 1]
    hit = ok_box[:, None] & (mx[:, None] - ex[:, None] <= qx0 + 7.5) & (mx[:, None] + ex[:, None] >= qx0 + 0.5) & \
        (my[:, None] - ey[:, None] <= qy0 + 7.5) & (my[:, None] + ey[:, None] >= qy0 + 0.5)
    lastq = lastp.reshape(-1, 2, 8, 2, 8).permute(0, 1, 3, 2, 4).reshape(-1, 4, 64).max(-1).values
    lastb = lastp.reshape(-1, 4, 4, 4, 4).permute(0, 1, 3, 2, 4).reshape(-1, 16, 16).max(-1).values
    bx0 = (tx * 16)[:, None] + (torch.arange(16, device=dev) % 4 * 4)[None]
    by0 = (ty * 16)[:, None] + (torch.arange(16, device=dev) // 4 * 4)[None]
    bhit = ok_box[:, None] & (mx[:, None] - ex[:, None] <= bx0 + 3.5) & (mx[:, None] + ex[:, None] >= bx0 + 0.5) & \
        (my[:, None] - ey[:, None] <= by0 + 3.5) & (my[:, None] + ey[:, None] >= by0 + 0.5)
    bk = bhit & (ii[:, None] <= lastb)
    hit_kq = hit & (ii[:, None] <= lastq)
    hit_k = bk.view(-1, 2, 2, 2, 2).permute(0, 1, 3, 2, 4).reshape(-1, 4, 4).any(-1) & hit_kq
    KQ_hits += int(hit_kq.sum())
    K_hits += int(hit_k.sum()); K_zero += int((vq[hit_k] == 0).sum()); K_valid += int(vq[hit_k].sum())
    tot_q += int(hit.sum()); tot_valid += int(vq[hit].sum())
    hist += torch.bincount(vq[hit_k].flatten(), minlength=65).float()
cum = (torch.cumsum(hist, 0) / hist.sum()).tolist()
stats = {"config": f"{name} (the benched scene: bench.scene_of('{name}', channels={channels}), seed {bench.SEEDS[name]})",
         "n_isect": int(n), "n_isect_at_or_before_the_tile_last_contributor": n_alive,
         "lazy_sort": bool(getattr(st.cfg, "lazy_sort", False)),
         "lib_sha256": hashlib.sha256(open(_L.LIB_PATH, "rb").read()).hexdigest(),
         "bwd_nominal_pairs": int(K_hits) * 64,
         "bwd_quadrant_replays": int(K_hits), "bwd_replays_with_no_valid_lane": K_zero / max(K_hits, 1),
         "bwd_valid_pairs": int(K_valid), "bwd_active_lane_fraction": K_valid / (64.0 * max(K_hits, 1)),
         "bwd_active_lane_fraction_of_nonempty_replays": K_valid / (64.0 * max(K_hits - K_zero, 1)),
         "bwd_valid_lane_quantiles": {"<=4": cum[4], "<=8": cum[8], "<=16": cum[16], "<=32": cum[32], "<=48": cum[48]},
         "fwd_quadrant_replays_by_box": int(tot_q), "fwd_valid_pairs": int(tot_valid),
         "note": "valid = alpha >= 1/255, sigma >= 0, at or before the pixel's last contributor; a replay = one (8x8 quadrant wave, "
                 "splat) iteration of k_raster_bwd_q"}
os.makedirs("gpurun_out", exist_ok=True)
json.dump(stats, open(os.environ.get("LANE_STATS_OUT", f"gpurun_out/lane_stats_{name}.json"), "w"), indent=1)
print(json.dumps(stats))
