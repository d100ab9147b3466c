"""How well are the 64 lanes of a quadrant-wave used?  For cfg2: per (tile, splat) intersection, which 8x8 quadrants the
tight alpha >= 1/255 box touches (what the kernels replay) and how many of their pixels actually pass alpha >= 1/255."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deblur4dgs_amd.synth import make_scene
from deblur4dgs_amd.exposure import render_exposure
import bench

sc = bench.scene_of("cfg2")  # the benched scene itself (seed 1001): bench.py quotes these statistics beside its timing
dev = "cuda:0"
L = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in sc.items()}
res = render_exposure(L["means"], L["quats"], L["scales"], L["opacities"], L["colors"], 3, L["motion_coefs"], L["rots"],
                      L["transls"], L["times"], L["RTs"], L["viewmat"], L["K"], 512, 288, return_depth=True)
st = res["state"]
po, iz = st.proj_out, st.isect
n = st.n_isect
S, N = st.cfg.S, st.cfg.N
tw, th = st.cfg.tiles
offs = po["tile_offsets"][: S * tw * th + 1].long()
gid = iz["sorted_gid"][:n].long()
tile_of = torch.repeat_interleave(torch.arange(S * tw * th, device=dev), (offs[1:] - offs[:-1]))
s_of = tile_of // (tw * th)
tl = tile_of % (tw * th)
ty, tx = tl // tw, tl % tw
geom = po["geom"].view(S * N, -1)[s_of * N + gid]  # [n, 8]: x, y, opacity, depth, conic a b c, ...
mx, my, op = geom[:, 0], geom[:, 1], geom[:, 2]
ca, cb, cc = geom[:, 4], geom[:, 5], geom[:, 6]
last = st.raster["last_ids"].view(S, 288, 512)
tot_q = 0; tot_valid = 0; tot_tile_valid = 0; hist = torch.zeros(65, device=dev)
tau = torch.log(255 * op) * 1.01 + 0.02
det = ca * cc - cb * cb
ex = torch.sqrt(2 * tau * cc / det) + 1e-3
ey = torch.sqrt(2 * tau * ca / det) + 1e-3
ok_box = (tau > 0) & (det > 0)
CH = 200_000
idx_in_list = torch.arange(n, device=dev)
B_hits = B_valid = B_zero = 0
K_hits = K_zero = K_geo_zero = 0
blk_len_k = torch.zeros(S * tw * th, 16, dtype=torch.long, device=dev)
blk_len_g = torch.zeros_like(blk_len_k)
q_len_g = torch.zeros(S * tw * th, 4, dtype=torch.long, device=dev)
last_tile = last.view(S, th, 16, tw, 16).permute(0, 1, 3, 2, 4).reshape(S * th * tw, 256).max(-1).values.long()
maxlen = int((offs[1:] - offs[:-1]).max())
NCHK = {64: (maxlen + 63) // 64 + 1, 128: (maxlen + 127) // 128 + 1, 256: (maxlen + 255) // 256 + 1}
chunk_cnt = {nb: torch.zeros(S * tw * th, NCHK[nb], 16, dtype=torch.int32, device=dev) for nb in NCHK}
chunk_cnt_g = {nb: torch.zeros(S * tw * th, NCHK[nb], 16, dtype=torch.int32, device=dev) for nb in NCHK}
chunk_q = {nb: torch.zeros(S * tw * th, NCHK[nb], 4, dtype=torch.int32, device=dev) for nb in NCHK}
blk_len = torch.zeros(S * tw * th, 16, dtype=torch.long, device=dev)
blk_len_nz = torch.zeros_like(blk_len)
q_len = torch.zeros(S * tw * th, 4, dtype=torch.long, device=dev)
for a0 in range(0, n, CH):
    sl = slice(a0, min(n, a0 + CH))
    px = (tx[sl] * 16)[:, None, None] + torch.arange(16, device=dev)[None, None, :] + 0.5
    py = (ty[sl] * 16)[:, None, None] + torch.arange(16, device=dev)[None, :, None] + 0.5
    dx, dy = mx[sl, None, None] - px, my[sl, None, None] - py
    sig = 0.5 * (ca[sl, None, None] * dx * dx + cc[sl, None, None] * dy * dy) + cb[sl, None, None] * dx * dy
    alpha = torch.clamp(op[sl, None, None] * torch.exp(-sig), max=0.999)
    inimg = (px < 512) & (py < 288)
    # contributes only up to the pixel's last contributor
    lastp = last[s_of[sl, None, None], py.long().clamp(max=287).expand(-1, 16, 16), px.long().clamp(max=511).expand(-1, 16, 16)]
    valid = (sig >= 0) & (alpha >= 1 / 255) & inimg & (idx_in_list[sl, None, None] <= lastp)
    vq = valid.view(-1, 2, 8, 2, 8).permute(0, 1, 3, 2, 4).reshape(-1, 4, 64).sum(-1)  # per quadrant valid lanes
    # replayed quadrants: tight box touches the quadrant
    qx0 = (tx[sl] * 16)[:, None] + torch.tensor([0, 8, 0, 8], device=dev)[None]
    qy0 = (ty[sl] * 16)[:, None] + torch.tensor([0, 0, 8, 8], device=dev)[None]
    hit = ok_box[sl, None] & (mx[sl, None] - ex[sl, None] <= qx0 + 7.5) & (mx[sl, None] + ex[sl, None] >= qx0 + 0.5) & \
        (my[sl, None] - ey[sl, None] <= qy0 + 7.5) & (my[sl, None] + ey[sl, None] >= qy0 + 0.5)
    anyv = vq > 0
    geo = (sig >= 0) & (alpha >= 1 / 255) & inimg
    gq = geo.view(-1, 2, 8, 2, 8).permute(0, 1, 3, 2, 4).reshape(-1, 4, 64).sum(-1)
    lastq = lastp.reshape(-1, 2, 8, 2, 8).permute(0, 1, 3, 2, 4).reshape(-1, 4, 64).max(-1).values
    gb = geo.view(-1, 4, 4, 4, 4).permute(0, 1, 3, 2, 4).reshape(-1, 16, 16).sum(-1)
    lastb = lastp.reshape(-1, 4, 4, 4, 4).permute(0, 1, 3, 2, 4).reshape(-1, 16, 16).max(-1).values
    # ---- 4x4 blocks: 16 per tile; row r of quadrant q handles block (q, r)
    vb = valid.view(-1, 4, 4, 4, 4).permute(0, 1, 3, 2, 4).reshape(-1, 16, 16).sum(-1)  # [n, block(by*4+bx)] valid lanes
    bx0 = (tx[sl] * 16)[:, None] + (torch.arange(16, device=dev) % 4 * 4)[None]
    by0 = (ty[sl] * 16)[:, None] + (torch.arange(16, device=dev) // 4 * 4)[None]
    bhit = ok_box[sl, None] & (mx[sl, None] - ex[sl, None] <= bx0 + 3.5) & (mx[sl, None] + ex[sl, None] >= bx0 + 0.5) & \
        (my[sl, None] - ey[sl, None] <= by0 + 3.5) & (my[sl, None] + ey[sl, None] >= by0 + 0.5)
    B_hits = B_hits + int(bhit.sum()); B_valid = B_valid + int(vb[bhit].sum()); B_zero = B_zero + int((vb[bhit] == 0).sum())
    # per-tile per-block list lengths accumulate by tile -> need scatter add
    blk_len.index_put_((tile_of[sl][:, None].expand(-1, 16), torch.arange(16, device=dev)[None].expand(bhit.shape[0], -1)),
                       bhit.long(), accumulate=True)
    blk_len_nz.index_put_((tile_of[sl][:, None].expand(-1, 16), torch.arange(16, device=dev)[None].expand(bhit.shape[0], -1)),
                          (bhit & (vb > 0)).long(), accumulate=True)
    ti16 = (tile_of[sl][:, None].expand(-1, 16), torch.arange(16, device=dev)[None].expand(bhit.shape[0], -1))
    bk = bhit & (idx_in_list[sl, None] <= lastb)
    # what the backward kernel replays (round 5): the box reaches a 4x4 block of the quadrant that still has a contributor at or
    # behind the splat (until round 5: box & <= the QUADRANT's last contributor - `hit_kq`)
    hit_kq = hit & (idx_in_list[sl, None] <= lastq)
    hit_k = bk.view(-1, 2, 2, 2, 2).permute(0, 1, 3, 2, 4).reshape(-1, 4, 4).any(-1) & hit_kq
    KQ_hits = (KQ_hits if "KQ_hits" in globals() else 0) + int(hit_kq.sum())
    K_hits += int(hit_k.sum()); K_zero += int((vq[hit_k] == 0).sum()); K_geo_zero += int((gq[hit_k] == 0).sum())
    K_valid = (K_valid if "K_valid" in globals() else 0) + int(vq[hit_k].sum())
    blk_len_k.index_put_(ti16, bk.long(), accumulate=True)
    blk_len_g.index_put_(ti16, (bk & (gb > 0)).long(), accumulate=True)
    q_len_g.index_put_((tile_of[sl][:, None].expand(-1, 4), torch.arange(4, device=dev)[None].expand(hit.shape[0], -1)),
                     (hit_k & (gq > 0)).long(), accumulate=True)
    for nb in NCHK:
        hi_t = torch.minimum(last_tile[tile_of[sl]], offs[tile_of[sl] + 1] - 1)
        ck = ((hi_t - idx_in_list[sl]).clamp(min=0) // nb)
        inb = (idx_in_list[sl] <= hi_t)
        ii = (tile_of[sl][:, None].expand(-1, 16), ck[:, None].expand(-1, 16), torch.arange(16, device=dev)[None].expand(bhit.shape[0], -1))
        chunk_cnt[nb].index_put_(ii, (bk & inb[:, None]).int(), accumulate=True)
        chunk_cnt_g[nb].index_put_(ii, (bk & (gb > 0) & inb[:, None]).int(), accumulate=True)
        i4 = (tile_of[sl][:, None].expand(-1, 4), ck[:, None].expand(-1, 4), torch.arange(4, device=dev)[None].expand(hit.shape[0], -1))
        chunk_q[nb].index_put_(i4, (hit_k & inb[:, None]).int(), accumulate=True)
    q_len.index_put_((tile_of[sl][:, None].expand(-1, 4), torch.arange(4, device=dev)[None].expand(hit.shape[0], -1)),
                     hit.long(), accumulate=True)
    tot_q += int(hit.sum()); tot_valid += int(vq[hit].sum()); tot_tile_valid += int(vq.sum())
    hist += torch.bincount(vq[hit].flatten(), minlength=65).float()
print(f"intersections {n}, quadrant replays by box {tot_q} ({tot_q / n:.2f} per isect), valid pairs in replayed quadrants {tot_valid}"
      f" (of all valid {tot_tile_valid}); lane utilisation {tot_valid / (64 * tot_q):.3f}")
print("replayed quadrants with 0 valid lanes: %.3f" % float(hist[0] / hist.sum()))
cum = torch.cumsum(hist, 0) / hist.sum()
print("valid-lane quantiles: <=8: %.3f  <=16: %.3f  <=32: %.3f  <=48: %.3f" % (cum[8], cum[16], cum[32], cum[48]))

print(f"4x4 block hits by box {B_hits} ({B_hits / n:.2f} per isect), zero-valid {B_zero / B_hits:.3f}, lane utilisation {B_valid / (16 * B_hits):.3f}")
# quadrant q = (qy, qx) owns blocks with by in {2qy, 2qy+1}, bx in {2qx, 2qx+1}
bl = blk_len.view(-1, 2, 2, 2, 2).permute(0, 1, 3, 2, 4).reshape(-1, 4, 4)  # [tile, quadrant, row]
it_max = bl.max(-1).values.sum()
it_avg = bl.float().mean(-1).sum()
print(f"wave iterations: now (quadrant lists) {int(q_len.sum())}; 4 rows x own block list: sum of max {int(it_max)}, sum of mean {float(it_avg):.0f}")
bl2 = blk_len_nz.view(-1, 2, 2, 2, 2).permute(0, 1, 3, 2, 4).reshape(-1, 4, 4)
print(f"   with an exact (non-empty only) block test: sum of max {int(bl2.max(-1).values.sum())}")
# alternative: 8 rows of 8 lanes?  2x4 px blocks are too small; alternative 16 lanes = 8x2? skip

print(f"bwd kernel today: quadrant replays (box reaches a 4x4 block with a contributor at or behind the splat) {K_hits} (with the quadrant's last contributor only: {KQ_hits}); zero-valid {K_zero / K_hits:.3f}; zero by geometry alone {K_geo_zero / K_hits:.3f}")
print(f"   quadrant replays with an exact ellipse test: {int(q_len_g.sum())}")
def summax(t):
    return int(t.view(-1, 2, 2, 2, 2).permute(0, 1, 3, 2, 4).reshape(-1, 4, 4).max(-1).values.sum())
print(f"4-row scheme iterations: box & <= block last: {summax(blk_len_k)};  + exact ellipse test: {summax(blk_len_g)}")

for nb in NCHK:
    c = chunk_cnt[nb].view(-1, NCHK[nb], 2, 2, 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(-1, NCHK[nb], 4, 4)
    cg = chunk_cnt_g[nb].view(-1, NCHK[nb], 2, 2, 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(-1, NCHK[nb], 4, 4)
    print(f"batch {nb}: quadrant iterations today {int(chunk_q[nb].sum())}; 4-row scheme sum over (batch, wave) of max row: "
          f"{int(c.max(-1).values.sum())} (box), {int(cg.max(-1).values.sum())} (exact); per-wave barrier-max over waves (box): {int(c.max(-1).values.max(-1).values.sum() * 4)}")

import json
import hashlib
from deblur4dgs_amd import _lib as _L
stats = {"config": "cfg2 (the benched scene: bench.scene_of('cfg2'), seed 1001)", "n_isect": int(n),
         "lib_sha256": hashlib.sha256(open(_L.LIB_PATH, "rb").read()).hexdigest(),
         "bwd_nominal_pairs": int(K_hits) * 64,
         "bwd_quadrant_replays": int(K_hits), "bwd_replays_with_no_valid_lane": K_zero / K_hits,
         "bwd_valid_pairs": int(K_valid), "bwd_active_lane_fraction": K_valid / (64.0 * K_hits),
         "bwd_active_lane_fraction_of_nonempty_replays": K_valid / (64.0 * (K_hits - K_zero)),
         "fwd_quadrant_replays_by_box": int(tot_q), "fwd_valid_pairs": int(tot_valid),
         "note": "valid = alpha >= 1/255, sigma >= 0, inside the image, at or before the pixel's last contributor; a replay = "
                 "one (8x8 quadrant wave, splat) iteration of k_raster_bwd_q"}
os.makedirs("gpurun_out", exist_ok=True)
json.dump(stats, open(os.environ.get("LANE_STATS_OUT", "gpurun_out/lane_stats_cfg2.json"), "w"), indent=1)
print(json.dumps(stats))
