"""Oracle: WHERE the reference's discrete decisions sit close to their thresholds (test infrastructure; see oracle/__init__.py).

The reference path (gsplat 1.1.1 `rasterization`, restated in oracle/raster.py) takes discrete decisions that a float32 evaluation
can take the other way from the fp64 oracle:

  per (pixel, splat)   alpha = min(0.999, o * exp(-sigma)) >= 1/255 (skip otherwise); the 0.999 clamp (a kink: gradient only);
                       T * (1 - alpha) <= 1e-4 (stop);
  per pixel            the compositing ORDER of two splats whose depths agree to the last float32 bits;
  per Gaussian         radius = ceil(3 * sqrt(lambda_max)); the tile rectangle floor / ceil((mean +- radius) / 16); the off-screen
                       and near / far culls;
  per blended pixel    the max / min winner of the exposure blend (flow3d/scene_model.py:392-393) when two sub-samples tie.

`north_star` asks for 1e-4 relative with no allowance; the parity tests allow a bounded FRACTION of elements to miss it and call the
misses "decision flips".  This module makes that a checkable statement of CAUSE: `fragile_pixels` returns, from the fp64 oracle alone,
the pixels at which some decision's margin is below `eps` (natural-log margin for alpha and T, i.e. a relative margin).  The tests
(tests/test_gpu_flip_cause.py) then show (i) every out-of-tolerance image element lies in that set, and (ii) with the loss cotangents
zeroed on that set - a pixel's cotangent scales everything that pixel contributes to every gradient - EVERY element of EVERY gradient
is within 1e-4 of the oracle, no allowance.  So whatever misses 1e-4 in the plain comparison is caused by a decision within `eps` of its
threshold at one of those pixels, and by nothing else.
"""
from __future__ import annotations

import math

import torch

from . import raster

TILE = raster.TILE
_BIG = 1e9


@torch.no_grad()
def pixel_margins(means2d, conics, opacities, depths, flatten_ids, isect_offsets, width: int, height: int, pos_err_px: float | None = None):
    """fp64 tensors of the oracle's projection + its sorted tile lists -> per pixel [H,W]:
    m_alpha  min |ln(o e^-sigma) - ln(1/255)| over the splats of the tile's list the pixel can still composite (T_before > 1e-4),
             LESS what the float32 projected centre alone moves ln(alpha) by: |d sigma / d centre|_1 x pos_err_px (default: 4 ulp of the
             image size - means2d = fx x / z + cx is a float32 of magnitude <= max(W, H)).  A splat two pixels wide has conics of 2 - 3
             px^-2: 1e-4 px of centre error is 5e-4 of alpha at its rim (measured: the one unexplained element of the reference's
             training shape sat at a margin of 1.04e-4 on such a splat);
    m_T      min |ln(T_after) - ln(1e-4)| over the splats it does composite,
    m_clamp  min |ln(o e^-sigma) - ln(0.999)| over the same reachable splats (gradient kink only),
    m_order  min relative depth difference of two list-adjacent splats that BOTH pass the alpha test at the pixel."""
    tile_w, tile_h = math.ceil(width / TILE), math.ceil(height / TILE)
    out = [torch.full((height, width), _BIG, dtype=torch.float64) for _ in range(4)]
    offs = isect_offsets.tolist()
    ln_min, ln_max, ln_stop = math.log(raster.ALPHA_MIN), math.log(raster.ALPHA_MAX), math.log(raster.T_STOP)
    if pos_err_px is None:
        pos_err_px = 4.0 * 2.0 ** -24 * 2.0 ** math.ceil(math.log2(max(width, height)))
    for ty in range(tile_h):
        ys0, ys1 = ty * TILE, min((ty + 1) * TILE, height)
        for tx in range(tile_w):
            t = ty * tile_w + tx
            s, e = offs[t], offs[t + 1]
            if e <= s:
                continue
            xs0, xs1 = tx * TILE, min((tx + 1) * TILE, width)
            g = flatten_ids[s:e]
            py, px = torch.meshgrid(torch.arange(ys0, ys1, dtype=torch.float64) + 0.5,
                                    torch.arange(xs0, xs1, dtype=torch.float64) + 0.5, indexing="ij")
            dx = means2d[g, 0][None, :] - px.reshape(-1, 1)
            dy = means2d[g, 1][None, :] - py.reshape(-1, 1)
            cn = conics[g]
            sigma = 0.5 * (cn[:, 0] * dx * dx + cn[:, 2] * dy * dy) + cn[:, 1] * dx * dy
            lna = torch.log(opacities[g])[None, :] - sigma
            alpha = torch.clamp(torch.exp(lna), max=raster.ALPHA_MAX)
            skip = (sigma < 0) | (alpha < raster.ALPHA_MIN)
            a = torch.where(skip, torch.zeros_like(alpha), alpha)
            T_after = torch.cumprod(1.0 - a, dim=1)
            T_before = torch.cat([torch.ones_like(T_after[:, :1]), T_after[:, :-1]], dim=1)
            reach = T_before > raster.T_STOP * (1.0 - 1e-2)  # (with slack: a stop that moves by one splat moves what is reachable)
            big = torch.full_like(lna, _BIG)
            grad1 = (cn[:, 0] * dx + cn[:, 1] * dy).abs() + (cn[:, 1] * dx + cn[:, 2] * dy).abs()
            d_alpha = torch.where(reach, ((lna - ln_min).abs() - grad1 * pos_err_px).clamp(min=0.0), big)
            d_T = torch.where(reach & ~skip, (torch.log(T_after.clamp(min=1e-300)) - ln_stop).abs(), big)
            d_clamp = torch.where(reach, (lna - ln_max).abs(), big)
            z = depths[g]
            if e - s > 1:
                rel = ((z[1:] - z[:-1]).abs() / z[1:].abs().clamp(min=1e-30))[None, :].expand(lna.shape[0], -1)
                both = (reach & ~skip)[:, 1:] & (reach & ~skip)[:, :-1]
                d_order = torch.where(both, rel, big[:, 1:]).min(dim=1)[0]
            else:
                d_order = big[:, 0]
            hh, ww = ys1 - ys0, xs1 - xs0
            for o, d in zip(out, (d_alpha.min(dim=1)[0], d_T.min(dim=1)[0], d_clamp.min(dim=1)[0], d_order)):
                o[ys0:ys1, xs0:xs1] = d.reshape(hh, ww)
    return dict(alpha=out[0], T=out[1], clamp=out[2], order=out[3])


@torch.no_grad()
def gaussian_toggle_mask(means, quats, scales, opacities, viewmat, K, width: int, height: int, near_plane=0.01, far_plane=1e10, eps2d=0.3,
                         eps_px: float = 1e-3, eps_rel: float = 1e-5, eps_alpha: float = 1e-2):
    """-> ([H,W] bool, number of such Gaussians): the pixels of the TILES whose membership in some Gaussian's list could toggle under a
    perturbation of its projected centre by `eps_px` pixels and of its pre-ceil radius / depth by `eps_rel` relative - ceil(radius), the
    floor / ceil of the tile rectangle, the off-screen cull, the near / far cull - at which that Gaussian passes the alpha test (within
    `eps_alpha`): elsewhere in the toggling tile it is skipped whether it is in the list or not.  (float32 evaluates means2d to ~1e-4 px
    at these image sizes; a tile beyond the 3-sigma radius is reached with alpha >= 1/255 only by splats of opacity > 0.35.)"""
    dt = torch.float64
    means, quats, scales, opacities, viewmat, K = (x.detach().to(dt) for x in (means, quats, scales, opacities, viewmat, K))
    Rcw, tcw = viewmat[:3, :3], viewmat[:3, 3]
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    mc = means @ Rcw.T + tcw
    z = mc[:, 2]
    cov_c = Rcw @ raster.quat_scale_to_covar(quats, scales) @ Rcw.T
    zs = torch.where(z.abs() < 1e-12, torch.full_like(z, 1e-12), z)
    rz = 1.0 / zs
    limx, limy = 1.3 * 0.5 * width / fx, 1.3 * 0.5 * height / fy
    tx = zs * torch.clamp(mc[:, 0] * rz, min=-limx, max=limx)
    ty = zs * torch.clamp(mc[:, 1] * rz, min=-limy, max=limy)
    O = torch.zeros_like(z)
    J = torch.stack([torch.stack([fx * rz, O, -fx * tx * rz * rz], -1), torch.stack([O, fy * rz, -fy * ty * rz * rz], -1)], dim=-2)
    cov2d = J @ cov_c @ J.transpose(-1, -2)
    a, b, c = cov2d[:, 0, 0] + eps2d, 0.5 * (cov2d[:, 0, 1] + cov2d[:, 1, 0]), cov2d[:, 1, 1] + eps2d
    det = a * c - b * b
    mid = 0.5 * (a + c)
    r_raw = 3.0 * torch.sqrt(mid + torch.sqrt(torch.clamp(mid * mid - det, min=0.01)))
    mx, my = fx * mc[:, 0] * rz + cx, fy * mc[:, 1] * rz + cy
    tile_w, tile_h = math.ceil(width / TILE), math.ceil(height / TILE)

    def rect(mx_, my_, r_):
        x0 = torch.floor((mx_ - r_) / TILE).clamp(0, tile_w)
        y0 = torch.floor((my_ - r_) / TILE).clamp(0, tile_h)
        x1 = torch.ceil((mx_ + r_) / TILE).clamp(0, tile_w)
        y1 = torch.ceil((my_ + r_) / TILE).clamp(0, tile_h)
        return x0, y0, x1, y1

    def visible(mx_, my_, r_, z_):
        return ((z_ >= near_plane) & (z_ <= far_plane) & (det > 0) & ~((mx_ + r_ <= 0) | (mx_ - r_ >= width) | (my_ + r_ <= 0) | (my_ - r_ >= height)))

    r_lo, r_hi = torch.ceil(r_raw * (1 - eps_rel)), torch.ceil(r_raw * (1 + eps_rel))
    # outer rectangle: everything any candidate bins; inner: what every candidate bins
    ox0, oy0, _, _ = rect(mx - eps_px, my - eps_px, r_hi)
    _, _, ox1, oy1 = rect(mx + eps_px, my + eps_px, r_hi)
    ix0, iy0, _, _ = rect(mx + eps_px, my + eps_px, r_lo)
    _, _, ix1, iy1 = rect(mx - eps_px, my - eps_px, r_lo)
    vis_all = visible(mx, my, r_lo, z * (1 - eps_rel)) & visible(mx, my, r_lo, z * (1 + eps_rel))
    vis_any = (visible(mx, my, r_hi + eps_px, z * (1 - eps_rel)) | visible(mx, my, r_hi + eps_px, z * (1 + eps_rel))
               | visible(mx, my, r_hi + eps_px, z))
    differs = vis_any & ((ox0 != ix0) | (oy0 != iy0) | (ox1 != ix1) | (oy1 != iy1) | ~vis_all)
    mask = torch.zeros(height, width, dtype=torch.bool)
    py, px = torch.meshgrid(torch.arange(height, dtype=dt) + 0.5, torch.arange(width, dtype=dt) + 0.5, indexing="ij")
    ln_min = math.log(raster.ALPHA_MIN)
    for i in differs.nonzero()[:, 0].tolist():
        X0, Y0, X1, Y1 = int(ox0[i]), int(oy0[i]), int(ox1[i]), int(oy1[i])
        sub = torch.ones(max(Y1 - Y0, 0), max(X1 - X0, 0), dtype=torch.bool)
        if bool(vis_all[i]):  # only the strips between the inner and the outer rectangle can toggle
            sub[max(int(iy0[i]) - Y0, 0):max(int(iy1[i]) - Y0, 0), max(int(ix0[i]) - X0, 0):max(int(ix1[i]) - X0, 0)] = False
        tiles = torch.zeros(tile_h, tile_w, dtype=torch.bool)
        tiles[Y0:Y1, X0:X1] = sub
        dx, dy = mx[i] - px, my[i] - py
        sigma = 0.5 * (c[i] * dx * dx + a[i] * dy * dy) / det[i] - (b[i] / det[i]) * dx * dy  # conic = (c, -b, a) / det
        passes = (torch.log(opacities[i]) - sigma) >= ln_min - eps_alpha
        mask |= tiles.repeat_interleave(TILE, 0).repeat_interleave(TILE, 1)[:height, :width] & passes
    return mask, int(differs.sum())


def fragile_pixels(margins: dict, eps: float, eps_order: float = 1e-6, with_clamp: bool = True):
    """[H,W] bool: some per-pixel decision margin (natural-log, i.e. relative) is below eps; depth-order near-ties below eps_order."""
    f = (margins["alpha"] < eps) | (margins["T"] < eps) | (margins["order"] < eps_order)
    if with_clamp:
        f = f | (margins["clamp"] < eps)
    return f


@torch.no_grad()
def blend_tie_mask(stack: torch.Tensor, channels=(3, 16), eps: float = 1e-4):
    """stack [S,H,W,C] = what the reference's max / min see (raw_0..raw_{S-2}, mean): pixels where the two extreme values of a
    policy channel (3 <- max, 16 <- min) differ by less than eps * max|channel| - the winner, hence the gradient's route, can flip."""
    S, H, W, C = stack.shape
    out = torch.zeros(H, W, dtype=torch.bool)
    if S < 2:
        return out
    for ch, largest in ((channels[0], True), (channels[1], False)):
        if ch < C:
            v = stack[..., ch].double()
            top = torch.topk(v, 2, dim=0, largest=largest)[0]
            # (an exact tie at 0 - no splat carries the channel at that pixel in either sub-sample - routes a gradient nobody receives)
            out |= ((top[0] - top[1]).abs() < eps * max(float(v.abs().max()), 1e-30)) & ~((top[0] == 0) & (top[1] == 0))
    return out
