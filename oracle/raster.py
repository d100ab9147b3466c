"""Oracle: restatement of `gsplat==1.1.1` `rasterization(packed=False)` in plain torch.

Test infrastructure (see oracle/__init__.py).  **Parity unpinned**: gsplat is a CUDA-only pip
dependency of the reference (`requirements.txt:137`, call site `flow3d/scene_model.py:360-373`)
that is not vendored and cannot be installed here; the reference holds no tests / golden images.
This file restates gsplat 1.1.1's published algorithm:

  projection   fully_fused_projection (quat+scale -> Sigma3, world->cam, near/far cull, perspective
               Jacobian with the 1.3*tan(fov) clamp, eps2d=0.3 blur, conic, radius=ceil(3*sqrt(l_max)),
               off-screen cull)
  binning      isect_tiles (tile AABB, key = tile_id<<32 | float-bits(depth)), stable sort,
               isect_offset_encode
  composite    rasterize_to_pixels (front-to-back; alpha=min(0.999,o*exp(-sigma)); skip sigma<0 or
               alpha<1/255; stop when T*(1-alpha)<=1e-4; + T*background)
  RGB+ED       last channel = sum(w*z) / max(alpha, 1e-10)

dtype-generic (fp32 / fp64), differentiable by autograd (discrete decisions are constants, exactly as
in gsplat's hand-written adjoint; `clamp` passes gradient on the closed interval like gsplat's `<=`).
Camera count C is 1 (the reference asserts it, scene_model.py:249).
"""
from __future__ import annotations

import math

import torch

from .deform import quat_wxyz_to_rotmat

TILE = 16
ALPHA_MIN = 1.0 / 255.0
ALPHA_MAX = 0.999
T_STOP = 1e-4


def quat_scale_to_covar(quats: torch.Tensor, scales: torch.Tensor) -> torch.Tensor:
    R = quat_wxyz_to_rotmat(quats)
    M = R * scales[..., None, :]
    return M @ M.transpose(-1, -2)


def project(
    means: torch.Tensor,  # [N,3]
    quats: torch.Tensor,  # [N,4] wxyz, un-normalised allowed
    scales: torch.Tensor,  # [N,3]
    viewmat: torch.Tensor,  # [4,4] world->cam
    K: torch.Tensor,  # [3,3]
    width: int,
    height: int,
    near_plane: float = 0.01,
    far_plane: float = 1e10,
    eps2d: float = 0.3,
    radius_clip: float = 0.0,
):
    """-> radii int32 [N], means2d [N,2], depths [N], conics [N,3].  Culled Gaussians get
    radii == 0 and zeros elsewhere (gsplat leaves them uninitialised)."""
    N = means.shape[0]
    dt, dev = means.dtype, means.device
    Rcw, tcw = viewmat[:3, :3], viewmat[:3, 3]
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]

    with torch.no_grad():
        z_all = (means @ Rcw.T + tcw)[:, 2]
        ok = (z_all >= near_plane) & (z_all <= far_plane)
    idx = ok.nonzero()[:, 0]

    mc = means[idx] @ Rcw.T + tcw  # [n,3]
    cov = quat_scale_to_covar(quats[idx], scales[idx])
    cov_c = Rcw @ cov @ Rcw.T

    x, y, z = mc.unbind(-1)
    tanx = 0.5 * width / fx
    tany = 0.5 * height / fy
    limx, limy = 1.3 * tanx, 1.3 * tany
    rz = 1.0 / z
    rz2 = rz * rz
    tx = z * torch.clamp(x * rz, min=-limx, max=limx)
    ty = z * torch.clamp(y * rz, min=-limy, max=limy)
    O = torch.zeros_like(z)
    J = torch.stack(
        [
            torch.stack([fx * rz, O, -fx * tx * rz2], -1),
            torch.stack([O, fy * rz, -fy * ty * rz2], -1),
        ],
        dim=-2,
    )  # [n,2,3]
    cov2d = J @ cov_c @ J.transpose(-1, -2)
    m2d = torch.stack([fx * x * rz + cx, fy * y * rz + cy], dim=-1)

    a = cov2d[:, 0, 0] + eps2d
    b = 0.5 * (cov2d[:, 0, 1] + cov2d[:, 1, 0])
    c = cov2d[:, 1, 1] + eps2d
    det = a * c - b * b

    with torch.no_grad():
        mid = 0.5 * (a + c)
        v1 = mid + torch.sqrt(torch.clamp(mid * mid - det, min=0.01))
        radius = torch.ceil(3.0 * torch.sqrt(v1))
        keep = (det > 0) & (radius > radius_clip)
        keep &= ~(
            (m2d[:, 0] + radius <= 0)
            | (m2d[:, 0] - radius >= width)
            | (m2d[:, 1] + radius <= 0)
            | (m2d[:, 1] - radius >= height)
        )
    kidx = keep.nonzero()[:, 0]
    gidx = idx[kidx]
    inv_det = 1.0 / det[kidx]
    conic_k = torch.stack([c[kidx] * inv_det, -b[kidx] * inv_det, a[kidx] * inv_det], dim=-1)

    radii = torch.zeros(N, dtype=torch.int32, device=dev)
    radii[gidx] = radius[kidx].to(torch.int32)
    means2d = torch.zeros(N, 2, dtype=dt, device=dev).index_put((gidx,), m2d[kidx])
    depths = torch.zeros(N, dtype=dt, device=dev).index_put((gidx,), z[kidx])
    conics = torch.zeros(N, 3, dtype=dt, device=dev).index_put((gidx,), conic_k)
    return radii, means2d, depths, conics


def tile_rect(means2d: torch.Tensor, radii: torch.Tensor, tile_w: int, tile_h: int):
    """isect_tiles: tile_min inclusive, tile_max exclusive; evaluated in float32 like gsplat."""
    m = means2d.detach().to(torch.float32)
    r = radii.to(torch.float32)
    tx, ty, tr = m[:, 0] / TILE, m[:, 1] / TILE, r / TILE
    x0 = torch.floor(tx - tr).clamp(0, tile_w).to(torch.int64)
    y0 = torch.floor(ty - tr).clamp(0, tile_h).to(torch.int64)
    x1 = torch.ceil(tx + tr).clamp(0, tile_w).to(torch.int64)
    y1 = torch.ceil(ty + tr).clamp(0, tile_h).to(torch.int64)
    vis = radii > 0
    zero = torch.zeros_like(x0)
    return (
        torch.where(vis, x0, zero),
        torch.where(vis, y0, zero),
        torch.where(vis, x1, zero),
        torch.where(vis, y1, zero),
    )


def isect_tiles(means2d, radii, depths, width: int, height: int):
    """-> tiles_per_gauss [N], flatten_ids [n_isect] (sorted by (tile, depth-bits), stable),
    isect_offsets [tile_h*tile_w + 1] (start of each tile's run; last = n_isect)."""
    tile_w = math.ceil(width / TILE)
    tile_h = math.ceil(height / TILE)
    x0, y0, x1, y1 = tile_rect(means2d, radii, tile_w, tile_h)
    nx, ny = x1 - x0, y1 - y0
    cnt = nx * ny
    N = means2d.shape[0]
    gid = torch.repeat_interleave(torch.arange(N, device=means2d.device), cnt)
    first = torch.cumsum(cnt, 0) - cnt
    k = torch.arange(gid.shape[0], device=means2d.device) - first[gid]
    tyy = y0[gid] + k // nx[gid].clamp(min=1)
    txx = x0[gid] + k % nx[gid].clamp(min=1)
    tile_id = tyy * tile_w + txx
    dbits = depths.detach().to(torch.float32).view(torch.int32).to(torch.int64)
    key = (tile_id << 32) | dbits[gid]
    order = torch.sort(key, stable=True)[1]
    flatten_ids = gid[order]
    skey = key[order] >> 32
    offsets = torch.searchsorted(skey, torch.arange(tile_w * tile_h + 1, device=key.device))
    return cnt, flatten_ids, offsets


def rasterize_to_pixels(
    means2d: torch.Tensor,  # [N,2]
    conics: torch.Tensor,  # [N,3]
    colors: torch.Tensor,  # [N,D]
    opacities: torch.Tensor,  # [N]
    width: int,
    height: int,
    flatten_ids: torch.Tensor,
    isect_offsets: torch.Tensor,
    background: torch.Tensor | None = None,  # [D]
):
    """-> render_colors [H,W,D], render_alphas [H,W,1], last_ids int64 [H,W] (index into flatten_ids)."""
    dt, dev = means2d.dtype, means2d.device
    D = colors.shape[-1]
    tile_w = math.ceil(width / TILE)
    tile_h = math.ceil(height / TILE)
    out = torch.zeros(height, width, D, dtype=dt, device=dev)
    Tfin = torch.ones(height, width, dtype=dt, device=dev)
    last = torch.zeros(height, width, dtype=torch.int64, device=dev)
    offs = isect_offsets.tolist()
    for ty in range(tile_h):
        ys0, ys1 = ty * TILE, min((ty + 1) * TILE, height)
        for tx in range(tile_w):
            t = ty * tile_w + tx
            s, e = offs[t], offs[t + 1]
            if e <= s:
                continue
            xs0, xs1 = tx * TILE, min((tx + 1) * TILE, width)
            g = flatten_ids[s:e]
            py, px = torch.meshgrid(
                torch.arange(ys0, ys1, device=dev, dtype=dt) + 0.5,
                torch.arange(xs0, xs1, device=dev, dtype=dt) + 0.5,
                indexing="ij",
            )
            dx = means2d[g, 0][None, :] - px.reshape(-1, 1)  # [P,n]
            dy = means2d[g, 1][None, :] - py.reshape(-1, 1)
            cn = conics[g]
            sigma = 0.5 * (cn[:, 0] * dx * dx + cn[:, 2] * dy * dy) + cn[:, 1] * dx * dy
            alpha = torch.clamp(opacities[g][None, :] * torch.exp(-sigma), max=ALPHA_MAX)
            with torch.no_grad():
                skip = (sigma < 0) | (alpha < ALPHA_MIN)
            alpha = torch.where(skip, torch.zeros_like(alpha), alpha)
            T_after = torch.cumprod(1.0 - alpha, dim=1)
            T_before = torch.cat([torch.ones_like(T_after[:, :1]), T_after[:, :-1]], dim=1)
            with torch.no_grad():
                # the loop `break`s at the first splat whose next_T <= 1e-4; T_after is monotone
                live = (T_after > T_STOP) & ~skip
                alive_any = T_after > T_STOP
            w = torch.where(live, alpha * T_before, torch.zeros_like(alpha))  # [P,n]
            col = w @ colors[g]  # [P,D]
            # final T = T after the last included splat = product over included (1-alpha)
            Tf = torch.prod(torch.where(alive_any, 1.0 - alpha, torch.ones_like(alpha)), dim=1)
            with torch.no_grad():
                ar = torch.arange(e - s, device=dev)[None, :].expand_as(live)
                li = torch.where(live, ar, torch.full_like(ar, -1)).max(dim=1)[0]
                li = torch.where(li >= 0, li + s, torch.zeros_like(li))
            hh, ww = ys1 - ys0, xs1 - xs0
            out[ys0:ys1, xs0:xs1] = col.reshape(hh, ww, D)
            Tfin[ys0:ys1, xs0:xs1] = Tf.reshape(hh, ww)
            last[ys0:ys1, xs0:xs1] = li.reshape(hh, ww)
    if background is not None:
        out = out + Tfin[..., None] * background
    return out, (1.0 - Tfin)[..., None], last


def rasterization(
    means,
    quats,
    scales,
    opacities,
    colors,
    viewmat,
    K,
    width: int,
    height: int,
    background=None,
    render_mode: str = "RGB",
    near_plane: float = 0.01,
    far_plane: float = 1e10,
    eps2d: float = 0.3,
    radius_clip: float = 0.0,
):
    """Single-camera restatement of `gsplat.rendering.rasterization(packed=False)`.
    -> render_colors [H,W,D(+1)], render_alphas [H,W,1], info dict."""
    assert render_mode in ("RGB", "D", "ED", "RGB+D", "RGB+ED")
    radii, means2d, depths, conics = project(
        means, quats, scales, viewmat, K, width, height, near_plane, far_plane, eps2d, radius_clip
    )
    if render_mode in ("RGB+D", "RGB+ED"):
        colors = torch.cat([colors, depths[:, None]], dim=-1)
        if background is not None:
            background = torch.cat([background, torch.zeros_like(background[:1])], dim=-1)
    elif render_mode in ("D", "ED"):
        colors = depths[:, None]
        if background is not None:
            background = torch.zeros_like(background[:1])
    tiles_per_gauss, flatten_ids, isect_offsets = isect_tiles(means2d, radii, depths, width, height)
    rc, ra, last_ids = rasterize_to_pixels(
        means2d, conics, colors, opacities, width, height, flatten_ids, isect_offsets, background
    )
    if render_mode in ("ED", "RGB+ED"):
        rc = torch.cat([rc[..., :-1], rc[..., -1:] / ra.clamp(min=1e-10)], dim=-1)
    info = {
        "radii": radii,
        "means2d": means2d,
        "depths": depths,
        "conics": conics,
        "tiles_per_gauss": tiles_per_gauss,
        "flatten_ids": flatten_ids,
        "isect_offsets": isect_offsets,
        "last_ids": last_ids,
        "n_isect": int(flatten_ids.shape[0]),
    }
    return rc, ra, info
