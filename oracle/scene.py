"""Oracle: the exposure loop of `SceneModel.render` (flow3d/scene_model.py:162-487).

Test infrastructure (see oracle/__init__.py).  Takes the host-side camera/time generator's outputs
(`RTs [S,3,4]`, `times [S]`; oracle/camera.py restates that generator) and restates:
channel assembly (:196-302), the S-loop deform -> camera delta -> rasterization (:323-384), the
exposure blend with its in-place quirk (:386-397) and the output dict (:466-487).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import deform, raster


def blend_exposure(renders: list[torch.Tensor], alphas: list[torch.Tensor], single: bool):
    """scene_model.py:386-397, literally.  `renders[-1]` is modified IN PLACE by the reference, and
    the max / min stacks are taken AFTER that write, so they see {raw_0..raw_{S-2}, avg}.
    Returns (blended, acc, exposure_stack) where exposure_stack[-1] is the blended image."""
    rs = list(renders)
    avg = rs[0] if single else torch.stack(rs, 0).mean(0)
    last = rs[-1].clone() if rs[-1].is_leaf else rs[-1] * 1.0  # fresh tensor we may write into
    rs[-1] = last
    last[..., 0 : avg.shape[-1]] = avg
    last[..., 3:4] = torch.stack(rs, 0).max(0)[0][..., 3:4]
    last[..., 16:17] = torch.stack(rs, 0).min(0)[0][..., 16:17]
    acc = torch.stack(alphas, 0).mean(0)
    return last, acc, torch.stack(rs, 0)


def render_exposure(
    fg: dict | None,  # RAW leaf params: means, quats, scales, colors, opacities, motion_coefs
    bg: dict | None,  # RAW leaf params (no motion_coefs)
    bases: dict | None,  # rots [K,T,6], transls [K,T,3]
    times: torch.Tensor,  # [S]
    RTs: torch.Tensor,  # [S,3,4]
    w2c: torch.Tensor,  # [4,4]
    K: torch.Tensor,  # [3,3]
    img_wh: tuple[int, int],
    bg_color: float | torch.Tensor = 1.0,
    return_depth: bool = False,
    return_mask: bool = False,
    target_ts: torch.Tensor | None = None,
    target_w2cs: torch.Tensor | None = None,
    single: bool = False,
    static_time: bool = False,
):
    """One blurry frame.  `single` = mode in {mid,start,end} (S must be 1).  Returns dict with the
    reference's keys plus oracle extras (`info` list per sub-sample)."""
    W, H = img_wh
    parts = [p for p in (fg, bg) if p is not None]
    dt, dev = parts[0]["means"].dtype, parts[0]["means"].device
    colors = torch.cat([deform.act_colors(p["colors"]) for p in parts], 0)
    scales = torch.cat([deform.act_scales(p["scales"]) for p in parts], 0)
    opac = torch.cat([deform.act_opacities(p["opacities"]) for p in parts], 0)
    N = colors.shape[0]
    D = colors.shape[-1]
    if not torch.is_tensor(bg_color):
        bg_color = torch.full((D,), float(bg_color), dtype=dt, device=dev)
    ds = {"img": D}
    if return_mask:  # :235-246
        mask = torch.zeros(N, 1, dtype=dt, device=dev)
        if fg is not None and bg is not None:
            mask[: fg["means"].shape[0]] = 1.0
        else:
            mask[:] = 1.0
        colors = torch.cat([colors, mask], -1)
        bg_color = torch.cat([bg_color, torch.zeros(1, dtype=dt, device=dev)])
        ds["mask"] = 1

    def poses(ts, canonical=False):
        if canonical and fg is not None:  # t is None (the viewer's canonical checkbox, flow3d/renderer.py:74-78): compute_poses_all(None)
            # = the undeformed means and the normalised quaternions (scene_model.py:84-85,103-105), background appended (:108-120)
            ps = [p for p in (fg, bg) if p is not None]
            return (torch.cat([p["means"] for p in ps], 0)[:, None].expand(-1, ts.shape[-1], -1),
                    torch.cat([deform.act_quats(p["quats"]) for p in ps], 0)[:, None].expand(-1, ts.shape[-1], -1))
        if fg is None:
            return bg["means"][:, None].expand(-1, ts.shape[-1], -1), deform.act_quats(bg["quats"])[
                :, None
            ].expand(-1, ts.shape[-1], -1)
        return deform.compute_poses_all(ts, fg, bases, bg)

    B = 0
    if target_ts is not None:  # :258-289
        B = target_ts.shape[0]
        tm, _ = poses(target_ts)  # [N,B,3]
        if target_w2cs is not None:
            tm = torch.einsum("bij,pbj->pbi", target_w2cs[:, :3], F.pad(tm, (0, 1), value=1.0))
        colors = torch.cat([colors, tm.flatten(-2)], -1)
        bg_color = torch.cat([bg_color, torch.zeros(3 * B, dtype=dt, device=dev)])
        ds["tracks_3d"] = 3 * B
    mode = "RGB"
    if return_depth:
        mode = "RGB+ED"
        ds["depth"] = 1

    renders, alphas, infos = [], [], []
    for s in range(times.shape[0]):
        m, q = poses(times[s : s + 1], canonical=static_time)  # (static_time: `time if t is not None else None`, scene_model.py:327-343)
        m, q = m[:, 0], q[:, 0]
        m = deform.camera_delta(m, RTs[s])
        rc, ra, info = raster.rasterization(
            m, q, scales, opac, colors, w2c, K, W, H, background=bg_color, render_mode=mode
        )
        renders.append(rc[None])
        alphas.append(ra[None])
        infos.append(info)
    blended, acc, stack = blend_exposure(renders, alphas, single)
    outs = torch.split(blended, list(ds.values()), dim=-1)
    out = {}
    for (name, dim), x in zip(ds.items(), outs):
        out[name] = x.reshape(1, H, W, B, 3) if name == "tracks_3d" else x
    out["acc"] = acc
    out["exposure_imgs"] = stack  # [S,1,H,W,D']
    out["pred_sharp_img"] = renders[len(renders) // 2][..., 0:3]
    out["info"] = infos
    out["raw_renders"] = renders
    return out
