"""ctypes binding of oracle/raster_ref.c (test infrastructure; see oracle/__init__.py)."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_libs = {}


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib(dtype=np.float32):
    name = "libraster_ref_f32.so" if dtype == np.float32 else "libraster_ref_f64.so"
    if name not in _libs:
        path = os.path.join(_HERE, name)
        if not os.path.exists(path):
            build()
        _libs[name] = ctypes.CDLL(path)
        _libs[name].ref_isect_count.restype = ctypes.c_int64
    return _libs[name]


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def rasterization(means, quats, scales, opac, colors, viewmat, K, W, H, background=None, render_mode="RGB",
                  near=0.01, far=1e10, eps2d=0.3, radius_clip=0.0, dtype=np.float32):
    """numpy in / numpy out.  Returns (render_colors [H,W,D'], alphas [H,W,1], ctx) where ctx can be
    handed to `backward`."""
    L = lib(dtype)
    R = ctypes.c_float if dtype == np.float32 else ctypes.c_double
    means, quats, scales, opac, colors = (_c(x, dtype) for x in (means, quats, scales, opac, colors))
    V, Kc = _c(viewmat, dtype), _c(K, dtype)
    N = means.shape[0]
    radii = np.zeros(N, np.int32)
    m2d = np.zeros((N, 2), dtype)
    dep = np.zeros(N, dtype)
    con = np.zeros((N, 3), dtype)
    L.ref_project_fwd(N, _p(means), _p(quats), _p(scales), _p(V), _p(Kc), W, H, R(near), R(far), R(eps2d),
                      R(radius_clip), _p(radii), _p(m2d), _p(dep), _p(con))
    ed = render_mode in ("RGB+ED", "RGB+D")
    cols = np.concatenate([colors, dep[:, None]], -1) if ed else colors
    cols = _c(cols, dtype)
    bgc = None
    if background is not None:
        bgc = _c(np.concatenate([background, [0.0]]) if ed else background, dtype)
    D = cols.shape[1]
    tpg = np.zeros(N, np.int32)
    n_isect = L.ref_isect_count(N, _p(m2d), _p(radii), W, H, _p(tpg))
    tw, th = (W + 15) // 16, (H + 15) // 16
    flat = np.zeros(max(n_isect, 1), np.int32)
    offs = np.zeros(tw * th + 1, np.int32)
    L.ref_isect_sort(N, _p(m2d), _p(radii), _p(dep), W, H, ctypes.c_int64(n_isect), _p(flat), _p(offs))
    out = np.zeros((H, W, D), dtype)
    alphas = np.zeros((H, W), dtype)
    last = np.zeros((H, W), np.int32)
    L.ref_raster_fwd(D, _p(m2d), _p(con), _p(cols), _p(opac), _p(bgc) if bgc is not None else None, W, H,
                     _p(flat), _p(offs), _p(out), _p(alphas), _p(last))
    raw = out
    if render_mode == "RGB+ED":
        out = out.copy()
        out[..., -1] = out[..., -1] / np.maximum(alphas, 1e-10)
    ctx = dict(L=L, R=R, dtype=dtype, means=means, quats=quats, scales=scales, opac=opac, cols=cols, V=V, K=Kc,
               W=W, H=H, eps2d=eps2d, radii=radii, m2d=m2d, dep=dep, con=con, flat=flat, offs=offs, raw=raw,
               alphas=alphas, last=last, bg=bgc, ed=(render_mode == "RGB+ED"), edd=ed, n_isect=int(n_isect),
               tiles_per_gauss=tpg)
    return out, alphas[..., None], ctx


def backward(ctx, v_out, v_alphas):
    """-> dict of grads wrt means, quats, scales, opac, colors (the user's D channels), viewmat, plus
    the intermediate v_means2d."""
    L, dtype, W, H = ctx["L"], ctx["dtype"], ctx["W"], ctx["H"]
    v_out = _c(v_out, dtype).copy()
    v_al = _c(v_alphas, dtype).reshape(H, W).copy()
    if ctx["ed"]:  # out_d = raw_d / max(alpha, 1e-10)
        a = ctx["alphas"]
        den = np.maximum(a, 1e-10)
        vd = v_out[..., -1]
        v_al += np.where(a > 1e-10, -vd * ctx["raw"][..., -1] / (den * den), 0.0).astype(dtype)
        v_out[..., -1] = vd / den
    N, D = ctx["cols"].shape
    v_m2d = np.zeros((N, 2), dtype)
    v_con = np.zeros((N, 3), dtype)
    v_col = np.zeros((N, D), dtype)
    v_op = np.zeros(N, dtype)
    L.ref_raster_bwd(D, _p(ctx["m2d"]), _p(ctx["con"]), _p(ctx["cols"]), _p(ctx["opac"]),
                     _p(ctx["bg"]) if ctx["bg"] is not None else None, W, H, _p(ctx["flat"]), _p(ctx["offs"]),
                     _p(ctx["alphas"]), _p(ctx["last"]), _p(v_out), _p(v_al), _p(v_m2d), _p(v_con), _p(v_col),
                     _p(v_op))
    v_dep = np.ascontiguousarray(v_col[:, -1]) if ctx["edd"] else np.zeros(N, dtype)
    v_means = np.zeros((N, 3), dtype)
    v_quats = np.zeros((N, 4), dtype)
    v_scales = np.zeros((N, 3), dtype)
    v_V = np.zeros((4, 4), dtype)
    R = ctx["R"]
    L.ref_project_bwd(N, _p(ctx["means"]), _p(ctx["quats"]), _p(ctx["scales"]), _p(ctx["V"]), _p(ctx["K"]), W, H,
                      R(ctx["eps2d"]), _p(ctx["radii"]), _p(ctx["con"]), _p(v_m2d), _p(v_dep), _p(v_con),
                      _p(v_means), _p(v_quats), _p(v_scales), _p(v_V))
    return dict(means=v_means, quats=v_quats, scales=v_scales, opac=v_op,
                colors=v_col[:, :-1] if ctx["edd"] else v_col, viewmat=v_V, means2d=v_m2d, conics=v_con)


def rasterization_torch(means, quats, scales, opacities, colors, viewmat, K, width, height, background=None,
                        render_mode="RGB", near_plane=0.01, far_plane=1e10, eps2d=0.3, radius_clip=0.0):
    """The scalar-C restatement behind ONE torch.autograd node, with oracle/raster.py `rasterization`'s signature and return
    triple (fp64): lets oracle/scene.py - the exposure loop, channel assembly and blend of `SceneModel.render`
    (flow3d/scene_model.py:162-487) - run at sizes the vectorised torch rasterizer does not finish in seconds (the
    reference's own training shape: 140 k Gaussians, 11 sub-samples, 17 channels), with gradients flowing on through torch
    autograd into the deformation and the camera generator.  tests/test_oracle_raster.py pins it to oracle/raster.py."""
    import torch

    class _Fn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, means, quats, scales, opacities, colors, viewmat):
            n = lambda t: t.detach().double().cpu().numpy()
            out, al, c = rasterization(n(means), n(quats), n(scales), n(opacities), n(colors), n(viewmat), n(K), width, height,
                                       background=None if background is None else n(background), render_mode=render_mode,
                                       near=near_plane, far=far_plane, eps2d=eps2d, radius_clip=radius_clip, dtype=np.float64)
            ctx.c = c
            seen.update(n_isect=c["n_isect"], radii=torch.from_numpy(c["radii"].copy()), means2d=torch.from_numpy(c["m2d"].copy()),
                        depths=torch.from_numpy(c["dep"].copy()), tiles_per_gauss=torch.from_numpy(c["tiles_per_gauss"].copy()),
                        conics=torch.from_numpy(c["con"].copy()), flatten_ids=torch.from_numpy(c["flat"][: c["n_isect"]].copy()).long(),
                        isect_offsets=torch.from_numpy(c["offs"].copy()).long(),
                        inputs=tuple(torch.from_numpy(c[k].copy()) for k in ("means", "quats", "scales", "opac")))
            return torch.from_numpy(np.ascontiguousarray(out)), torch.from_numpy(np.ascontiguousarray(al))

        @staticmethod
        def backward(ctx, v_out, v_al):
            g = backward(ctx.c, v_out.detach().double().numpy(), v_al.detach().double().numpy())
            t = lambda k: torch.from_numpy(np.ascontiguousarray(g[k]))
            seen["v_means2d"] = t("means2d")  # (the densification statistics' input: flow3d/scene_model.py:456-461)
            return t("means"), t("quats"), t("scales"), t("opac"), t("colors"), t("viewmat")

    seen = {}  # what the forward saw, for the caller's `info` (plain tensors: not differentiable, as in oracle/raster.py)
    rc, ra = _Fn.apply(means, quats, scales, opacities, colors, viewmat)
    return rc, ra, seen
