"""CPU twin of the fused exposure render ("CPU twins (`*_cpu`, host pointers, no stream) for config 1").

`render_exposure_cpu` mirrors `deblur4dgs_amd.exposure.render_exposure` on CPU tensors through d4gs_forward_cpu /
d4gs_backward_cpu (csrc/cpu_twin.hip: scalar fp32, one thread, inside libd4gs.so).  It is a SEPARATE entry point for
BASELINE config 1 - the reference's CPU-runnable plumbing case - and is never a fallback: `render_exposure`,
`rasterization` and `SceneModel.render` still refuse CPU tensors, and this module raises for device tensors.
Arithmetic rules: flow3d/params.py:39-43,142-180, flow3d/transforms.py:41-53, flow3d/scene_model.py:67-120,352-397,
gsplat 1.1.1 `rasterization(packed=False)`.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L


def _c(t):
    return None if t is None else t.detach().to(torch.float32).contiguous()


class FrameCpuFn(torch.autograd.Function):
    """One autograd node over d4gs_forward_cpu / d4gs_backward_cpu.  Differentiable outputs: blended, acc, renders, alphas."""

    @staticmethod
    def forward(ctx, meta, means, quats, scales, opacities, colors, motion_coefs, rots, transls, times, RTs, viewmat, Kmat,
                background):
        dims, policy, blend = meta
        N, S, H, W = dims.N, dims.S, dims.height, dims.width
        NCH = dims.D + (dims.depth_mode != L.DEPTH_NONE)
        leaves = dict(means=_c(means), quats=_c(quats), scales=_c(scales), opacities=_c(opacities), colors=_c(colors),
                      motion_coefs=_c(motion_coefs), rots=_c(rots), transls=_c(transls), times=_c(times), RTs=_c(RTs),
                      viewmat=_c(viewmat), Kmat=_c(Kmat))
        io = dict(renders=torch.empty(S, H, W, NCH), alphas=torch.empty(S, H, W), means2d=torch.empty(S, N, 2),
                  radii=torch.empty(S, N, dtype=torch.int32), n_isect=torch.zeros(4, dtype=torch.int64),
                  background=_c(background))
        if blend:
            io.update(blended=torch.empty(H, W, NCH), acc=torch.empty(H, W))
        pol = (C.c_int32 * NCH)(*policy) if policy is not None else None
        pin = L.fill(L.ProjIn(), **leaves)
        fio = L.fill(L.FrameIO(), **io)
        fio.policy = pol
        L.check(L.lib().d4gs_forward_cpu(C.byref(dims), C.byref(pin), C.byref(fio)), "d4gs_forward_cpu")
        ctx.keep = (dims, leaves, io, pol, policy)
        ctx.shapes = {k: (None if v is None else v.shape) for k, v in
                      dict(means=means, quats=quats, scales=scales, opacities=opacities, colors=colors, motion_coefs=motion_coefs,
                           rots=rots, transls=transls, times=times, RTs=RTs, viewmat=viewmat).items()}
        ctx.mark_non_differentiable(io["means2d"], io["radii"], io["n_isect"])
        return (io.get("blended"), io.get("acc"), io["renders"], io["alphas"][..., None], io["means2d"], io["radii"],
                io["n_isect"])

    @staticmethod
    def backward(ctx, v_blended, v_acc, v_renders, v_alphas, *_):
        dims, leaves, io, pol, policy = ctx.keep
        N, S, G, K, T, D = dims.N, dims.S, dims.G, dims.K, dims.T, dims.D
        z = torch.zeros
        g = dict(v_means=z(N, 3), v_quats=z(N, 4), v_scales=z(N, 3), v_opacities=z(N), v_colors=z(N, max(D, 1)),
                 v_motion_coefs=z(G, K) if G else None, v_rots=z(K, T, 6) if G else None, v_transls=z(K, T, 3) if G else None,
                 v_times=z(S) if G else None, v_RTs=z(S, 3, 4) if leaves["RTs"] is not None else None, v_viewmat=z(4, 4))
        v_means2d = z(S, N, 2)
        keep = [_c(v_blended), _c(v_acc), _c(v_renders), None if v_alphas is None else _c(v_alphas[..., 0])]
        fg = L.fill(L.FrameGrads(), v_blended=keep[0] if io.get("blended") is not None else None,
                    v_acc=keep[1] if io.get("acc") is not None else None,
                    v_renders=keep[2] if (keep[2] is not None or io.get("blended") is not None) else z(io["renders"].shape),
                    v_alphas=keep[3], v_means2d=v_means2d)
        lg = L.fill(L.LeafGrads(), **g)
        pin = L.fill(L.ProjIn(), **leaves)
        fio = L.fill(L.FrameIO(), **io)
        fio.policy = pol
        L.check(L.lib().d4gs_backward_cpu(C.byref(dims), C.byref(pin), C.byref(fio), C.byref(fg), C.byref(lg)),
                "d4gs_backward_cpu")
        ctx.v_means2d = v_means2d
        sh = ctx.shapes
        out = lambda name, key: None if sh[name] is None or g[key] is None else g[key][..., :D].reshape(sh[name]) if name == "colors" \
            else g[key].reshape(sh[name])
        return (None, out("means", "v_means"), out("quats", "v_quats"), out("scales", "v_scales"), out("opacities", "v_opacities"),
                out("colors", "v_colors"), out("motion_coefs", "v_motion_coefs"), out("rots", "v_rots"), out("transls", "v_transls"),
                out("times", "v_times"), out("RTs", "v_RTs"), out("viewmat", "v_viewmat"), None, None)


def render_exposure_cpu(means, quats, scales, opacities, colors, n_sigmoid, motion_coefs, rots, transls, times, RTs, w2c, Kmat,
                        width, height, background=None, return_depth=False, policy=None, blend=True, raw_params=True,
                        exact_cull=True):
    """`render_exposure` (deblur4dgs_amd/exposure.py) on CPU tensors -> dict(renders [S,H,W,D'], alphas [S,H,W,1],
    blended [H,W,D'] | None, acc [H,W] | None, means2d [S,N,2], radii [S,N], n_isect int64[4])."""
    for t in (means, quats, scales, opacities, colors, w2c, Kmat):
        if t.is_cuda:
            raise RuntimeError("render_exposure_cpu takes CPU tensors (the device path is deblur4dgs_amd.exposure.render_exposure)")
    N = means.shape[0]
    G = 0 if motion_coefs is None else motion_coefs.shape[0]
    S = 1 if times is None else times.shape[0]
    if RTs is not None:
        S = RTs.shape[0]
    flags = (L.RAW_PARAMS if raw_params else 0) | (L.RAW_COLORS if n_sigmoid > 0 else 0) | (L.EXACT_CULL if exact_cull else 0)
    dims = L.Dims(N=N, G=G, K=0 if G == 0 else rots.shape[0], T=0 if G == 0 else rots.shape[1], S=S, D=colors.shape[-1],
                  width=width, height=height, depth_mode=L.DEPTH_ED if return_depth else L.DEPTH_NONE, flags=flags,
                  n_sigmoid=n_sigmoid, near_plane=0.01, far_plane=1e10, eps2d=0.3, radius_clip=0.0)
    if blend and policy is None:  # the device path's default: channel 3 <- max_S, channel 16 <- min_S (scene_model.py:392-393)
        from .exposure import reference_policy

        policy = reference_policy(dims.D + (1 if return_depth else 0))
    out = FrameCpuFn.apply((dims, policy, blend), means, quats, scales, opacities, colors, motion_coefs, rots, transls, times, RTs,
                           w2c, Kmat, background)
    return dict(zip(("blended", "acc", "renders", "alphas", "means2d", "radii", "n_isect"), out))
