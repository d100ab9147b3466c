"""Fused exposure render: all S sub-samples of one blurry frame in one deform->project->bin->sort->composite
pass, then the exposure blend (reference flow3d/scene_model.py:323-397).

`render_exposure` is the engine under `SceneModel.render` (seam S2); it takes RAW leaf parameters (the
activations of flow3d/params.py:39-43 run inside the HIP kernels) and the host-side generator's per-sub-sample
`times` / `RTs`.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L
from .engine import FrameFn, RenderCfg, State, _stream, frame_supported, render_instances, resolve_lazy

POLICY_MEAN, POLICY_MAX, POLICY_MIN = 0, 1, 2


def reference_policy(n_channels: int) -> list[int]:
    """The reference hard-codes channel 3 <- max_S and channel 16 <- min_S (scene_model.py:392-393)."""
    p = [POLICY_MEAN] * n_channels
    if n_channels > 3:
        p[3] = POLICY_MAX
    if n_channels > 16:
        p[16] = POLICY_MIN
    return p


class BlendFn(torch.autograd.Function):
    """out[c] = mean_S, or max/min over {raw_0..raw_{S-2}, mean} (the reference's in-place quirk); acc = mean_S alpha."""

    @staticmethod
    def forward(ctx, renders, alphas, policy):
        S, H, W, Cn = renders.shape
        renders = renders.contiguous()
        alphas = alphas.contiguous()
        out = torch.empty(H, W, Cn, dtype=torch.float32, device=renders.device)
        acc = torch.empty(H, W, dtype=torch.float32, device=renders.device)
        pol = (C.c_int32 * Cn)(*policy)
        L.check(L.lib().d4gs_blend_fwd(S, H * W, Cn, pol, L.ptr(renders), L.ptr(alphas), L.ptr(out), L.ptr(acc),
                                       _stream()), "d4gs_blend_fwd")
        ctx.save_for_backward(renders, out)
        ctx.policy = list(policy)
        ctx.S = S
        return out, acc

    @staticmethod
    def backward(ctx, v_out, v_acc):
        renders, out = ctx.saved_tensors
        S, H, W, Cn = renders.shape
        v_out = torch.zeros_like(out) if v_out is None else v_out.to(torch.float32).contiguous()
        v_acc = None if v_acc is None else v_acc.to(torch.float32).contiguous()
        v_r = torch.empty_like(renders)
        v_a = torch.empty(S, H, W, dtype=torch.float32, device=renders.device)
        pol = (C.c_int32 * Cn)(*ctx.policy)
        L.check(L.lib().d4gs_blend_bwd(S, H * W, Cn, pol, L.ptr(renders), L.ptr(out), L.ptr(v_out), L.ptr(v_acc),
                                       L.ptr(v_r), L.ptr(v_a), _stream()), "d4gs_blend_bwd")
        return v_r, v_a, None


def render_exposure(
    means: torch.Tensor,  # [N,3]   leaf
    quats: torch.Tensor,  # [N,4]   raw leaf (wxyz)
    scales: torch.Tensor,  # [N,3]  raw leaf (log)
    opacities: torch.Tensor,  # [N] raw leaf (logit)
    colors: torch.Tensor,  # [N,D]  first `n_sigmoid` channels raw (logit), the rest used as given
    n_sigmoid: int,
    motion_coefs: torch.Tensor | None,  # [G,K] raw leaf; the FIRST G Gaussians are dynamic
    rots: torch.Tensor | None,  # [K,T,6]
    transls: torch.Tensor | None,  # [K,T,3]
    times: torch.Tensor | None,  # [S]
    RTs: torch.Tensor | None,  # [S,3,4] camera deltas (None = identity)
    w2c: torch.Tensor,  # [4,4]
    Kmat: torch.Tensor,  # [3,3]
    width: int,
    height: int,
    background: torch.Tensor | None = None,  # [D]
    return_depth: bool = False,
    policy: list[int] | None = None,
    blend: bool = True,
    raw_params: bool = True,
    exact_cull: bool = True,
    grad_arena: dict | None = None,
    control_stats: dict | None = None,
    deferred_size_check: bool = False,
    fused: bool = False,
    lazy_sort: bool | None = None,
    near_target: int = 0,
    exact_tiles: bool | None = None,
):
    """-> dict(renders [S,H,W,D'], alphas [S,H,W,1], blended [H,W,D'] | None, acc [H,W] | None,
               means2d [S,N,2], radii [S,N], state).
    fused=True takes the one-call path (engine.FrameFn) when the channel count needs no chunking: one autograd node, every
    output differentiable; `means2d` is then a plain tensor whose gradient lands in state.xys_sink / state.v_means2d."""
    N = means.shape[0]
    G = 0 if motion_coefs is None else motion_coefs.shape[0]
    S = 1 if times is None else times.shape[0]
    if RTs is not None:
        S = RTs.shape[0]
    flags = (L.RAW_PARAMS if raw_params else 0) | (L.RAW_COLORS if n_sigmoid > 0 else 0)
    cfg = RenderCfg(N=N, G=G, K=0 if G == 0 else rots.shape[0], T=0 if G == 0 else rots.shape[1], S=S,
                    D=colors.shape[-1], width=width, height=height,
                    depth_mode=L.DEPTH_ED if return_depth else L.DEPTH_NONE, flags=flags, n_sigmoid=n_sigmoid,
                    exact_cull=exact_cull, grad_arena=grad_arena, control_stats=control_stats,
                    deferred_size_check=deferred_size_check, lazy_sort=lazy_sort, near_target=near_target, exact_tiles=exact_tiles)
    if fused and frame_supported(cfg):
        # ONE autograd node over d4gs_forward / d4gs_backward: same kernels and bits as the staged chain below, a fraction
        # of its host work.  `means2d` is then a plain tensor; its gradient goes to st.xys_sink / st.v_means2d.
        resolve_lazy(cfg, means.device)
        st = State(cfg)
        st.want_grad = torch.is_grad_enabled()
        pol = (reference_policy(cfg.NCH) if policy is None else list(policy)) if blend else None
        bl, acc, rc, ra = FrameFn.apply(st, pol, means, quats, scales, opacities, colors, motion_coefs, rots, transls, times, RTs,
                                        w2c, Kmat, background)
        io = st.frame_io
        return dict(renders=rc, alphas=ra.unsqueeze(-1), means2d=io["means2d"], radii=io["radii"], state=st, blended=bl, acc=acc)
    rc, ra, means2d, radii, st = render_instances(cfg, means, quats, scales, opacities, colors, motion_coefs, rots,
                                                  transls, times, RTs, w2c, Kmat, background)
    out = dict(renders=rc, alphas=ra, means2d=means2d, radii=radii, state=st, blended=None, acc=None)
    if blend:
        pol = reference_policy(cfg.NCH) if policy is None else policy
        out["blended"], out["acc"] = BlendFn.apply(rc, ra.squeeze(-1), pol)  # a view both ways (select would zero-fill + copy)
    return out
