"""The photometric term of the reference's training loss, fused.

`photometric_loss(pred, gt, mask)` = 0.8 * L1 + 0.2 * (1 - SSIM) on `pred * mask` vs `gt * mask`, the expression the
reference evaluates three to four times per step (flow3d/trainer.py:388-392,575-586) with `pytorch_msssim.SSIM` and
~85 eager launches per evaluation; here: two kernels forward, one backward (`csrc/photometric.hip`).  Images are
channel-last [B,H,W,3] as the rasterizer returns them (no permute), the mask is [B,H,W] or [B,H,W,1].
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class PhotometricFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, gt, mask, w_l1, w_ssim):
        if not pred.is_cuda:
            raise RuntimeError("deblur4dgs_amd.losses runs on an MI355X (ROCm) device only; got a CPU tensor")
        B, H, W, Cc = pred.shape
        p = pred.detach().float().contiguous()
        g = gt.detach().float().contiguous()
        m = None if mask is None else mask.detach().float().reshape(B, H, W).contiguous()
        lib = L.lib()
        nb = lib.d4gs_photometric_blocks(B, H, W)
        maps = torch.empty(B, H - 10, W - 10, 3, 3, device=p.device, dtype=torch.float32)
        scratch = torch.empty(2 * nb + 3, device=p.device, dtype=torch.float32)
        stream = C.c_void_p(L.raw_stream(p.device.index))
        L.check(lib.d4gs_photometric_fwd(_p(p), _p(g), _p(m), B, H, W, Cc, w_l1, w_ssim, _p(maps), _p(scratch),
                                         _p(scratch[2 * nb:]), stream), "d4gs_photometric_fwd")
        ctx.keep = (p, g, m, maps)
        ctx.w = (float(w_l1), float(w_ssim))
        out = scratch[2 * nb:]
        return out[0].clone(), out[1].clone(), out[2].clone()

    @staticmethod
    def backward(ctx, v_loss, v_l1, v_ssim):
        p, g, m, maps = ctx.keep
        B, H, W, Cc = p.shape
        v = v_loss.detach().float().reshape(1).contiguous()
        out = torch.empty_like(p)
        stream = C.c_void_p(L.raw_stream(p.device.index))
        L.check(L.lib().d4gs_photometric_bwd(_p(p), _p(g), _p(m), _p(maps), _p(v), B, H, W, Cc, ctx.w[0], ctx.w[1], _p(out),
                                             stream), "d4gs_photometric_bwd")
        return out, None, None, None, None


def photometric_loss(pred, gt, mask=None, w_l1: float = 0.8, w_ssim: float = 0.2, return_terms: bool = False):
    """-> scalar loss (and, with return_terms, the detached L1 and SSIM values).  Differentiable w.r.t. `pred` only
    (the reference's `imgs` and masks are data)."""
    loss, l1, ssim = PhotometricFn.apply(pred, gt, mask, float(w_l1), float(w_ssim))
    return (loss, l1.detach(), ssim.detach()) if return_terms else loss
