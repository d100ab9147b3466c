"""Densification statistics: fused replacement for the accumulation loop of
`Trainer._prepare_control_step` (reference flow3d/trainer.py:953-990).

`running_stats` is the reference's dict of per-Gaussian tensors (`xys_grad_norm_acc` f32, `vis_count` i64,
`max_radii` f32); they are updated in place by one HIP kernel per render instead of ~10 torch launches per
sub-sample.
"""
from __future__ import annotations

import torch

from . import _lib as L
from .engine import _stream


@torch.no_grad()
def accumulate_control_stats(running_stats: dict, xys_grad: torch.Tensor, radii: torch.Tensor, img_wh, batch_size: int,
                             update_max_radii: bool = False):
    """xys_grad [S,N,2] (= means2d.grad of the fused render), radii int32 [S,N].
    `update_max_radii=False` reproduces the reference, whose `max_radii` update is an out-of-place `index_put` whose
    result is discarded (trainer.py:987-989): the statistic never changes there."""
    S, N = radii.shape[-2], radii.shape[-1]
    acc, vis, mr = running_stats["xys_grad_norm_acc"], running_stats["vis_count"], running_stats["max_radii"]
    assert acc.dtype == torch.float32 and vis.dtype == torch.int64 and mr.dtype == torch.float32
    assert acc.shape == (N,) and vis.shape == (N,) and mr.shape == (N,)
    g = xys_grad.reshape(S, N, 2).to(torch.float32).contiguous()
    r = radii.reshape(S, N).to(torch.int32).contiguous()
    L.check(L.lib().d4gs_control_stats(S, N, L.ptr(g), L.ptr(r), int(img_wh[0]), int(img_wh[1]), int(batch_size),
                                       L.ptr(acc), L.ptr(vis), L.ptr(mr), int(update_max_radii), _stream()),
            "d4gs_control_stats")


def accumulate_from_model(running_stats: dict, model, batch_size: int, update_max_radii: bool = False):
    """Same inputs the reference reads: `model._current_xys[i].grad`, `model._current_radii[i]`, `_current_img_wh`."""
    xys = torch.cat([x.grad for x in model._current_xys], 0)
    rad = torch.cat(list(model._current_radii), 0)
    accumulate_control_stats(running_stats, xys, rad, model._current_img_wh, batch_size, update_max_radii)
