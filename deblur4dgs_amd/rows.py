"""Row surgery of the adaptive-control steps as ONE plan + one gather per tensor.

`RowPlan(split, dup)` / `RowPlan(cull)` builds the source-row map of `GaussianParams.densify_params` /
`cull_params` (reference flow3d/params.py:86-118): kept rows (not split) in order, then the duplicated rows, then the
block of split rows twice; `plan.gather(x, ...)` applies it to one per-Gaussian tensor - a parameter, an Adam moment
(new rows zero: `dup_in_optim`, flow3d/trainer.py:1199-1217), a running statistic.  On a ROCm device both steps are HIP
kernels (`d4gs_control_plan`: single-block stream compaction; `d4gs_gather_rows`: one streaming launch per tensor); on
CPU tensors - the host-logic tests of the control decisions - the same plan is a handful of torch index ops.
"""
from __future__ import annotations

import ctypes as C
import math

import torch

from . import _lib as L

I64_MAX = (1 << 63) - 1


class RowPlan:
    def __init__(self, split_or_cull: torch.Tensor, dup: torch.Tensor | None = None):
        f = split_or_cull.to(torch.bool)
        self.n_in = f.shape[0]
        dev = f.device
        if f.is_cuda and self.n_in > 0:
            from .engine import _stream

            a = f.to(torch.uint8).contiguous()
            b = None if dup is None else dup.to(torch.uint8).contiguous()
            self.src = torch.empty(3 * self.n_in if dup is not None else self.n_in, dtype=torch.int32, device=dev)
            counts = torch.empty(4, dtype=torch.int32, device=dev)
            L.check(L.lib().d4gs_control_plan(self.n_in, L.ptr(a), L.ptr(b), L.ptr(self.src), L.ptr(counts), _stream()),
                    "d4gs_control_plan")
            self.n_keep, self.n_dup, self.n_split, self.n_out = counts.tolist()  # the one host sync of a control step
        else:
            keep = (~f).nonzero()[:, 0]
            parts = [keep]
            self.n_keep, self.n_dup, self.n_split = keep.shape[0], 0, 0
            if dup is not None:
                d, s = dup.to(torch.bool).nonzero()[:, 0], f.nonzero()[:, 0]
                parts += [d, s, s]
                self.n_dup, self.n_split = d.shape[0], s.shape[0]
            self.src = torch.cat(parts).to(torch.int32)
            self.n_out = self.src.shape[0]

    @classmethod
    def from_permutation(cls, perm: torch.Tensor) -> "RowPlan":
        """A plan that only REORDERS rows (out[i] = in[perm[i]]): the spatial re-ordering step (control.spatial_order_step)."""
        p = cls.__new__(cls)
        p.n_in = p.n_keep = p.n_out = perm.shape[0]
        p.n_dup = p.n_split = 0
        p.src = perm.to(torch.int32).contiguous()
        return p

    @property
    def n_new(self) -> int:
        return self.n_dup + 2 * self.n_split

    def gather(self, x: torch.Tensor, zero_new: bool = False, split_add: float | None = None) -> torch.Tensor:
        """x [n_in, ...] -> [n_out, ...]: x[src]; `zero_new`: rows of new Gaussians are 0 (fresh Adam moments);
        `split_add`: added to the rows of split halves (scales: -log 1.6)."""
        assert x.shape[0] == self.n_in
        if not x.is_cuda or self.n_in == 0:
            out = x.index_select(0, self.src[: self.n_out].long())
            if zero_new:
                out[self.n_keep:] = 0
            if split_add is not None:
                out[self.n_keep + self.n_dup:] += split_add
            return out
        from .engine import _stream

        xc = x.contiguous()
        words = xc.element_size() // 4
        assert words in (1, 2) and (split_add is None or xc.dtype == torch.float32), "32- or 64-bit rows (f32, i64)"
        w = words * (xc[0].numel() if xc.dim() > 1 else 1)
        out = torch.empty(self.n_out, *xc.shape[1:], dtype=xc.dtype, device=xc.device)
        L.check(L.lib().d4gs_gather_rows(L.ptr(self.src), self.n_out, w, L.ptr(xc), L.ptr(out),
                                         self.n_keep if zero_new else I64_MAX,
                                         self.n_keep + self.n_dup if split_add is not None else I64_MAX,
                                         C.c_float(0.0 if split_add is None else split_add), _stream()), "d4gs_gather_rows")
        return out


SPLIT_LOG_SCALE = -math.log(1.6)  # the two halves of a split get scales / 1.6 (params.py:95-97)
