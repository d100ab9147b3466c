"""Multi-GPU: one process per GPU, `torch.distributed` (backend "nccl" == RCCL over xGMI on ROCm).

The reference is single-process / single-GPU (zero collective call sites), so this is new design
constrained only by "the sharded result equals the single-GPU result":

* exposure sharding (BASELINE config 4): the S sub-samples of ONE blurry frame are independent given replicated
  leaf parameters (flow3d/scene_model.py:323-384); rank r renders {s : s % P == r}.  The only coupling is the blend
  (scene_model.py:386-397):   SUM all-reduce of [H,W,D'+1] (colours + alpha)  ->  mean,
                              MAX all-reduce of the max/min-policy channels (min packed as -x),
  and in the backward one MIN all-reduce of the winning sub-sample index for those channels, then ONE flat SUM
  all-reduce of all leaf gradients (replicated params -> data-parallel gradient sum).
* view sharding: every rank renders a full frame of its own camera view; only the flat gradient all-reduce remains.

xGMI is point-to-point (7 links x ~153 GB/s per GPU): messages here are 2.4-24 MB, i.e. latency/launch-bound, so
they are fused into as few collectives as possible (one flat buffer per reduction type).
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from .exposure import POLICY_MAX, POLICY_MEAN, POLICY_MIN, reference_policy


def owned_subsamples(S: int, world: int, rank: int) -> list[int]:
    return [s for s in range(S) if s % world == rank]


class ShardedBlendFn(torch.autograd.Function):
    """Distributed counterpart of exposure.BlendFn.  `renders` holds only this rank's sub-samples."""

    @staticmethod
    def forward(ctx, renders, alphas, s_ids, S, policy, group):
        # renders [S_loc,H,W,C], alphas [S_loc,H,W], s_ids: global sub-sample index of each local slice
        Sl, H, W, Cn = renders.shape
        dev = renders.device
        summed = torch.cat([renders.sum(0), alphas.sum(0)[..., None]], -1) if Sl > 0 else \
            torch.zeros(H, W, Cn + 1, device=dev, dtype=renders.dtype)
        dist.all_reduce(summed, op=dist.ReduceOp.SUM, group=group)
        mean = summed[..., :Cn] / S if S > 1 else summed[..., :Cn]
        acc = summed[..., Cn] / S if S > 1 else summed[..., Cn]
        pol_ch = [c for c, p in enumerate(policy) if p != POLICY_MEAN]
        out = mean.clone()
        ctx.pol_ch = pol_ch
        if pol_ch and S > 1:
            sign = torch.tensor([1.0 if policy[c] == POLICY_MAX else -1.0 for c in pol_ch], device=dev,
                                dtype=renders.dtype)
            # candidates: raw_s for owned s <= S-2 (the reference's last slot holds the mean instead of raw_{S-1})
            keep = [i for i, s in enumerate(s_ids) if s <= S - 2]
            if keep:
                loc = (renders[keep][..., pol_ch] * sign).amax(0)
            else:
                loc = torch.full((H, W, len(pol_ch)), -float("inf"), device=dev, dtype=renders.dtype)
            dist.all_reduce(loc, op=dist.ReduceOp.MAX, group=group)
            best = torch.maximum(loc, mean[..., pol_ch] * sign)
            out[..., pol_ch] = best * sign
        ctx.save_for_backward(renders, out)
        ctx.meta = (list(s_ids), S, list(policy), group)
        return out, acc

    @staticmethod
    def backward(ctx, v_out, v_acc):
        renders, out = ctx.saved_tensors
        s_ids, S, policy, group = ctx.meta
        Sl, H, W, Cn = renders.shape
        dev = renders.device
        v_out = torch.zeros_like(out) if v_out is None else v_out
        v_r = (v_out / S).expand(Sl, H, W, Cn).clone() if S > 1 else v_out.expand(Sl, H, W, Cn).clone()
        pol_ch = ctx.pol_ch
        if pol_ch and S > 1 and Sl > 0:
            # winner = lowest s <= S-2 whose raw value equals the blended value; none -> the mean receives it
            sid = torch.tensor(s_ids, device=dev).view(Sl, 1, 1, 1)
            eq = (renders[..., pol_ch] == out[..., pol_ch][None]) & (sid <= S - 2)
            cand = torch.where(eq, sid.expand_as(eq), torch.full_like(eq, S, dtype=sid.dtype)).amin(0)
            dist.all_reduce(cand, op=dist.ReduceOp.MIN, group=group)
            g = v_out[..., pol_ch]
            win = cand[None] == sid  # [Sl,H,W,n]
            has_winner = (cand < S)[None]
            v_r[..., pol_ch] = torch.where(has_winner, torch.where(win, g[None], torch.zeros_like(g)[None]),
                                           (g / S)[None].expand(Sl, H, W, len(pol_ch)))
        elif pol_ch and S > 1:
            cand = torch.full((H, W, len(pol_ch)), S, device=dev, dtype=torch.int64)
            dist.all_reduce(cand, op=dist.ReduceOp.MIN, group=group)
        v_a = None
        if v_acc is not None:
            v_a = (v_acc / S if S > 1 else v_acc).expand(Sl, H, W).clone()
        return v_r, v_a, None, None, None, None


class FlatGradAllReduce:
    """All leaf gradients in ONE flat buffer -> one SUM all-reduce (latency-bound sizes; see module docstring)."""

    def __init__(self, leaves: dict, group=None):
        self.names = list(leaves)
        self.group = group
        n = sum(leaves[k].numel() for k in self.names)
        ref = leaves[self.names[0]]
        self.flat = torch.zeros(n, dtype=ref.dtype, device=ref.device)
        self.views, off = {}, 0
        for k in self.names:
            m = leaves[k].numel()
            self.views[k] = self.flat[off:off + m].view_as(leaves[k])
            off += m

    def reduce(self, leaves: dict, average: bool = False):
        for k in self.names:
            g = leaves[k].grad
            if g is None:
                self.views[k].zero_()
            elif g.data_ptr() != self.views[k].data_ptr():  # already written in place when the views were passed to
                self.views[k].copy_(g)                      # the renderer as its `grad_arena`
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
        if average:
            self.flat /= dist.get_world_size(self.group)
        for k in self.names:
            if leaves[k].grad is None or leaves[k].grad.data_ptr() != self.views[k].data_ptr():
                leaves[k].grad = self.views[k]


class ShardedExposure:
    """Driver used by bench.py and the training-style callers: one step = fwd + bwd + gradient all-reduce."""

    def __init__(self, world: int, rank: int, mode: str = "exposure", group=None):
        assert mode in ("exposure", "views")
        self.world, self.rank, self.mode, self.group = world, rank, mode, group
        self.reducer = None

    def step(self, leaves: dict, Kmat, W: int, H: int, background, wimg, wacc):
        from .exposure import render_exposure

        S = leaves["times"].shape[0]
        if self.reducer is None:
            self.reducer = FlatGradAllReduce(leaves, self.group)
        # the renderer writes this step's leaf gradients straight into the flat all-reduce buffer; a .grad that still
        # aliases it from the previous step must go first (autograd would add the buffer to itself)
        for k, v in leaves.items():
            if v.grad is not None and v.grad.data_ptr() == self.reducer.views[k].data_ptr():
                v.grad = None
        if self.mode == "views":
            res = render_exposure(leaves["means"], leaves["quats"], leaves["scales"], leaves["opacities"],
                                  leaves["colors"], 3, leaves["motion_coefs"], leaves["rots"], leaves["transls"],
                                  leaves["times"], leaves["RTs"], leaves["viewmat"], Kmat, W, H, background=background,
                                  return_depth=True, grad_arena=self.reducer.views)
            loss = torch.dot(res["blended"].reshape(-1), wimg.reshape(-1)) + torch.dot(res["acc"].reshape(-1), wacc.reshape(-1))
            # data-parallel mean of the per-view gradients: the 1 / world factor rides on the loss, so the SUM
            # all-reduce needs no 24 MB division pass afterwards
            (loss * (1.0 / self.world)).backward()
            self.reducer.reduce(leaves)
            return res["state"]
        own = owned_subsamples(S, self.world, self.rank)
        idx = torch.tensor(own, device=leaves["times"].device, dtype=torch.long)
        res = render_exposure(leaves["means"], leaves["quats"], leaves["scales"], leaves["opacities"], leaves["colors"],
                              3, leaves["motion_coefs"], leaves["rots"], leaves["transls"],
                              leaves["times"].index_select(0, idx), leaves["RTs"].index_select(0, idx),
                              leaves["viewmat"], Kmat, W, H, background=background, return_depth=True, blend=False,
                              grad_arena=self.reducer.views)
        pol = reference_policy(res["renders"].shape[-1])
        blended, acc = ShardedBlendFn.apply(res["renders"], res["alphas"].squeeze(-1), own, S, pol, self.group)
        loss = torch.dot(blended.reshape(-1), wimg.reshape(-1)) + torch.dot(acc.reshape(-1), wacc.reshape(-1))
        loss.backward()
        self.reducer.reduce(leaves)
        return res["state"]
