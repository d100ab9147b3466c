"""Multi-GPU: one process per GPU, `torch.distributed` (backend "nccl" == RCCL over xGMI on ROCm).

The reference is single-process / single-GPU (zero collective call sites), so this is new design
constrained only by "the sharded result equals the single-GPU result":

* exposure sharding (BASELINE config 4): the S sub-samples of ONE blurry frame are independent given replicated
  leaf parameters (flow3d/scene_model.py:323-384); rank r renders {s : s % P == r}.  The only coupling is the blend
  (scene_model.py:386-397).  Default (`ShardedBlendFn`, any S): the blended image is REDUCED - SUM all-reduce of
  [H,W,D'+1] + MAX all-reduce of the policy channels (min packed as -x) forward (2.9 MB on cfg4), MIN
  all-reduce of the winning sub-sample backward - and equals the single-GPU blend up to fp32 summation order.
  When the caller needs the per-sub-sample stack (`need_stack=True`: the trainer's pairwise exposure losses,
  flow3d/trainer.py:599-618; needs S % P == 0) `GatherBlendFn` all-gathers the ranks' sub-sample colour images
  [S/P,H,W,D'] and alphas (23.6 MB on cfg4) and runs the same HIP blend kernels as the single-GPU path
  (`k_blend_fwd/bwd`) on the full stack: the blended image is then BITWISE the single-GPU image (same summation
  order), the max / min channels and their winners need no extra collective, and the backward keeps the slice of the
  stack gradient that belongs to the rank's own sub-samples (the loss is evaluated redundantly, so no reduction).
  Then the leaf gradients: ONE flat buffer, all-reduced in two pieces - the per-Gaussian leaves (24 MB on cfg2) as
  soon as the projection backward has written them (async on RCCL's stream, overlapping the rest of autograd: the
  MoveModel backward and the host-side launch work), the small shared leaves at the end.
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


def _shard_desc(Sl, S, s_ids, Cn, n_pixels, policy):
    import ctypes as C

    from . import _lib as L

    pol = (C.c_int32 * Cn)(*policy)
    first = s_ids[0] if s_ids else 0
    stride = (s_ids[1] - s_ids[0]) if len(s_ids) > 1 else 1
    assert all(s == first + j * stride for j, s in enumerate(s_ids)), "owned sub-samples must be an arithmetic sequence"
    return L.ShardBlend(S, Sl, first, stride, Cn, n_pixels, pol), pol


class ShardedBlendFn(torch.autograd.Function):
    """Distributed counterpart of exposure.BlendFn: `renders` holds only this rank's sub-samples, the blended image is
    REDUCED over the ranks (SUM of colours + alpha, MAX of the policy channels; MIN of the winning sub-sample backward).
    Device tensors: the arithmetic is the HIP kernels d4gs_blend_shard_* (csrc/blend.hip), torch.distributed only moves
    the three small buffers.  CPU tensors (the gloo tests of the N > 1 logic): the same steps restated in torch."""

    @staticmethod
    def forward(ctx, renders, alphas, s_ids, S, policy, group):
        # renders [S_loc,H,W,C], alphas [S_loc,H,W], s_ids: global sub-sample index of each local slice
        Sl, H, W, Cn = renders.shape
        dev = renders.device
        if renders.is_cuda:
            import ctypes as C

            from . import _lib as L
            from .engine import _stream

            renders, alphas = renders.to(torch.float32).contiguous(), alphas.to(torch.float32).contiguous()
            desc, _keep = _shard_desc(Sl, S, list(s_ids), Cn, H * W, policy)
            npol = sum(1 for p in policy if p != POLICY_MEAN)
            part = torch.empty(H, W, Cn + 1, dtype=torch.float32, device=dev)
            cand = torch.empty(H, W, max(npol, 1), dtype=torch.float32, device=dev)
            lib = L.lib()
            L.check(lib.d4gs_blend_shard_partial_fwd(C.byref(desc), L.ptr(renders), L.ptr(alphas), L.ptr(part), L.ptr(cand),
                                                     _stream()), "d4gs_blend_shard_partial_fwd")
            dist.all_reduce(part, op=dist.ReduceOp.SUM, group=group)
            if npol and S > 1:
                dist.all_reduce(cand, op=dist.ReduceOp.MAX, group=group)
            out = torch.empty(H, W, Cn, dtype=torch.float32, device=dev)
            acc = torch.empty(H, W, dtype=torch.float32, device=dev)
            L.check(lib.d4gs_blend_shard_finish_fwd(C.byref(desc), L.ptr(part), L.ptr(cand), L.ptr(out), L.ptr(acc), _stream()),
                    "d4gs_blend_shard_finish_fwd")
            ctx.save_for_backward(renders, out)
            ctx.meta = (list(s_ids), S, list(policy), group)
            ctx.npol = npol
            return out, acc
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
        if renders.is_cuda:
            import ctypes as C

            from . import _lib as L
            from .engine import _stream

            desc, _keep = _shard_desc(Sl, S, s_ids, Cn, H * W, policy)
            lib = L.lib()
            win = torch.empty(H, W, max(ctx.npol, 1), dtype=torch.int32, device=dev)
            if ctx.npol and S > 1:
                L.check(lib.d4gs_blend_shard_winner(C.byref(desc), L.ptr(renders), L.ptr(out), L.ptr(win), _stream()),
                        "d4gs_blend_shard_winner")
                dist.all_reduce(win, op=dist.ReduceOp.MIN, group=group)
            v_out = v_out.to(torch.float32).contiguous()
            v_acc = None if v_acc is None else v_acc.to(torch.float32).contiguous()
            v_r = torch.empty_like(renders)
            v_a = torch.empty(Sl, H, W, dtype=torch.float32, device=dev)
            L.check(lib.d4gs_blend_shard_bwd(C.byref(desc), L.ptr(v_out), L.ptr(v_acc), L.ptr(win), L.ptr(v_r), L.ptr(v_a),
                                             _stream()), "d4gs_blend_shard_bwd")
            return v_r, v_a, None, None, None, None
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


class GatherBlendFn(torch.autograd.Function):
    """All-gather the ranks' sub-sample images, then blend the full stack locally with `blend_fn` (exposure.BlendFn:
    the HIP kernels; the gloo CPU tests inject a torch restatement).  Needs S % world == 0."""

    @staticmethod
    def forward(ctx, renders, alphas, S, policy, group, blend_fn):
        Sl, H, W, Cn = renders.shape
        P = S // Sl
        assert Sl * P == S and P == dist.get_world_size(group), "GatherBlendFn needs S % world_size == 0"
        # two all-gathers straight into the blend kernel's input layout (colours [S,H,W,C], alphas [S,H,W]): no packing
        # copy; with one sub-sample per rank (BASELINE cfg4) the rank-major result IS the s-major stack
        full_r = torch.empty(P * Sl, H, W, Cn, dtype=renders.dtype, device=renders.device)
        full_a = torch.empty(P * Sl, H, W, dtype=alphas.dtype, device=alphas.device)
        dist.all_gather_into_tensor(full_r, renders.contiguous(), group=group)
        dist.all_gather_into_tensor(full_a, alphas.contiguous(), group=group)
        if Sl > 1:  # rank r holds s = j * P + r (owned_subsamples): [P,Sl] -> s-major
            full_r = full_r.view(P, Sl, H, W, Cn).transpose(0, 1).reshape(S, H, W, Cn)
            full_a = full_a.view(P, Sl, H, W).transpose(0, 1).reshape(S, H, W)
        with torch.enable_grad():
            st_r = full_r.requires_grad_()
            st_a = full_a.requires_grad_()
            out, acc = blend_fn(st_r, st_a, policy)
        stack = (st_r.detach(), st_a.detach())
        ctx.save_for_backward(st_r, st_a, out, acc)
        ctx.meta = (S, Sl, P, dist.get_rank(group))
        return out.detach(), acc.detach(), stack[0], stack[1]

    @staticmethod
    def backward(ctx, v_out, v_acc, v_stack_r, v_stack_a):
        st_r, st_a, out, acc = ctx.saved_tensors
        S, Sl, P, rank = ctx.meta
        outs, grads = [], []
        for o, g in ((out, v_out), (acc, v_acc)):
            if g is not None:
                outs.append(o)
                grads.append(g)
        g_r, g_a = torch.autograd.grad(outs, [st_r, st_a], grads, allow_unused=True) if outs else (None, None)
        g_r = torch.zeros_like(st_r) if g_r is None else g_r
        g_a = torch.zeros_like(st_a) if g_a is None else g_a
        if v_stack_r is not None:  # losses on the per-sub-sample images (trainer.py:599-618)
            g_r = g_r + v_stack_r
        if v_stack_a is not None:
            g_a = g_a + v_stack_a
        own = slice(rank, S, P)  # this rank's sub-samples; the loss is replicated, so their gradient is local
        return g_r[own].contiguous(), g_a[own].contiguous(), None, None, None, None


PER_GAUSSIAN = ("means", "quats", "scales", "opacities", "colors", "motion_coefs")


class FlatGradAllReduce:
    """All leaf gradients in ONE flat buffer (latency-bound sizes; see module docstring), laid out
    [per-Gaussian leaves | shared leaves].  `reduce()` = one SUM all-reduce per piece; with `arm()` before the
    backward, the per-Gaussian piece is launched asynchronously the moment autograd has accumulated the last
    per-Gaussian leaf (they are written together by d4gs_project_bwd), so it travels while the rest of the backward
    (MoveModel, host launch work) runs; `reduce()` then only waits for it and reduces the small shared piece."""

    def __init__(self, leaves: dict, group=None):
        self.names = [k for k in leaves if k in PER_GAUSSIAN] + [k for k in leaves if k not in PER_GAUSSIAN]
        self.group = group
        n = sum(leaves[k].numel() for k in self.names)
        ref = leaves[self.names[0]]
        self.flat = torch.zeros(n, dtype=ref.dtype, device=ref.device)
        self.views, off = {}, 0
        self.n_big = 0
        for k in self.names:
            m = leaves[k].numel()
            self.views[k] = self.flat[off:off + m].view_as(leaves[k])
            off += m
            if k in PER_GAUSSIAN:
                self.n_big = off
        self._work = None
        self._hooks = []
        self._pending = set()

    def _fill(self, leaves, names):
        for k in names:
            g = leaves[k].grad
            if g is None:
                self.views[k].zero_()
            elif g.data_ptr() != self.views[k].data_ptr():  # already written in place when the views were passed to
                self.views[k].copy_(g)                      # the renderer as its `grad_arena`

    def arm(self, leaves: dict):
        """Call before backward(): launch the per-Gaussian all-reduce from autograd's post-accumulate hooks."""
        self.disarm()
        big = [k for k in self.names if k in PER_GAUSSIAN]
        if not big or not hasattr(torch.Tensor, "register_post_accumulate_grad_hook"):
            return
        self._pending = set(big)

        def make(k):
            def hook(_p):
                self._pending.discard(k)
                if not self._pending and self._work is None:
                    self._fill(leaves, big)
                    self._work = dist.all_reduce(self.flat[: self.n_big], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            return hook

        self._hooks = [leaves[k].register_post_accumulate_grad_hook(make(k)) for k in big]

    def disarm(self):
        for h in self._hooks:
            h.remove()
        self._hooks, self._work, self._pending = [], None, set()

    def reduce(self, leaves: dict, average: bool = False):
        small = [k for k in self.names if k not in PER_GAUSSIAN]
        if self._work is not None:  # the big piece is already in flight
            self._fill(leaves, small)
            if self.n_big < self.flat.numel():
                dist.all_reduce(self.flat[self.n_big:], op=dist.ReduceOp.SUM, group=self.group)
            self._work.wait()
        else:
            self._fill(leaves, self.names)
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
        self.disarm()
        if average:
            self.flat /= dist.get_world_size(self.group)
        for k in self.names:
            if leaves[k].grad is None or leaves[k].grad.data_ptr() != self.views[k].data_ptr():
                leaves[k].grad = self.views[k]


def mesh_coords(world: int, rank: int, V: int, E: int) -> tuple[int, int]:
    """rank -> (view v, exposure slot e) of a V x E mesh; the E ranks of one view are adjacent (v * E .. v * E + E - 1)."""
    assert V * E == world and 0 <= rank < world, f"mesh {V}x{E} does not cover world size {world}"
    return rank // E, rank % E


def mesh_exposure_group(world: int, rank: int, V: int, E: int):
    """The exposure sub-group of this rank's view.  `dist.new_group` is collective over the WORLD: every rank creates all
    V groups, in the same order, and keeps its own."""
    v_mine, _ = mesh_coords(world, rank, V, E)
    mine = None
    for v in range(V):
        g = dist.new_group(ranks=list(range(v * E, (v + 1) * E)))
        if v == v_mine:
            mine = g
    return mine


class ShardedExposure:
    """Driver used by bench.py and the training-style callers: one step = fwd + bwd + gradient all-reduce.

    mode "exposure" (BASELINE config 4): the S sub-samples of ONE frame over all ranks.  "views": one full frame of its own
    camera view per rank (data parallel).  "mesh" (`mesh=(V, E)`, V * E == world): V views, each view's sub-samples split E
    ways - the blend collectives run inside the view's exposure sub-group (`mesh_exposure_group`), the flat gradient
    all-reduce over the world, the 1 / V of the data-parallel mean rides on the loss.  The reference's own training step is
    several independent renders behind one backward (three render groups, flow3d/trainer.py:209-231), so views x exposure is
    the decomposition of the work it actually does; "exposure" == mesh (1, world), "views" == mesh (world, 1)."""

    def __init__(self, world: int, rank: int, mode: str = "exposure", group=None, need_stack: bool = False,
                 mesh: tuple[int, int] | None = None, exposure_group=None):
        assert mode in ("exposure", "views", "mesh")
        self.world, self.rank, self.mode, self.group = world, rank, mode, group
        if mode == "exposure":
            self.V, self.E, self.v, self.e, self.group_e = 1, world, 0, rank, group
        elif mode == "views":
            self.V, self.E, self.v, self.e, self.group_e = world, 1, rank, 0, None
        else:
            assert mesh is not None, "mode 'mesh' needs mesh=(V, E)"
            self.V, self.E = mesh
            self.v, self.e = mesh_coords(world, rank, self.V, self.E)
            # (E == 1: no blend collective at all; otherwise the caller may hand in the group it made, or it is made here)
            self.group_e = None if self.E == 1 else (exposure_group if exposure_group is not None
                                                     else mesh_exposure_group(world, rank, self.V, self.E))
        # exposure sharding, forward collective: need_stack=False (default) REDUCES the blended image - SUM of [H,W,D'+1]
        # + MAX of the policy channels, 2.9 MB on cfg4 - equal to the single-GPU blend up to fp32 summation
        # order (<= 2e-6 * max|value|, tests/test_parallel_gloo.py); need_stack=True ALL-GATHERS the per-sub-sample stack
        # (23.6 MB on cfg4) and blends it with the single-GPU kernels: bitwise equal, and the stack the reference's
        # pairwise exposure losses read (flow3d/trainer.py:599-618) is present on every rank
        self.need_stack = need_stack
        self.fused = True  # renders go through the one-call autograd node (engine.FrameFn); False: the staged chain
        self.reducer = None
        self.deferred_size_check = False  # True: no render waits for its list sizes on the host (engine.RenderCfg) - needed
        #                                   to capture the step, collectives included, in a HIP graph
        self.render = None  # None: deblur4dgs_amd.exposure.render_exposure.  bench.py --dry-run puts a synthetic CPU image
        #                     function here to run the N > 1 control flow (collectives, timing, agreement) on gloo without a GPU

    def step(self, leaves: dict, Kmat, W: int, H: int, background, wimg, wacc):
        if self.render is not None:
            render_exposure = self.render
        else:
            from .exposure import render_exposure

        S = leaves["times"].shape[0]
        if self.reducer is None:
            self.reducer = FlatGradAllReduce(leaves, self.group)
        # the renderer writes this step's leaf gradients straight into the flat all-reduce buffer; a .grad that still
        # aliases it from the previous step must go first (autograd would add the buffer to itself)
        for k, v in leaves.items():
            if v.grad is not None and v.grad.data_ptr() == self.reducer.views[k].data_ptr():
                v.grad = None
        # data-parallel mean of the per-view gradients: the 1 / V factor rides on the loss, so the SUM all-reduce needs no
        # 24 MB division pass afterwards
        view_scale = 1.0 / self.V
        if self.E == 1:  # a full frame of this rank's own view; only the flat gradient all-reduce couples the ranks
            res = render_exposure(leaves["means"], leaves["quats"], leaves["scales"], leaves["opacities"],
                                  leaves["colors"], 3, leaves["motion_coefs"], leaves["rots"], leaves["transls"],
                                  leaves["times"], leaves["RTs"], leaves["viewmat"], Kmat, W, H, background=background,
                                  return_depth=True, grad_arena=self.reducer.views,
                                  deferred_size_check=self.deferred_size_check, fused=self.fused)
            loss = torch.dot(res["blended"].reshape(-1), wimg.reshape(-1)) + torch.dot(res["acc"].reshape(-1), wacc.reshape(-1))
            self.reducer.arm(leaves)
            (loss * view_scale).backward()
            self.reducer.reduce(leaves)
            return res["state"]
        own = owned_subsamples(S, self.E, self.e)
        sel = slice(self.e, S, self.E)  # == own, as a strided view (no index tensor, no host-side index build)
        res = render_exposure(leaves["means"], leaves["quats"], leaves["scales"], leaves["opacities"], leaves["colors"],
                              3, leaves["motion_coefs"], leaves["rots"], leaves["transls"],
                              leaves["times"][sel], leaves["RTs"][sel],
                              leaves["viewmat"], Kmat, W, H, background=background, return_depth=True, blend=False,
                              grad_arena=self.reducer.views, deferred_size_check=self.deferred_size_check, fused=self.fused)
        pol = reference_policy(res["renders"].shape[-1])
        if self.need_stack and S % self.E == 0:  # one all-gather, then the single-GPU blend kernels on the full stack
            from .exposure import BlendFn

            blended, acc, _stack_r, _stack_a = GatherBlendFn.apply(res["renders"], res["alphas"].squeeze(-1), S, pol,
                                                                   self.group_e, BlendFn.apply)
        else:  # reduce: SUM + MAX forward, MIN of the winning sub-sample backward (also the ragged-S path)
            blended, acc = ShardedBlendFn.apply(res["renders"], res["alphas"].squeeze(-1), own, S, pol, self.group_e)
        loss = torch.dot(blended.reshape(-1), wimg.reshape(-1)) + torch.dot(acc.reshape(-1), wacc.reshape(-1))
        self.reducer.arm(leaves)
        (loss if self.V == 1 else loss * view_scale).backward()
        self.reducer.reduce(leaves)
        return res["state"]
