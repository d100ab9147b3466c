"""Adaptive Gaussian control.

* statistics: fused replacement for the accumulation loop of `Trainer._prepare_control_step`
  (reference flow3d/trainer.py:953-990) - one HIP kernel per render;
* control steps: `densify_step`, `cull_step`, `reset_opacity_step` mirror `Trainer._densify_control_step`,
  `_cull_control_step`, `_reset_opacity_control_step` (trainer.py:992-1166) and the optimizer-state surgery of
  `dup_in_optim` / `remove_from_optim` / `reset_in_optim` (trainer.py:1199-1252).  They run every 100 steps on a few
  MB of parameters (host logic over torch index ops, as upstream); what matters for the hot path is that N changes
  between steps - every workspace of the renderer is sized per call, so nothing has to be re-queried by hand.

`running_stats` is the reference's dict of per-Gaussian tensors (`xys_grad_norm_acc` f32, `vis_count` i64,
`max_radii` f32); they are updated in place by one HIP kernel per render instead of ~10 torch launches per
sub-sample.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch

from . import _lib as L
from .engine import _stream
from .rows import RowPlan


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


# ---------------------------------------------------------------------------------------------------------------------
@dataclass
class ControlCfg:
    """The reference's adaptive-control knobs with their defaults (flow3d/configs.py:51-67)."""
    warmup_steps: int = 200
    control_every: int = 100
    reset_opacity_every_n_controls: int = 30
    stop_control_by_screen_steps: int = 4000
    stop_control_steps: int = 4000
    densify_xys_grad_threshold: float = 0.0002
    densify_scale_threshold: float = 0.01
    densify_screen_threshold: float = 0.05
    stop_densify_steps: int = 15000
    cull_opacity_threshold: float = 0.1
    cull_scale_threshold: float = 0.5
    cull_screen_threshold: float = 0.15

    @property
    def reset_opacity_every(self) -> int:
        return self.control_every * self.reset_opacity_every_n_controls


def new_running_stats(n: int, device) -> dict:
    return {"xys_grad_norm_acc": torch.zeros(n, device=device), "vis_count": torch.zeros(n, dtype=torch.int64, device=device),
            "max_radii": torch.zeros(n, device=device)}


def _swap_param(optimizer, i: int, p_new, edit_state):
    """Re-key optimizer.state from the old Parameter of group i to `p_new`, editing every per-row tensor."""
    old = optimizer.param_groups[i]["params"][0]
    state = optimizer.state.get(old, {})
    if len(state) == 0:  # the optimizer has not stepped yet: nothing to carry over (and, as upstream, the group
        return False     # keeps pointing at the old Parameter - callers re-create optimizers in that case)
    for key in list(state):
        if key == "step":
            continue
        state[key] = edit_state(state[key])
    del optimizer.state[old]
    optimizer.state[p_new] = state
    optimizer.param_groups[i]["params"] = [p_new]
    return True


def dup_in_optim(optimizer, new_params: list, plan: RowPlan):
    """Adam moments after `densify_params` (trainer.py:1199-1217): rows of the kept Gaussians, then zeros for every new
    row - the same RowPlan as the parameters, gathered with `zero_new`."""
    assert len(optimizer.param_groups) == len(new_params)
    for i, p in enumerate(new_params):
        if not _swap_param(optimizer, i, p, lambda m: plan.gather(m, zero_new=True)):
            return


def remove_from_optim(optimizer, new_params: list, plan: RowPlan):
    """trainer.py:1219-1236."""
    assert len(optimizer.param_groups) == len(new_params)
    for i, p in enumerate(new_params):
        if not _swap_param(optimizer, i, p, plan.gather):
            return


def reset_in_optim(optimizer, new_params: list):
    """Upstream zeroes every entry of the state here, `step` included."""
    assert len(optimizer.param_groups) == len(new_params)
    for i, p in enumerate(new_params):
        old = optimizer.param_groups[i]["params"][0]
        state = optimizer.state.get(old, {})
        if len(state) == 0:
            return
        for key in list(state):
            state[key] = torch.zeros_like(state[key])
        del optimizer.state[old]
        optimizer.state[p] = state
        optimizer.param_groups[i]["params"] = [p]


def _parts(model, only_fg: bool):
    return [("fg", model.fg)] + ([("bg", model.bg)] if (model.bg is not None and not only_fg) else [])


@torch.no_grad()
def densify_step(model, running_stats: dict, optimizers: dict, cfg: ControlCfg, global_step: int, only_fg: bool = False):
    """trainer.py:992-1079.  `optimizers` maps "fg.params.means" etc. to single-parameter optimizers (may be empty).
    Returns (n_split, n_dup)."""
    vis = running_stats["vis_count"]
    assert (vis > 0).any()
    grad_avg = running_stats["xys_grad_norm_acc"] / vis.clamp_min(1)
    grad_high = grad_avg > cfg.densify_xys_grad_threshold
    scale_big = model.get_scales_all().amax(-1) > cfg.densify_scale_threshold
    if global_step < cfg.stop_control_by_screen_steps:
        radius_big = running_stats["max_radii"] > cfg.densify_screen_threshold
    else:
        radius_big = torch.zeros_like(grad_high)
    split = grad_high & (scale_big | radius_big)
    dup = grad_high & ~scale_big
    nfg = model.num_fg_gaussians
    masks = {"fg": (split[:nfg], dup[:nfg]), "bg": (split[nfg:], dup[nfg:])}
    plans = {}
    for name, part in _parts(model, only_fg):  # one compaction plan per Gaussian set serves parameters, Adam moments
        sp, du = masks[name]                   # and statistics
        plans[name] = plan = RowPlan(sp, du)
        for pname, p_new in part.densify_params(sp, du, plan=plan).items():
            opt = optimizers.get(f"{name}.params.{pname}")
            if opt is not None:
                dup_in_optim(opt, [p_new], plan)
    for k, v in running_stats.items():  # same row order as densify_params: kept, duplicated, split twice
        chunks = []
        for name, lo, hi in (("fg", 0, nfg), ("bg", nfg, v.shape[0])):
            chunks.append(plans[name].gather(v[lo:hi]) if name in plans else v[lo:hi])
        running_stats[k] = torch.cat(chunks, 0)
    return sum(p.n_split for p in plans.values()), sum(p.n_dup for p in plans.values())


@torch.no_grad()
def cull_step(model, running_stats: dict, optimizers: dict, cfg: ControlCfg, global_step: int, only_fg: bool = False):
    """trainer.py:1081-1141.  Returns the number of culled Gaussians."""
    opac = model.get_opacities_all()
    cull = opac < cfg.cull_opacity_threshold
    nfg = model.num_fg_gaussians
    if global_step > cfg.reset_opacity_every:
        thr = torch.full((opac.shape[0],), cfg.cull_scale_threshold, device=opac.device)
        thr[nfg:] *= model.bg_scene_scale
        cull = cull | (model.get_scales_all().amax(-1) > thr)
        if global_step < cfg.stop_control_by_screen_steps:
            cull = cull | (running_stats["max_radii"] > cfg.cull_screen_threshold)
    masks = {"fg": cull[:nfg], "bg": cull[nfg:]}
    plans = {}
    for name, part in _parts(model, only_fg):
        plans[name] = plan = RowPlan(masks[name])
        for pname, p_new in part.cull_params(masks[name], plan=plan).items():
            opt = optimizers.get(f"{name}.params.{pname}")
            if opt is not None:
                remove_from_optim(opt, [p_new], plan)
    for k, v in running_stats.items():
        chunks = []
        for name, lo, hi in (("fg", 0, nfg), ("bg", nfg, v.shape[0])):
            chunks.append(plans[name].gather(v[lo:hi]) if name in plans else v[lo:hi])
        running_stats[k] = torch.cat(chunks, 0)
    return sum(p.n_in - p.n_keep for p in plans.values())


def morton_permutation(means: torch.Tensor, viewmat: torch.Tensor | None = None, bits: int = 10) -> torch.Tensor:
    """Permutation that sorts the rows of `means` [n, 3] along a Morton curve.  With `viewmat` [4, 4] (world -> camera): the 2-D
    curve over the image-plane position (x / z, y / z) seen from that camera - depth is deliberately NOT part of the key; without:
    the 3-D curve over the bounding box.  `bits` per axis."""
    if means.shape[0] == 0:
        return torch.zeros(0, dtype=torch.int64, device=means.device)
    if viewmat is not None:
        pc = means @ viewmat[:3, :3].T.to(means) + viewmat[:3, 3].to(means)
        z = pc[:, 2:3].clamp_min(1e-6)
        pts = pc[:, :2] / z
        pts = pts.clamp(pts.median(0).values - 4 * pts.std(0), pts.median(0).values + 4 * pts.std(0))  # far off-screen outliers
    else:
        pts = means
    lo, hi = pts.amin(0), pts.amax(0)
    q = ((pts - lo) / (hi - lo).clamp_min(1e-12) * ((1 << bits) - 1)).round().long().clamp_(0, (1 << bits) - 1)
    code = torch.zeros(means.shape[0], dtype=torch.int64, device=means.device)
    nd = pts.shape[1]
    for b in range(bits):
        for a in range(nd):
            code |= ((q[:, a] >> b) & 1) << (nd * b + a)
    return torch.argsort(code, stable=True)


@torch.no_grad()
def spatial_order_step(model, running_stats: dict | None, optimizers: dict, only_fg: bool = False, viewmat="first"):
    """Re-order the Gaussians of each set (fg, bg - the dynamic ones stay first) along a Morton curve: same scene, same image (the
    composite orders by depth, ties by index: tests/test_gpu_control.py), but Gaussians that are neighbours on screen become
    neighbours in memory - the binning pass writes longer runs per (block, tile list) and the per-instance gathers hit the L2.
    `viewmat`: the camera whose image plane carries the curve - "first" = the model's first training camera (forward-facing
    captures like the reference's stereo-blur scenes: every training view sees roughly that layout), a [4, 4] tensor, or None = the
    3-D curve over world space.  Prefer a camera: keyed on the image plane only, a run of neighbouring Gaussians mixes near (large
    footprint) and far ones, whereas the 3-D curve packs the near ones together and a wave of `k_gather` - one lane per Gaussian,
    the wave streams the rows of its 64 - then owns 64 long row spans (measured, bench.py --spatial-order: cfg5 9.63 -> 8.95 ms with
    the camera's curve, k_emit 1 124 -> 717 us and k_gather 909 -> 667; with the 3-D curve k_emit 693 but k_gather 1 021; cfg2 with 4x
    larger splats: k_gather 152 -> 1 029 us under the 3-D curve).  No reference counterpart: the reference's order is whatever
    initialisation and densification (kept, duplicated, split rows: flow3d/params.py:86-118) left; meant to ride on the control
    steps that re-write every row anyway (every `ControlCfg.control_every` steps).  Parameters, Adam moments and running statistics
    move together, like in `densify_step` / `cull_step`."""
    nfg = model.num_fg_gaussians
    if isinstance(viewmat, str):
        viewmat = model.w2cs[0] if viewmat == "first" else None
    plans = {}
    for name, part in _parts(model, only_fg):
        plans[name] = plan = RowPlan.from_permutation(morton_permutation(part.params["means"].detach(), viewmat))
        for pname, p_new in part.reorder_params(plan).items():
            opt = optimizers.get(f"{name}.params.{pname}")
            if opt is not None:
                remove_from_optim(opt, [p_new], plan)  # (a pure gather of the state rows: the same call the cull step makes)
    if running_stats is not None:
        for k, v in running_stats.items():
            chunks = []
            for name, lo, hi in (("fg", 0, nfg), ("bg", nfg, v.shape[0])):
                chunks.append(plans[name].gather(v[lo:hi]) if name in plans else v[lo:hi])
            running_stats[k] = torch.cat(chunks, 0)
    return plans


@torch.no_grad()
def reset_opacity_step(model, optimizers: dict, cfg: ControlCfg, only_fg: bool = False):
    """trainer.py:1143-1166: every opacity logit := logit(0.8 * cull threshold), Adam state zeroed."""
    new_val = torch.logit(torch.tensor(0.8 * cfg.cull_opacity_threshold))
    for name, part in _parts(model, only_fg):
        for pname, p in part.reset_opacities(new_val).items():
            opt = optimizers.get(f"{name}.params.{pname}")
            if opt is not None:
                reset_in_optim(opt, [p])
