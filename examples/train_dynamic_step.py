#!/usr/bin/env python3
"""A stage-2 style training step on the drop-in scene model, on synthetic data.

Mirrors the SHAPE of the reference's `Trainer.train_step` (flow3d/trainer.py:203-274) - three render groups per step
(static `bg_only` blurry frame, dynamic full blurry frame with mask / track / depth channels, static `mid` frame),
the reference's photometric loss (0.8 L1 + 0.2 (1 - SSIM), fused), one Adam optimizer per parameter tensor, the
densification statistics of `_prepare_control_step` and (with --control-every) the densify / cull control steps -
without the reference's data pipeline or its PWC-Net / depth / track losses (out of scope).  It exists to show the seam in a real autograd + optimizer loop:

    python examples/train_dynamic_step.py --steps 20
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from deblur4dgs_amd import engine  # noqa: E402
from deblur4dgs_amd.control import ControlCfg, accumulate_from_model, cull_step, densify_step, spatial_order_step  # noqa: E402
from deblur4dgs_amd.losses import photometric_loss  # noqa: E402
from deblur4dgs_amd.scene_model import GaussianParams, MotionBases, SceneModel  # noqa: E402
from deblur4dgs_amd.synth import make_scene  # noqa: E402


def build(n_fg=40_000, n_bg=100_000, K=20, W=512, H=288, dev="cuda:0", seed=0):
    """The reference's default scene size (run_training_dynamic.py:118-120): 40 k fg + 100 k bg, 20 bases."""
    sc = make_scene(n_fg + n_bg, n_fg, K, 1, W, H, seed=seed)
    keys = ("means", "quats", "scales", "colors", "opacities")
    fg = GaussianParams(*[sc[k][:n_fg].clone() for k in keys], motion_coefs=sc["motion_coefs"].clone())
    bg = GaussianParams(*[sc[k][n_fg:].clone() for k in keys])
    model = SceneModel(sc["K"][None].clone(), sc["viewmat"][None].clone(), fg, MotionBases(sc["rots"], sc["transls"]), bg)
    return model.to(dev), sc


def train(steps=20, dev="cuda:0", W=512, H=288, verbose=True, control_every=0, fused_stats=True, deferred=True,
          graph=False, **kw):
    """fused_stats: the densification statistics come out of the rasterizer's backward (attach_control_stats) instead
    of a pass over `_current_xys[i].grad`; deferred: no render waits for its intersection count on the host
    (`deferred_size_check`), the counts are verified once per step; graph: the three renders, the loss and the whole
    backward of a step are captured ONCE in a HIP graph (after two eager warm-up steps) and replayed - the step is then
    one hipGraphLaunch plus the optimizers (re-captured whenever a control step changes N)."""
    assert not graph or (fused_stats and deferred), "graph capture needs the sync-free step"
    model, sc = build(W=W, H=H, dev=dev, **kw)
    model.deferred_size_check = bool(deferred)
    w2c, K = sc["viewmat"][None].to(dev), sc["K"][None].to(dev)
    # targets: renders of a perturbed copy of the scene (so the loss has something to fit)
    with torch.no_grad():
        tgt_model, _ = build(W=W, H=H, dev=dev, seed=1, **kw)
        tgt_dyn = tgt_model.render(3, w2c, K, (W, H), mode="blury")["img"]
        tgt_sta = tgt_model.render(3, w2c, K, (W, H), bg_only=True, mode="blury")["img"]
    # (torch's fused Adam faults the GPU when its gradients live in a CUDA-graph memory pool - scripts/graph_bisect.py;
    # the graph mode therefore uses the plain implementation)
    adam = lambda p, lr: torch.optim.Adam([p], lr=lr, fused=p.is_cuda and not graph)
    lrs = {"means": 1.6e-4, "colors": 1e-2, "opacities": 1e-2, "scales": 5e-3, "quats": 5e-3, "motion_coefs": 5e-3}
    # one Adam per tensor, keyed like the reference's Trainer.optimizers (trainer.py:1168-1196): the control steps
    # re-key them when rows are added / removed
    optimizers = {f"{part}.params.{n}": adam(p, lrs[n]) for part in ("fg", "bg") for n, p in getattr(model, part).params.items()}
    others = [adam(p, 1.6e-4) for p in model.motion_bases.parameters()] + [adam(p, 5e-4) for p in model.move_model.parameters()]
    opts = lambda: list(optimizers.values()) + others
    cfg = ControlCfg()
    N = model.num_gaussians
    stats = {"xys_grad_norm_acc": torch.zeros(N, device=dev), "vis_count": torch.zeros(N, dtype=torch.int64, device=dev),
             "max_radii": torch.zeros(N, device=dev)}
    target_ts = torch.tensor([1.0, 2.0, 4.0, 5.0], device=dev)
    target_w2cs = w2c.expand(4, 4, 4).contiguous()
    losses = []
    warm = min(5, steps // 2)  # lazy initialisation (Adam state, code objects) stays out of the timing
    def fwd_bwd():
        out1 = model.render(3, w2c, K, (W, H), bg_only=True, return_depth=True, return_mask=True, mode="blury")
        if fused_stats:  # statistics of the dynamic render, accumulated by its own backward (trainer.py:953-990)
            model.attach_control_stats(stats, batch_size=1)
        out2 = model.render(3, w2c, K, (W, H), target_ts=target_ts, target_w2cs=target_w2cs, return_depth=True,
                            return_mask=True, mode="blury")  # 17 channels
        model.detach_control_stats()
        side = (model._current_xys, model._current_radii, model._current_img_wh)
        out3 = model.render(3, w2c, K, (W, H), bg_only=True, return_depth=True, mode="mid")
        # the reference's photometric term, 0.8 L1 + 0.2 (1 - SSIM) (trainer.py:388-392,575-586), fused
        loss = photometric_loss(out1["img"], tgt_sta) + photometric_loss(out2["img"], tgt_dyn) + \
            0.1 * photometric_loss(out3["img"], tgt_sta) + 1e-3 * out2["tracks_3d"].square().mean()
        loss.backward()
        return loss.detach(), side

    params = [p for o in opts() for g in o.param_groups for p in g["params"]]
    captured = None  # (graph, static loss, engine.GraphWatch)
    eager_since_capture = 0
    for it in range(steps):
        if it == warm:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        if captured is not None:
            # a replayed graph keeps the list capacities of its capture: the counts of the PREVIOUS replay (copied to pinned
            # memory by nodes of the graph itself) are looked at before the next one - an overflowed replay rendered nothing,
            # so its step is void; go back to eager steps, which size the lists from the new counts, and capture again
            try:
                captured[2].check()
            except RuntimeError as e:
                if verbose:
                    print(f"step {it:3d}  {e}")
                captured, eager_since_capture = None, 0
        if graph and captured is None and eager_since_capture >= 2:
            for p_ in params:
                p_.grad = None
            g_, watch = torch.cuda.CUDAGraph(), engine.GraphWatch()
            with watch.capturing(), torch.cuda.graph(g_):
                loss_static, _ = fwd_bwd()
            captured = (g_, loss_static, watch)
        if captured is not None:
            captured[0].replay()  # gradients land in the same .grad tensors every step
            captured[2].replayed()
            loss, side = captured[1], None
        else:
            for o in opts():
                o.zero_grad(set_to_none=True)
            loss, side = fwd_bwd()
            eager_since_capture += 1
        xys2, radii2, wh2 = side if side is not None else (None, None, None)
        for o in opts():
            o.step()
        if not fused_stats:
            model._current_xys, model._current_radii, model._current_img_wh = xys2, radii2, wh2
            accumulate_from_model(stats, model, batch_size=1)
        if deferred:
            engine.check_deferred()  # raises if a render of this step overflowed its intersection lists
        losses.append(loss.detach().clone())  # no host sync inside the loop
        if control_every and it > 0 and it % control_every == 0:  # adaptive control: N changes between steps
            n_split, n_dup = densify_step(model, stats, optimizers, cfg, global_step=it)
            n_cull = cull_step(model, stats, optimizers, cfg, global_step=it)
            spatial_order_step(model, stats, optimizers)  # the control step rewrites every row anyway: leave them in Morton order of
            #                                                the first camera's image plane (binning / gather locality)
            for v in stats.values():
                v.zero_()
            params = [p for o in opts() for g in o.param_groups for p in g["params"]]
            captured, eager_since_capture = None, 0  # N changed: new shapes, new parameters -> capture again
            if verbose:
                print(f"step {it:3d}  control: split {n_split}, dup {n_dup}, cull {n_cull} -> {model.num_gaussians} Gaussians")
        if verbose and (it % 10 == 0 or it == steps - 1):
            print(f"step {it:3d}  loss {float(losses[-1]):.5f}")
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (steps - warm)
    losses = [float(l) for l in losses]
    if verbose:
        print(f"{1e3 * dt:.2f} ms / step  (3 render groups: 11 + 11 + 1 sub-samples, {model.num_gaussians} Gaussians, fwd + bwd + Adam)")
        if os.environ.get("D4GS_EXAMPLE_STATS"):  # what the engine measured per render shape: live-row fraction, list capacity
            for key, f in engine._LIVE_FRAC.items():
                print("  shape", key, "live fraction %.3f" % f, "capacity (rectangles / exact tiles)",
                      [(engine._guess_get(key + (xt,)) or (None,))[0] for xt in (False, True)])
        print(f"visible-instance count accumulated: {int(stats['vis_count'].sum())}")
    return losses, stats, dt


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--control-every", type=int, default=0, help="densify + cull every N steps (0: never)")
    ap.add_argument("--round1", action="store_true", help="statistics as a separate pass, host waits for every list size")
    ap.add_argument("--graph", action="store_true", help="replay the step's renders + loss + backward from a HIP graph")
    a = ap.parse_args()
    train(a.steps, control_every=a.control_every, fused_stats=not a.round1, deferred=not a.round1, graph=a.graph)
