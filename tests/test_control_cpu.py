"""Adaptive control, host side: parameter surgery against golden vectors produced by the reference's own
`GaussianParams` (tests/golden F6), optimizer-state surgery and the control-step decisions on a small CPU model."""
import os

import numpy as np
import torch

from deblur4dgs_amd import control
from deblur4dgs_amd.scene_model import GaussianParams, MotionBases, SceneModel

NAMES = ("means", "quats", "scales", "colors", "opacities", "motion_coefs")


def _gp(z, c):
    raw = {k: torch.tensor(z[f"c{c}_in_{k}"]) for k in NAMES if f"c{c}_in_{k}" in z.files}
    return GaussianParams(raw["means"], raw["quats"], raw["scales"], raw["colors"], raw["opacities"], raw.get("motion_coefs"))


def test_param_surgery_matches_reference_golden(golden_dir):
    z = np.load(os.path.join(golden_dir, "f6_control_params.npz"))
    for c in range(int(z["n_cases"])):
        split, dup, cull = (torch.tensor(z[f"c{c}_{m}"]) for m in ("split", "dup", "cull"))
        for op, run in (("densify", lambda g: g.densify_params(split, dup)), ("cull", lambda g: g.cull_params(cull)),
                        ("reset", lambda g: g.reset_opacities(torch.logit(torch.tensor(0.08))))):
            gp = _gp(z, c)
            out = run(gp)
            want = {k[len(f"c{c}_{op}_"):]: z[k] for k in z.files if k.startswith(f"c{c}_{op}_")}
            assert set(out) == set(want)
            for k, v in out.items():
                assert isinstance(v, torch.nn.Parameter) and gp.params[k] is v
                np.testing.assert_array_equal(v.detach().numpy(), want[k])


def _model(n_fg=30, n_bg=20, K=3, seed=0):
    g = torch.Generator().manual_seed(seed)
    mk = lambda n, coefs: GaussianParams(torch.randn(n, 3, generator=g), torch.randn(n, 4, generator=g),
                                         torch.randn(n, 3, generator=g) - 4.0, torch.randn(n, 3, generator=g),
                                         torch.randn(n, generator=g), torch.randn(n, K, generator=g) if coefs else None)
    rots = torch.tensor([1.0, 0, 0, 0, 1, 0]).repeat(K, 5, 1)
    return SceneModel(torch.eye(3)[None], torch.eye(4)[None], mk(n_fg, True), MotionBases(rots, torch.zeros(K, 5, 3)),
                      mk(n_bg, False))


def _optimizers(model, steps=2):
    opts = {}
    for part in ("fg", "bg"):
        for name, p in getattr(model, part).params.items():
            opts[f"{part}.params.{name}"] = torch.optim.Adam([p], lr=1e-3)
    for _ in range(steps):
        loss = sum((p ** 2).sum() for part in ("fg", "bg") for p in getattr(model, part).params.values())
        for o in opts.values():
            o.zero_grad()
        loss.backward()
        for o in opts.values():
            o.step()
    return opts


def test_densify_step_rows_and_adam_state():
    model = _model()
    opts = _optimizers(model)
    N, nfg = model.num_gaussians, model.num_fg_gaussians
    stats = control.new_running_stats(N, "cpu")
    stats["vis_count"] += 4
    stats["xys_grad_norm_acc"] = torch.linspace(0, 0.004, N)          # avg grad 0 .. 0.001: upper ~80 % above 2e-4
    stats["max_radii"] = torch.zeros(N)
    with torch.no_grad():
        model.fg.params["scales"][::2] = 0.0                           # exp(0) = 1 > 0.01: these split, the others dup
    cfg = control.ControlCfg()
    grad_high = stats["xys_grad_norm_acc"] / 4 > cfg.densify_xys_grad_threshold
    big = model.get_scales_all().amax(-1) > cfg.densify_scale_threshold
    split, dup = grad_high & big, grad_high & ~big
    old = {k: v.detach().clone() for k, v in model.fg.params.items()}
    m_old = opts["fg.params.means"].state[model.fg.params["means"]]["exp_avg"].clone()
    acc_old = stats["xys_grad_norm_acc"].clone()
    n_split, n_dup = control.densify_step(model, stats, opts, cfg, global_step=500)
    assert (n_split, n_dup) == (int(split.sum()), int(dup.sum())) and n_split > 0 and n_dup > 0
    sf, df = split[:nfg], dup[:nfg]
    sb, db = split[nfg:], dup[nfg:]
    assert model.num_fg_gaussians == nfg + int(sf.sum()) + int(df.sum())
    want_means = torch.cat([old["means"][~sf], old["means"][df], old["means"][sf], old["means"][sf]], 0)
    assert torch.equal(model.fg.params["means"].detach(), want_means)
    want_scales = torch.cat([old["scales"][~sf], old["scales"][df], old["scales"][sf] - np.log(1.6), old["scales"][sf] - np.log(1.6)], 0)
    assert torch.allclose(model.fg.params["scales"].detach(), want_scales)
    # optimizer: re-keyed to the new Parameter; kept rows carry their moments, new rows start at zero
    opt = opts["fg.params.means"]
    p = model.fg.params["means"]
    assert opt.param_groups[0]["params"][0] is p and set(opt.state.keys()) == {p}
    m = opt.state[p]["exp_avg"]
    n_keep = int((~sf).sum())
    assert m.shape == p.shape and torch.equal(m[:n_keep], m_old[~sf]) and (m[n_keep:] == 0).all()
    # running stats follow the same row order, fg block then bg block
    want_acc = torch.cat([acc_old[:nfg][~sf], acc_old[:nfg][df], acc_old[:nfg][sf].repeat(2),
                          acc_old[nfg:][~sb], acc_old[nfg:][db], acc_old[nfg:][sb].repeat(2)])
    assert torch.equal(stats["xys_grad_norm_acc"], want_acc) and stats["vis_count"].shape[0] == model.num_gaussians
    # the optimizer still steps on the new shapes
    (model.fg.params["means"] ** 2).sum().backward()
    opt.step()


def test_cull_and_reset_steps():
    model = _model(seed=3)
    opts = _optimizers(model)
    N, nfg = model.num_gaussians, model.num_fg_gaussians
    stats = control.new_running_stats(N, "cpu")
    stats["max_radii"] = torch.rand(N) * 0.3
    cfg = control.ControlCfg()
    opac = model.get_opacities_all()
    # before the first opacity reset only the opacity test applies (trainer.py:1101)
    want = opac < cfg.cull_opacity_threshold
    v_old = opts["bg.params.opacities"].state[model.bg.params["opacities"]]["exp_avg_sq"].clone()
    n = control.cull_step(model, stats, opts, cfg, global_step=1000)
    assert n == int(want.sum()) and model.num_gaussians == N - n and stats["max_radii"].shape[0] == N - n
    p = model.bg.params["opacities"]
    assert torch.equal(opts["bg.params.opacities"].state[p]["exp_avg_sq"], v_old[~want[nfg:]])
    # after it, scale (bg threshold times bg_scene_scale) and screen-radius tests join in
    N2, nfg2 = model.num_gaussians, model.num_fg_gaussians
    with torch.no_grad():
        model.fg.params["scales"][0] = 1.0                              # exp(1) > 0.5
    thr = torch.full((N2,), cfg.cull_scale_threshold)
    thr[nfg2:] *= model.bg_scene_scale
    want2 = (model.get_opacities_all() < cfg.cull_opacity_threshold) | (model.get_scales_all().amax(-1) > thr) | \
        (stats["max_radii"] > cfg.cull_screen_threshold)
    n2 = control.cull_step(model, stats, opts, cfg, global_step=cfg.reset_opacity_every + 1)
    assert n2 == int(want2.sum()) and n2 >= 1
    # only_fg leaves the background alone
    nbg = model.num_bg_gaussians
    control.cull_step(model, stats, opts, cfg, global_step=cfg.reset_opacity_every + 1, only_fg=True)
    assert model.num_bg_gaussians == nbg and stats["vis_count"].shape[0] == model.num_gaussians
    # reset
    control.reset_opacity_step(model, opts, cfg)
    val = float(torch.logit(torch.tensor(0.8 * cfg.cull_opacity_threshold)))
    for part in (model.fg, model.bg):
        assert torch.allclose(part.params["opacities"].detach(), torch.full_like(part.params["opacities"], val))
    st = opts["fg.params.opacities"].state[model.fg.params["opacities"]]
    assert float(st["exp_avg"].abs().sum()) == 0 and float(st["step"]) == 0


def test_optimizer_without_state_is_left_alone_like_upstream():
    model = _model(seed=5)
    opts = _optimizers(model, steps=0)
    old = model.fg.params["means"]
    stats = control.new_running_stats(model.num_gaussians, "cpu")
    stats["vis_count"] += 1
    stats["xys_grad_norm_acc"] += 1.0
    control.densify_step(model, stats, opts, control.ControlCfg(), global_step=10)
    assert opts["fg.params.means"].param_groups[0]["params"][0] is old  # trainer.py:1204-1206 returns early


def test_spatial_order_step_permutes_every_per_gaussian_tensor_together():
    """control.spatial_order_step: a pure re-ordering (3-D Morton curve of the means, per Gaussian set - dynamic ones stay first) of
    parameters, Adam moments and running statistics; neighbours in space become neighbours in memory."""
    model = _model(n_fg=200, n_bg=150)
    opts = _optimizers(model)
    N, nfg = model.num_gaussians, model.num_fg_gaussians
    stats = control.new_running_stats(N, "cpu")
    stats["xys_grad_norm_acc"] = torch.arange(N, dtype=torch.float32)  # = the old row index
    stats["vis_count"] = torch.arange(N)
    old = {part: {k: v.detach().clone() for k, v in getattr(model, part).params.items()} for part in ("fg", "bg")}
    mom = {k: o.state[o.param_groups[0]["params"][0]]["exp_avg"].clone() for k, o in opts.items()}
    plans = control.spatial_order_step(model, stats, opts)
    assert model.num_fg_gaussians == nfg and model.num_gaussians == N
    for part, off in (("fg", 0), ("bg", nfg)):
        perm = plans[part].src.long()
        n = perm.shape[0]
        assert sorted(perm.tolist()) == list(range(n))
        for k, v in getattr(model, part).params.items():
            assert torch.equal(v.detach(), old[part][k][perm]), (part, k)
            opt = opts[f"{part}.params.{k}"]
            assert opt.param_groups[0]["params"][0] is v and torch.equal(opt.state[v]["exp_avg"], mom[f"{part}.params.{k}"][perm])
        assert torch.equal(stats["xys_grad_norm_acc"][off:off + n], (perm + off).float())
        assert torch.equal(stats["vis_count"][off:off + n], perm + off)
        m_old, m_new = old[part]["means"], getattr(model, part).params["means"].detach()
        step = lambda m: (m[1:] - m[:-1]).norm(dim=-1).mean()
        assert step(m_new) < 0.85 * step(m_old)  # consecutive rows are now close (on the first camera's image plane)
    (model.fg.params["means"] ** 2).sum().backward()  # the optimizers still step
    opts["fg.params.means"].step()
