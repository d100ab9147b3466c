"""GPU parity (seam S2): `SceneModel.render` - same signature / output dict / side channels as the reference's
flow3d/scene_model.py:162-487 - against the oracle's restatement of that method (oracle/scene.py) driven by the
oracle's restatement of the host-side generator (oracle/camera.py) with the same MoveModel weights."""
import pytest
import torch

from deblur4dgs_amd.synth import make_scene
from oracle import camera as ocam
from oracle import scene as oscene
from tests.util import check, frac_bad, record, rel_err

pytestmark = pytest.mark.gpu
GRAD_FLIP_FRAC = 2e-3   # elements allowed to miss 1e-4 (a discrete alpha / T decision taken differently in fp32)
MM_TOL = 1e-4           # MoveModel grads (sums over all pixels and Gaussians): measured <= 2e-5


def _build(N, G, K, W, H, seed, dev, has_bg=True):
    from deblur4dgs_amd.scene_model import GaussianParams, MotionBases, SceneModel

    sc = make_scene(N, G, K, 1, W, H, seed=seed, dtype=torch.float32, T=8)
    sc["scales"] = sc["scales"] + 1.3
    keys = ("means", "quats", "scales", "colors", "opacities")
    fg = GaussianParams(*[sc[k][:G].clone() for k in keys], motion_coefs=sc["motion_coefs"].clone())
    bg = GaussianParams(*[sc[k][G:].clone() for k in keys]) if has_bg else None
    mb = MotionBases(sc["rots"].clone(), sc["transls"].clone())
    model = SceneModel(sc["K"][None].clone(), sc["viewmat"][None].clone(), fg, mb, bg).to(dev)
    torch.manual_seed(seed)
    with torch.no_grad():  # non-trivial camera deltas and exposure half-widths
        for head in (model.move_model.RT_head0, model.move_model.RT_head1):
            head[-1].bias.copy_(0.004 * torch.randn(6))
        model.move_model.time_params.copy_(torch.tensor([[0.5, 0.3, 0.45, 0.6, 0.2, 0.5, 0.7, 0.5]]))
    return model, sc


def _oracle(model, sc, t, W, H, mode, stage, return_depth, return_mask, target_ts, target_w2cs):
    G = model.num_fg_gaussians
    dd = lambda x: x.detach().double().cpu().clone().requires_grad_()
    fg = {k: dd(v) for k, v in model.fg.params.items()}
    bg = {k: dd(v) for k, v in model.bg.params.items()} if model.bg is not None else None
    bases = {k: dd(v) for k, v in model.motion_bases.params.items()}
    sd = {k: v.detach().cpu().double().requires_grad_() for k, v in model.move_model.state_dict().items()}
    w2c = sc["viewmat"].double()
    RTs, times, dT = ocam.forward_start_end_mid(sd, w2c[:3, :3], w2c[:3, 3:4], t, 11, stage)  # fp64, differentiable
    sel = {"mid": slice(5, 6), "start": slice(0, 1), "end": slice(10, 11)}.get(mode, slice(None))
    out = oscene.render_exposure(fg, bg, bases, times[0, sel].double(), RTs[sel].double(), w2c, sc["K"].double(), (W, H),
                                 bg_color=1.0, return_depth=return_depth, return_mask=return_mask,
                                 target_ts=target_ts, target_w2cs=target_w2cs, single=mode in ("mid", "start", "end"))
    return out, (fg, bg, bases, sd), dT


@pytest.mark.parametrize("mode,stage,t,tracks", [("blury", "second", 3.0, True), ("mid", "second", 2.0, False),
                                                 ("blury", "first", 3.0, False)])
def test_render_matches_oracle(mode, stage, t, tracks):
    dev = torch.device("cuda:0")
    N, G, K, W, H = 900, 500, 4, 64, 48
    model, sc = _build(N, G, K, W, H, 17, dev)
    tt = torch.tensor([1.0, 2.5, 4.0, 6.0]) if tracks else None
    tw = None
    if tracks:
        from oracle.camera import se3_to_SE3

        tw = torch.cat([se3_to_SE3(0.01 * torch.randn(4, 6)), torch.tensor([0, 0, 0, 1.0]).expand(4, 1, 4)], 1)
    ref, (fg, bg, bases, mm_sd), dT = _oracle(model, sc, t, W, H, mode, stage, True, True, None if tt is None else tt.double(),
                                       None if tw is None else tw.double())
    out = model.render(t, sc["viewmat"][None].to(dev), sc["K"][None].to(dev), (W, H),
                       target_ts=None if tt is None else tt.to(dev), target_w2cs=None if tw is None else tw.to(dev),
                       return_depth=True, return_mask=True, mode=mode, stage=stage)
    torch.cuda.synchronize()
    S = 1 if mode == "mid" else 11
    assert out["img"].shape == (1, H, W, 3) and out["mask"].shape == (1, H, W, 1) and out["depth"].shape == (1, H, W, 1)
    assert out["acc"].shape == (1, H, W, 1) and out["deltaT"].shape == (1, 1, 1) and out["RTs"].shape == (S, 3, 4)
    Dp = 3 + 1 + (12 if tracks else 0) + 1
    assert out["exposure_imgs"].shape == (S, 1, H, W, Dp)
    if tracks:
        assert out["tracks_3d"].shape == (1, H, W, 4, 3) and out["pred_sharp_img"].shape == (1, H, W, 3)
    for k in ("img", "mask", "depth", "acc") + (("tracks_3d",) if tracks else ()):
        assert frac_bad(out[k].cpu(), ref[k], 1e-4) <= GRAD_FLIP_FRAC, (k, rel_err(out[k].cpu(), ref[k]))
    assert frac_bad(out["exposure_imgs"].cpu(), ref["exposure_imgs"], 1e-4) <= GRAD_FLIP_FRAC
    assert abs(out["deltaT"].item() - dT.item()) < 1e-6
    assert len(model._current_xys) == S and model._current_xys[0].shape == (1, N, 2)
    assert model._current_radii[0].shape == (1, N) and model._current_radii[0].dtype == torch.int32

    # backward: image loss + the side-channel contract (means2d.grad per sub-sample, trainer.py:975)
    g = torch.Generator().manual_seed(0)
    w = torch.randn(1, H, W, 3, generator=g)
    wd = torch.randn(1, H, W, 1, generator=g)
    (out["img"] * w.to(dev)).sum().add((out["depth"] * wd.to(dev)).sum()).backward()
    ((ref["img"] * w.double()).sum() + (ref["depth"] * wd.double()).sum()).backward()
    torch.cuda.synchronize()
    for name, got_p, ref_p in (("fg.means", model.fg.params["means"], fg["means"]),
                               ("bg.scales", model.bg.params["scales"], bg["scales"]),
                               ("fg.motion_coefs", model.fg.params["motion_coefs"], fg["motion_coefs"]),
                               ("rots", model.motion_bases.params["rots"], bases["rots"]),
                               ("transls", model.motion_bases.params["transls"], bases["transls"])):
        r, b4 = record(f"S2 render mode={mode} stage={stage}", name, got_p.grad.cpu(), ref_p.grad)
        assert b4 <= GRAD_FLIP_FRAC, (name, r, b4)
    assert all(x.grad is not None and x.grad.shape == (1, N, 2) for x in model._current_xys)
    # a12 through the whole render, by value: every MoveModel parameter against fp64 autograd of the oracle chain
    # (oracle.camera generator -> oracle.scene render), not merely "non-zero"
    for name, p in model.move_model.named_parameters():
        want = mm_sd[name].grad if mm_sd[name].grad is not None else torch.zeros_like(mm_sd[name])
        got = p.grad.cpu() if p.grad is not None else torch.zeros_like(p).cpu()
        r, _ = record(f"S2 render mode={mode} stage={stage}", f"move_model.{name}", got, want)
        assert (got.double() - want).abs().max() <= MM_TOL * max(float(want.abs().max()), 1e-8), (name, r)
    if mode == "blury":
        assert model.move_model.RT_head0[-1].bias.grad.abs().sum() > 0  # camera deltas are trained through v_RTs
        assert (model.move_model.time_params.grad.abs().sum() > 0) == (stage == "second")


@pytest.mark.parametrize("has_bg,B,depth,mask", [(False, 4, True, False),   # trainer.py:506-517 without a background:
                                                  #   3 + 12 track channels + depth = 16 -> depth sits on index 15 (mean)
                                                  (True, 2, True, True), (True, 3, True, True),
                                                  (True, 6, True, True),     # 3 + 1 + 18 + depth = 23 channels: 2 chunks
                                                  (False, 7, False, False),  # 24 channels, no depth
                                                  (True, 11, True, True)])   # 3 + 1 + 33 + 1 = 38 channels: 3 chunks
def test_any_channel_layout_matches_the_reference_policy(has_bg, B, depth, mask):
    """layouts the reference accepts but the 16-channel padding used to reject or mis-blend.  The engine
    composites any channel count in chunks over one projection / one set of tile lists and the blend policy
    (channel 3 <- max, 16 <- min; scene_model.py:392-393) is evaluated on the reference's own layout."""
    dev = torch.device("cuda:0")
    N, G, K, W, H = 700, 700 if not has_bg else 400, 3, 64, 48
    model, sc = _build(N, G, K, W, H, 31 + B, dev, has_bg=has_bg)
    tt = torch.linspace(0.5, 6.5, B)
    from oracle.camera import se3_to_SE3

    g = torch.Generator().manual_seed(B)
    tw = torch.cat([se3_to_SE3(0.01 * torch.randn(B, 6, generator=g)), torch.tensor([0, 0, 0, 1.0]).expand(B, 1, 4)], 1)
    ref, (fg, bg, bases, mm_sd), dT = _oracle(model, sc, 3.0, W, H, "blury", "second", depth, mask, tt.double(), tw.double())
    out = model.render(3.0, sc["viewmat"][None].to(dev), sc["K"][None].to(dev), (W, H), target_ts=tt.to(dev),
                       target_w2cs=tw.to(dev), return_depth=depth, return_mask=mask, mode="blury", stage="second")
    Dp = 3 + int(mask) + 3 * B + int(depth)
    assert out["exposure_imgs"].shape == (11, 1, H, W, Dp) and out["tracks_3d"].shape == (1, H, W, B, 3)
    case = f"S2 layout bg={has_bg} B={B} depth={depth} mask={mask} ({Dp} channels)"
    names = ("img", "tracks_3d", "acc") + (("depth",) if depth else ()) + (("mask",) if mask else ())
    for k in names:
        check(case, k, out[k].cpu(), ref[k], 1e-4, GRAD_FLIP_FRAC)
    check(case, "exposure_imgs", out["exposure_imgs"].cpu(), ref["exposure_imgs"], 1e-4, GRAD_FLIP_FRAC)
    w = torch.randn(out["tracks_3d"].shape, generator=g)
    wi = torch.randn(1, H, W, 3, generator=g)
    ((out["tracks_3d"] * w.to(dev)).sum() + (out["img"] * wi.to(dev)).sum() + (out["depth"].sum() if depth else 0.0)).backward()
    ((ref["tracks_3d"] * w.double()).sum() + (ref["img"] * wi.double()).sum() + (ref["depth"].sum() if depth else 0.0)).backward()
    torch.cuda.synchronize()
    for name, got_p, ref_p in (("fg.means", model.fg.params["means"], fg["means"]),
                               ("fg.colors", model.fg.params["colors"], fg["colors"]),
                               ("fg.opacities", model.fg.params["opacities"], fg["opacities"]),
                               ("fg.motion_coefs", model.fg.params["motion_coefs"], fg["motion_coefs"]),
                               ("rots", model.motion_bases.params["rots"], bases["rots"]),
                               ("transls", model.motion_bases.params["transls"], bases["transls"])):
        check(case, name, got_p.grad.cpu(), ref_p.grad, 1e-4, GRAD_FLIP_FRAC)


@pytest.mark.parametrize("which,mode,use_filter", [("fg", "start", False), ("bg", "end", False), ("all", "mid", True),
                                                   ("fg", "blury", True)])
def test_render_variants_match_oracle(which, mode, use_filter):
    """The other call shapes of `render()` the reference uses (scene_model.py:196-232,298-321,355-358): fg_only / bg_only
    (validator.py:104-151, trainer.py:320), mode start / end / mid, and `filter_mask` - images and gradients against the
    oracle evaluated on the same subset of Gaussians."""
    dev = torch.device("cuda:0")
    N, G, K, W, H = 800, 450, 3, 64, 48
    model, sc = _build(N, G, K, W, H, 41, dev)
    fm = None
    if use_filter:
        n_sel = {"fg": G, "bg": N - G, "all": N}[which]
        fm = torch.rand(n_sel, generator=torch.Generator().manual_seed(5)) < 0.7
    t, stage = 3.0, "second"
    dd = lambda x: x.detach().double().cpu().clone()
    pick = lambda params, m: {k: (dd(v) if m is None else dd(v)[m]).requires_grad_() for k, v in params.items()}
    fg = bg = None
    if which in ("fg", "all"):
        fg = pick(model.fg.params, None if fm is None else fm[:G])
    if which in ("bg", "all"):
        bg = pick(model.bg.params, None if fm is None else (fm if which == "bg" else fm[G:]))
    bases = {k: dd(v).requires_grad_() for k, v in model.motion_bases.params.items()} if fg is not None else None
    sd = {k: v.detach().cpu().double() for k, v in model.move_model.state_dict().items()}
    w2c = sc["viewmat"].double()
    RTs, times, dT = ocam.forward_start_end_mid(sd, w2c[:3, :3], w2c[:3, 3:4], t, 11, stage)
    sel = {"mid": slice(5, 6), "start": slice(0, 1), "end": slice(10, 11)}.get(mode, slice(None))
    ref = oscene.render_exposure(fg, bg, bases, times[0, sel], RTs[sel], w2c, sc["K"].double(), (W, H), bg_color=1.0,
                                 return_depth=True, return_mask=True, single=mode in ("mid", "start", "end"))
    out = model.render(t, sc["viewmat"][None].to(dev), sc["K"][None].to(dev), (W, H), return_depth=True, return_mask=True,
                       fg_only=which == "fg", bg_only=which == "bg", filter_mask=None if fm is None else fm.to(dev),
                       mode=mode, stage=stage)
    case = f"S2 variant {which} mode={mode} filter={use_filter}"
    for k in ("img", "mask", "depth", "acc"):
        check(case, k, out[k].cpu(), ref[k], 1e-4, GRAD_FLIP_FRAC)
    g = torch.Generator().manual_seed(1)
    w = torch.randn(1, H, W, 3, generator=g)
    ((out["img"] * w.to(dev)).sum() + out["depth"].sum()).backward()
    ((ref["img"] * w.double()).sum() + ref["depth"].sum()).backward()
    torch.cuda.synchronize()
    if fg is not None:
        m = None if fm is None else fm[:G]
        for k in ("means", "opacities", "motion_coefs"):
            got = model.fg.params[k].grad.cpu()
            want = torch.zeros_like(got, dtype=torch.float64)
            if m is None:
                want = fg[k].grad
            else:
                want[m] = fg[k].grad  # filtered-out Gaussians receive no gradient
            check(case, f"fg.{k}", got, want, 1e-4, GRAD_FLIP_FRAC)
    if bg is not None:
        m = None if fm is None else (fm if which == "bg" else fm[G:])
        got = model.bg.params["scales"].grad.cpu()
        want = torch.zeros_like(got, dtype=torch.float64)
        if m is None:
            want = bg["scales"].grad
        else:
            want[m] = bg["scales"].grad
        check(case, "bg.scales", got, want, 1e-4, GRAD_FLIP_FRAC)


def test_render_from_a_loaded_reference_checkpoint(tmp_path):
    """save the reference Trainer's checkpoint dict, load it back, render on the device: the images equal
    the original model's once its exposure half-widths are at 0.5 too (the loader resets `time_params`)."""
    from deblur4dgs_amd.checkpoint import load_reference_checkpoint, reference_checkpoint_dict

    dev = torch.device("cuda:0")
    N, G, K, W, H = 900, 500, 4, 64, 48
    model, sc = _build(N, G, K, W, H, 29, dev)  # non-default time_params, non-zero heads
    path = tmp_path / "ckpt.pt"
    torch.save(reference_checkpoint_dict(model, global_step=10), path)
    loaded, meta = load_reference_checkpoint(str(path), device=dev)
    assert meta["global_step"] == 10 and loaded.num_fg_gaussians == G and loaded.num_bg_gaussians == N - G
    args = (3.0, sc["viewmat"][None].to(dev), sc["K"][None].to(dev), (W, H))
    with torch.no_grad():
        a = loaded.render(*args, return_depth=True, mode="blury")
        b = model.render(*args, return_depth=True, mode="blury")
        assert not torch.equal(a["img"], b["img"])            # 0.6 vs the reset 0.5 exposure half-width
        model.move_model.time_params.fill_(0.5)
        c = model.render(*args, return_depth=True, mode="blury")
    for k in ("img", "depth", "acc", "RTs", "exposure_imgs"):
        assert torch.equal(a[k], c[k]), k
    # and it trains: gradients reach the loaded leaves
    out = loaded.render(*args, return_depth=True, mode="blury")
    out["img"].square().sum().backward()
    assert loaded.fg.params["means"].grad.abs().sum() > 0 and loaded.move_model.time_params.grad is not None


def _oracle_view(model, w2c, K, t, W, H, stage, return_mask=False):
    """The viewer's render (mode "mid") through the oracle; t None = the canonical pose."""
    dd = lambda x: x.detach().double().cpu().clone()
    fg = {k: dd(v) for k, v in model.fg.params.items()}
    bg = {k: dd(v) for k, v in model.bg.params.items()} if model.bg is not None else None
    bases = {k: dd(v) for k, v in model.motion_bases.params.items()}
    sd = {k: v.detach().cpu().double() for k, v in model.move_model.state_dict().items()}
    w2c, K = w2c.double().cpu(), K.double().cpu()
    RTs, times, _ = ocam.forward_start_end_mid(sd, w2c[:3, :3], w2c[:3, 3:4], 0.0 if t is None else t, 11, stage)
    return oscene.render_exposure(fg, bg, bases, times[0, 5:6].double(), RTs[5:6].double(), w2c, K, (W, H), bg_color=1.0,
                                  return_mask=return_mask, single=True, static_time=t is None)


@pytest.mark.parametrize("t,stage", [(None, "first"), (None, "second"), (2, "second")])
def test_canonical_pose_render_matches_the_oracle(t, stage):
    """`render(t=None, ...)`: the undeformed foreground (raw means, normalised quaternions: scene_model.py:84-85,103-105 through
    `time if t is not None else None`, :327-343) + background, camera delta of the mid sub-sample - by VALUE against the oracle.
    (Upstream, t=None only survives stage "first": stage "second" evaluates `int(None)`, move_model.py:126; the product reads the
    MoveModel at timestep 0 there, whose exposure half-width is 0 by the index rule, and renders the same canonical image.)"""
    dev = torch.device("cuda:0")
    N, G, K, W, H = 900, 500, 4, 64, 48
    model, sc = _build(N, G, K, W, H, 23, dev)
    with torch.no_grad():
        out = model.render(t, sc["viewmat"][None].to(dev), sc["K"][None].to(dev), (W, H), return_mask=True, stage=stage)
    ref = _oracle_view(model, sc["viewmat"], sc["K"], t, W, H, stage, return_mask=True)
    case = f"S2 render t={t} stage={stage}"
    for k in ("img", "mask", "acc"):
        check(case, k, out[k].cpu(), ref[k], 1e-4, GRAD_FLIP_FRAC)
    if t is None:  # the canonical image differs from every deformed frame (the test would not notice a swallowed `t is None` otherwise)
        with torch.no_grad():
            moved = model.render(2, sc["viewmat"][None].to(dev), sc["K"][None].to(dev), (W, H), stage="second")
        assert rel_err(moved["img"].cpu(), ref["img"]) > 1e-2


def test_render_view_values_and_inference_mode():
    """`Renderer.render_fn`'s arithmetic (flow3d/renderer.py:57-89) by value: K from the vertical fov, w2c = inv(c2w), the mid
    sub-sample's image as uint8 - for a frame index and for the viewer's canonical checkbox (t = None) - against the oracle driven
    with the same K / w2c; plus the inference-mode variants (shape contract)."""
    import math

    from deblur4dgs_amd.scene_model import render_view
    from oracle.camera import se3_to_SE3

    dev = torch.device("cuda:0")
    W, H, fov = 80, 48, 1.2
    model, sc = _build(600, 300, 3, W, H, 5, dev)
    g = torch.Generator().manual_seed(3)
    w2c = torch.cat([se3_to_SE3(0.03 * torch.randn(1, 6, generator=g))[0], torch.tensor([[0, 0, 0, 1.0]])], 0)
    c2w = torch.linalg.inv(w2c)
    focal = 0.5 * H / math.tan(0.5 * fov)
    K = torch.tensor([[focal, 0.0, W / 2.0], [0.0, focal, H / 2.0], [0.0, 0.0, 1.0]])
    for t in (2, None):
        img = render_view(model, t, c2w.to(dev), fov, (W, H))
        assert img.shape == (H, W, 3) and img.dtype == torch.uint8
        ref = _oracle_view(model, torch.linalg.inv(c2w.float()), K, t, W, H, "second")["img"][0]
        ref8 = (ref * 255.0).to(torch.uint8)  # truncation, as `.astype(np.uint8)` upstream
        diff = (img.cpu().int() - ref8.int()).abs()
        assert int(diff.max()) <= 1 and float((diff > 0).float().mean()) < 0.02, (t, int(diff.max()), float((diff > 0).float().mean()))
        assert float(ref8.float().std()) > 10  # (a real image, not a constant background)
    with torch.no_grad():
        o = model.render(2, sc["viewmat"][None].to(dev), sc["K"][None].to(dev), (80, 48), bg_only=True)
        assert o["img"].shape == (1, 48, 80, 3)
        o = model.render(2, sc["viewmat"][None].to(dev), sc["K"][None].to(dev), (80, 48), fg_only=True, return_mask=True,
                         filter_mask=torch.arange(300, device=dev) % 3 != 0)
        assert o["mask"].shape == (1, 48, 80, 1)


def test_control_stats_match_reference_accumulation():
    """fused densification statistics vs the literal restatement of trainer.py:967-989."""
    from deblur4dgs_amd.control import accumulate_from_model
    from oracle import control as octl

    dev = torch.device("cuda:0")
    N, G, K, W, H = 1500, 800, 3, 96, 64
    model, sc = _build(N, G, K, W, H, 23, dev)
    stats = {"xys_grad_norm_acc": torch.rand(N, device=dev), "vis_count": torch.randint(0, 5, (N,), device=dev),
             "max_radii": torch.rand(N, device=dev) * 0.05}
    ref = {k: v.clone().cpu() for k, v in stats.items()}
    for _ in range(2):  # two renders accumulate
        out = model.render(3.0, sc["viewmat"][None].to(dev), sc["K"][None].to(dev), (W, H), return_depth=True,
                           mode="blury")
        out["img"].square().sum().backward()
        accumulate_from_model(stats, model, batch_size=2)
        octl.prepare_control_step(ref, [x.grad.cpu() for x in model._current_xys],
                                  [r.cpu() for r in model._current_radii], (W, H), 2)
    torch.cuda.synchronize()
    assert torch.equal(stats["vis_count"].cpu(), ref["vis_count"])
    assert torch.equal(stats["max_radii"].cpu(), ref["max_radii"])  # reference quirk: never updated (index_put)
    assert torch.allclose(stats["xys_grad_norm_acc"].cpu(), ref["xys_grad_norm_acc"], rtol=1e-5, atol=1e-7)
    # fused: the same numbers out of the raster backward's gather epilogue (no torch.cat, no extra pass), bit for bit
    for B in (None, 6):  # B = 6 track frames -> 23 channels -> chunked composite -> statistics on the summed gradient
        fused = {"xys_grad_norm_acc": torch.rand(N, device=dev), "vis_count": torch.randint(0, 5, (N,), device=dev),
                 "max_radii": torch.rand(N, device=dev) * 0.05}
        sep = {k: v.clone() for k, v in fused.items()}
        kw = {} if B is None else dict(target_ts=torch.linspace(0.5, 6.0, B).to(dev),
                                       target_w2cs=torch.eye(4, device=dev).expand(B, 4, 4).contiguous())
        for it in range(2):
            model.attach_control_stats(fused, batch_size=3)
            out = model.render(3.0, sc["viewmat"][None].to(dev), sc["K"][None].to(dev), (W, H), return_depth=True,
                               mode="blury", **kw)
            (out["img"].square().sum() + (out["tracks_3d"].sum() if B else 0.0)).backward()
            model.detach_control_stats()
            accumulate_from_model(sep, model, batch_size=3)
        torch.cuda.synchronize()
        for k in fused:
            assert torch.equal(fused[k], sep[k]), (k, B)
        assert int((fused["vis_count"] - 0).sum()) > 0
    # the intended semantics are available behind a flag
    before = stats["max_radii"].clone()
    accumulate_from_model(stats, model, batch_size=2, update_max_radii=True)
    rad = torch.cat(list(model._current_radii), 0).float().amax(0) / max(W, H)
    assert torch.allclose(stats["max_radii"], torch.maximum(before, rad), rtol=1e-6, atol=0)


def test_training_loop_example_reduces_the_loss():
    """examples/train_dynamic_step.py: three render groups (incl. the 17-channel dynamic one), Adam on every leaf,
    densification statistics - the loss must go down and stay finite."""
    import importlib.util
    import os

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "train_dynamic_step.py")
    spec = importlib.util.spec_from_file_location("train_dynamic_step", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    losses, stats, _ = mod.train(steps=12, W=128, H=96, n_fg=3000, n_bg=5000, K=6, verbose=False)
    assert all(l == l and l < 1e3 for l in losses)
    assert losses[-1] < 0.9 * losses[0], losses
    assert int(stats["vis_count"].sum()) > 0 and float(stats["xys_grad_norm_acc"].sum()) > 0


def test_training_with_control_steps_changes_n_and_keeps_training():
    """SURVEY 8f-1 end to end: statistics kernel -> densify / cull decisions -> parameter + Adam-state surgery; the
    renderer is called with a different N right after (every workspace is sized per call)."""
    import importlib.util
    import os

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "train_dynamic_step.py")
    spec = importlib.util.spec_from_file_location("train_dynamic_step_ctl", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    n0 = 8000
    losses, stats, _ = mod.train(steps=16, W=128, H=96, n_fg=3000, n_bg=5000, K=6, verbose=False, control_every=5)
    assert all(l == l and l < 1e3 for l in losses)
    assert stats["vis_count"].shape[0] != n0            # Gaussians were added / removed (controls at steps 5, 10, 15)
    assert losses[14] < losses[11] and losses[9] < losses[6]   # and the optimizers keep working on the new rows
