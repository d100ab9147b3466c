"""the CPU twins d4gs_forward_cpu / d4gs_backward_cpu (csrc/cpu_twin.hip, product code inside libd4gs.so) against
the fp64 torch oracle - no GPU needed.  BASELINE.json configs[0] (10 k static Gaussians, 1 camera, 288x512, N_exposure = 1) in
full, and small dynamic scenes through the whole seam (deformation, camera delta, S sub-samples, blend policies): images
and EVERY leaf gradient at north_star's 1e-4.  The twin is fp32 and scalar; the allowances are the ones of the device tests."""
import pytest
import torch

from deblur4dgs_amd.synth import make_scene
from oracle import scene as oscene
from tests.test_gpu_exposure import _split
from tests.util import frac_bad, rel_err

TOL, FLIPS = 1e-4, 2e-3


def _close(name, got, ref, flips=FLIPS):
    bad = frac_bad(got, ref, TOL)
    assert bad <= flips, f"{name}: {bad:.2e} of the elements off by > {TOL:g} x max|ref| (max err {rel_err(got, ref):.2e})"


@pytest.mark.parametrize("N,G,K,S,W,H,mask,depth", [(400, 250, 3, 3, 64, 48, True, True), (300, 300, 5, 2, 48, 40, False, False),
                                                    (350, 0, 1, 1, 56, 40, False, True)])
def test_cpu_twin_matches_the_oracle(N, G, K, S, W, H, mask, depth):
    from deblur4dgs_amd.cpu_twin import render_exposure_cpu

    sc = make_scene(N, G, K, S, W, H, seed=700 + N, dtype=torch.float64, cam_jitter=0.01)
    sc["scales"] = sc["scales"] + 1.2
    keys = ("means", "quats", "scales", "colors", "opacities")
    if G:
        fg, bg, bases = _split(sc, torch.float64)
    else:
        fg, bases, bg = None, None, {k: sc[k].clone().requires_grad_() for k in keys}
    times, RTs, w2c = sc["times"].clone().requires_grad_(), sc["RTs"].clone().requires_grad_(), sc["viewmat"].clone().requires_grad_()
    out = oscene.render_exposure(fg, bg, bases, times, RTs, w2c, sc["K"], (W, H), bg_color=1.0, return_depth=depth, return_mask=mask,
                                 single=(S == 1))
    blended_ref = torch.cat([out[k] for k in ("img", "mask", "depth") if k in out], -1)[0]
    g = torch.Generator().manual_seed(1)
    w_b = torch.randn(blended_ref.shape, generator=g, dtype=torch.float64)
    w_a = torch.randn(out["acc"][0].shape, generator=g, dtype=torch.float64)
    raw_stack = torch.stack(out["raw_renders"], 0)[:, 0]
    w_r = 0.1 * torch.randn(raw_stack.shape, generator=g, dtype=torch.float64)
    ((blended_ref * w_b).sum() + (out["acc"][0] * w_a).sum() + (raw_stack * w_r).sum()).backward()

    cat = lambda k: torch.cat([p[k].detach() for p in (fg, bg) if p is not None], 0).float().requires_grad_()
    P = {k: cat(k) for k in keys}
    leaf = lambda t: None if t is None else t.detach().float().requires_grad_()
    coefs = leaf(fg["motion_coefs"]) if G else None
    rots, transls = (leaf(bases["rots"]), leaf(bases["transls"])) if G else (None, None)
    tms, rts, vm = leaf(times), leaf(RTs), leaf(w2c)
    colors_in, bgc = P["colors"], torch.ones(3)
    policy = None
    if mask:
        mk = torch.zeros(N, 1)
        mk[: (G if 0 < G < N else N)] = 1.0
        colors_in, bgc = torch.cat([colors_in, mk], -1), torch.cat([bgc, torch.zeros(1)])
        policy = [0, 0, 0, 1] + ([0] if depth else [])  # the reference's channel 3 <- max (scene_model.py:390)
    res = render_exposure_cpu(P["means"], P["quats"], P["scales"], P["opacities"], colors_in, 3, coefs, rots, transls, tms, rts, vm,
                              sc["K"].float(), W, H, background=bgc, return_depth=depth, policy=policy)
    _close("renders", res["renders"], raw_stack)
    _close("blended", res["blended"], blended_ref)
    _close("acc", res["acc"], out["acc"][0, ..., 0])
    assert int(res["n_isect"][0]) > 0 and 0 < int(res["n_isect"][3]) <= int(res["n_isect"][2]) <= int(res["n_isect"][0])
    ((res["blended"] * w_b.float()).sum() + (res["acc"] * w_a[..., 0].float()).sum() + (res["renders"] * w_r.float()).sum()).backward()
    ref_cat = lambda k: torch.cat([p[k].grad for p in (fg, bg) if p is not None], 0)
    for k in keys:
        _close(k, P[k].grad, ref_cat(k))
    if G:
        _close("motion_coefs", coefs.grad, fg["motion_coefs"].grad)
        _close("rots", rots.grad, bases["rots"].grad, 0.0)
        _close("transls", transls.grad, bases["transls"].grad, 0.0)
        _close("times", tms.grad, times.grad, 0.0)
    _close("RTs", rts.grad, RTs.grad, 0.0)
    _close("viewmat", vm.grad[:3], w2c.grad[:3], 0.0)


def test_cfg1_on_the_cpu_twin_in_full():
    """BASELINE.json configs[0] on its exact workload (seed 1000, identity camera delta) - the CPU-runnable case."""
    from deblur4dgs_amd.cpu_twin import render_exposure_cpu

    N, W, H = 10_000, 512, 288
    sc = make_scene(N, 0, 1, 1, W, H, seed=1000, dtype=torch.float64, cam_jitter=0.0)
    keys = ("means", "quats", "scales", "colors", "opacities")
    bg = {k: sc[k].clone().requires_grad_() for k in keys}
    w2c = sc["viewmat"].clone().requires_grad_()
    ref = oscene.render_exposure(None, bg, None, sc["times"], sc["RTs"], w2c, sc["K"], (W, H), bg_color=1.0, return_depth=True, single=True)
    ref_img = torch.cat([ref["img"], ref["depth"]], -1)[0]
    g = torch.Generator().manual_seed(0)
    w_i = torch.randn(ref_img.shape, generator=g, dtype=torch.float64)
    w_a = torch.randn(H, W, generator=g, dtype=torch.float64)
    ((ref_img * w_i).sum() + (ref["acc"][0, ..., 0] * w_a).sum()).backward()
    P = {k: sc[k].float().requires_grad_() for k in keys}
    vm = sc["viewmat"].float().requires_grad_()
    res = render_exposure_cpu(P["means"], P["quats"], P["scales"], P["opacities"], P["colors"], 3, None, None, None, None,
                              sc["RTs"].float(), vm, sc["K"].float(), W, H, background=torch.ones(3), return_depth=True)
    assert res["renders"].shape == (1, H, W, 4) and 0 < int(res["n_isect"][0]) <= ref["info"][0]["n_isect"]
    assert ((res["radii"][0] > 0) != (ref["info"][0]["radii"] > 0)).float().mean() < 1e-3
    _close("blended", res["blended"], ref_img, 1e-4)
    _close("acc", res["acc"], ref["acc"][0, ..., 0], 1e-4)
    ((res["blended"] * w_i.float()).sum() + (res["acc"] * w_a.float()).sum()).backward()
    for k in keys:
        _close(k, P[k].grad, bg[k].grad, 1e-4)
    _close("viewmat", vm.grad[:3], w2c.grad[:3], 0.0)


def test_cpu_twin_is_an_entry_point_not_a_fallback():
    from deblur4dgs_amd import exposure
    from deblur4dgs_amd.cpu_twin import render_exposure_cpu

    sc = make_scene(50, 0, 1, 1, 32, 32, seed=5, dtype=torch.float32, cam_jitter=0.0)
    with pytest.raises(RuntimeError, match="CPU tensor"):  # the device seam still refuses CPU tensors
        exposure.render_exposure(sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["colors"], 3, None, None, None, None,
                                 sc["RTs"], sc["viewmat"], sc["K"], 32, 32)
    src = open(exposure.__file__).read() + open(exposure.__file__.replace("exposure.py", "engine.py")).read()
    assert "cpu_twin" not in src and "_cpu(" not in src  # nothing on the device path routes here


def test_cpu_twin_activated_inputs_no_background_no_blend():
    """The gsplat-seam flavour of the flags: activated scales / opacities (raw_params=False), no background, blend off."""
    from deblur4dgs_amd.cpu_twin import render_exposure_cpu
    from oracle import raster
    from tests.util import static_inputs

    W, H, N = 72, 56, 500
    inp = static_inputs(N, W, H, seed=77, dtype=torch.float64, D=3, scale_mul=3.0)
    ref = {k: v.clone().requires_grad_() for k, v in inp.items() if k in ("means", "quats", "scales", "opac", "colors")}
    rc, ra, _ = raster.rasterization(ref["means"], ref["quats"], ref["scales"], ref["opac"], ref["colors"], inp["V"], inp["K"], W, H,
                                     background=None, render_mode="RGB")
    g = torch.Generator().manual_seed(3)
    w = torch.randn(rc.shape, generator=g, dtype=torch.float64)
    ((rc * w).sum() + 0.3 * ra.sum()).backward()
    P = {k: inp[k].float().requires_grad_() for k in ref}
    res = render_exposure_cpu(P["means"], P["quats"], P["scales"], P["opac"], P["colors"], 0, None, None, None, None, None,
                              inp["V"].float(), inp["K"].float(), W, H, background=None, return_depth=False, blend=False,
                              raw_params=False)
    assert res["blended"] is None and res["renders"].shape == (1, H, W, 3)
    _close("renders", res["renders"][0], rc)
    _close("alphas", res["alphas"][0, ..., 0], ra[..., 0])
    ((res["renders"][0] * w.float()).sum() + 0.3 * res["alphas"].sum()).backward()
    for k in ref:
        _close(k, P[k].grad, ref[k].grad)


def test_cpu_twin_default_policy_is_the_reference_policy():
    """`policy=None` must mean what it means on the device path - channel 3 <- max_S, channel 16 <- min_S
    (scene_model.py:392-393) - not "mean everywhere".  S > 1 and NCH > 3 (RGB + depth), default arguments."""
    from deblur4dgs_amd.cpu_twin import render_exposure_cpu
    from deblur4dgs_amd.exposure import reference_policy

    N, G, K, S, W, H = 300, 200, 3, 3, 48, 40
    sc = make_scene(N, G, K, S, W, H, seed=911, dtype=torch.float32, cam_jitter=0.01)
    sc["scales"] = sc["scales"] + 1.2
    args = (sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["colors"], 3, sc["motion_coefs"], sc["rots"], sc["transls"],
            sc["times"], sc["RTs"], sc["viewmat"], sc["K"], W, H)
    dflt = render_exposure_cpu(*args, background=torch.ones(3), return_depth=True)
    expl = render_exposure_cpu(*args, background=torch.ones(3), return_depth=True, policy=reference_policy(4))
    mean = render_exposure_cpu(*args, background=torch.ones(3), return_depth=True, policy=[0, 0, 0, 0])
    assert torch.equal(dflt["blended"], expl["blended"])
    assert torch.equal(dflt["blended"][..., :3], mean["blended"][..., :3])
    assert not torch.equal(dflt["blended"][..., 3], mean["blended"][..., 3])  # the depth channel takes the max, not the mean
    raw = dflt["renders"][..., 3]
    want = torch.maximum(raw[:-1].amax(0), raw.mean(0))  # max{raw_0 .. raw_{S-2}, mean}: the reference's in-place quirk
    assert torch.allclose(dflt["blended"][..., 3], want, rtol=1e-6, atol=1e-6)
