"""The flip allowance of the parity tests, asserted as a CAUSE instead of a count.

`north_star` asks for 1e-4 relative; `tests/util.py check()` lets a bounded fraction of elements miss it "where one discrete decision
(alpha >= 1/255, T <= 1e-4, ceil(radius)) falls the other way in fp32".  Here that sentence is tested.  From the fp64 oracle ALONE
(oracle/margins.py) take the set F of pixels at which some decision of the reference's algorithm sits within `eps` (relative) of its
threshold - the alpha test, the 0.999 clamp, the transmittance stop, a depth-order near-tie, a tile whose membership in a Gaussian's
rectangle can toggle, a max / min blend tie.  Then:

  (i)   every image / alpha element the device misses by more than 1e-4 x max|ref| lies in F;
  (ii)  with the loss cotangents zeroed on F on BOTH sides - a pixel's cotangent scales everything that pixel contributes to any
        gradient - EVERY element of EVERY gradient is within 1e-4 x max|ref| of the oracle: no allowance at all;
  (iii) F is small (a fraction of a percent of the pixels), so (ii) is a statement about the frame, not about what is left of it.

So the misses of the plain comparison are caused by decisions within eps of their thresholds at those pixels, and by nothing else
(not by an error of the kernels' arithmetic that happens to stay under the allowed count).  The smallest eps of the ladder 1e-5, 1e-4,
1e-3 that explains a case is recorded in gpurun_out/flip_cause.json (committed under profiles/)."""
import json
import os

import pytest
import torch

from oracle import margins, raster
from tests.util import frac_bad, record, rel_err, static_inputs

pytestmark = pytest.mark.gpu
TOL = 1e-4
LADDER = (1e-5, 1e-4, 1e-3)
MAX_FRAGILE = 0.05
_LOG = []


def _note(**kw):
    _LOG.append(kw)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(_LOG, open("gpurun_out/flip_cause.json", "w"), indent=1)


def _eps_px(W, H):
    return 32 * 6e-8 * max(W, H)  # float32 evaluates a projected centre to a few ulp of the image size


def _bad_px(got, ref):
    """[H,W] bool: some channel of the pixel off by more than TOL x max|ref|."""
    return ((got.double() - ref.double()).abs() > TOL * float(ref.abs().max())).any(-1)


@pytest.mark.parametrize("mode,D,N,W,H,scale_mul,seed", [("RGB+ED", 3, 2500, 128, 80, 3.0, 9001), ("RGB", 3, 3000, 160, 96, 3.0, 103),
                                                         ("RGB+ED", 16, 1200, 80, 64, 3.0, 216), ("RGB+ED", 3, 1500, 96, 64, 8.0, 31)])
def test_every_miss_of_the_rasterization_seam_is_a_decision_at_its_threshold(mode, D, N, W, H, scale_mul, seed):
    from tests.test_gpu_rasterization import _run_gpu

    inp = static_inputs(N, W, H, seed=seed, dtype=torch.float64, D=D, scale_mul=scale_mul)
    bg = torch.linspace(0.1, 0.9, D, dtype=torch.float64)
    names = ("means", "quats", "scales", "opac", "colors", "V")
    t = {k: v.clone().requires_grad_(k != "K") for k, v in inp.items()}
    ref_c, ref_a, info = raster.rasterization(t["means"], t["quats"], t["scales"], t["opac"], t["colors"], t["V"], t["K"], W, H,
                                              background=bg, render_mode=mode)
    info["means2d"].retain_grad()
    m = margins.pixel_margins(info["means2d"].detach(), info["conics"].detach(), inp["opac"], info["depths"].detach(),
                              info["flatten_ids"], info["isect_offsets"], W, H)
    toggles, n_tog = margins.gaussian_toggle_mask(inp["means"], inp["quats"], inp["scales"], inp["opac"], inp["V"], inp["K"], W, H, eps_px=_eps_px(W, H))
    g = torch.Generator().manual_seed(9)
    w_c = torch.randn(ref_c.shape, generator=g, dtype=torch.float64)
    w_a = torch.randn(ref_a.shape, generator=g, dtype=torch.float64)

    def oracle_grads(keep):
        for k in names:
            t[k].grad = None
        info["means2d"].grad = None
        ((ref_c * w_c * keep[..., None]).sum() + (ref_a * w_a * keep[..., None]).sum()).backward(retain_graph=True)
        return dict({k: t[k].grad.clone() for k in names}, means2d=info["means2d"].grad.clone())

    def device_grads(keep):
        rc, ra, inf, tg = _run_gpu(inp, W, H, mode, bg, requires_grad=True)
        inf["means2d"].retain_grad()
        dev = rc.device
        k = keep.to(dev).float()[..., None]
        ((rc[0] * w_c.to(dev).float() * k).sum() + (ra[0] * w_a.to(dev).float() * k).sum()).backward()
        torch.cuda.synchronize()
        return rc[0].detach().cpu(), ra[0].detach().cpu(), dict({n: tg[n].grad.cpu() for n in names}, means2d=inf["means2d"].grad[0].cpu())

    everything = torch.ones(H, W, dtype=torch.bool)
    rc, ra, g_dev = device_grads(everything)
    g_ref = oracle_grads(everything.double())
    bad = _bad_px(rc, ref_c.detach()) | _bad_px(ra, ref_a.detach())
    plain = {k: frac_bad(g_dev[k][:3] if k == "V" else g_dev[k], g_ref[k][:3] if k == "V" else g_ref[k], TOL) for k in g_ref}
    case = f"flip cause S1 {mode} D={D} N={N} {W}x{H} x{scale_mul:g}"
    explained = None
    for eps in LADDER:
        F = margins.fragile_pixels(m, eps) | toggles
        if bool((bad & ~F).any()):
            continue  # an image miss at a pixel with no decision within eps of its threshold: not explained at this eps
        keep = ~F
        _, _, h_dev = device_grads(keep)
        h_ref = oracle_grads(keep.double())
        worst = {k: rel_err(h_dev[k][:3] if k == "V" else h_dev[k], h_ref[k][:3] if k == "V" else h_ref[k]) for k in h_ref}
        if max(worst.values()) <= TOL:
            explained = dict(eps=eps, fragile_fraction=float(F.float().mean()), worst_masked_rel_err=worst)
            for k in h_ref:
                record(case + f" (cotangents zeroed on the {int(F.sum())} fragile pixels, eps {eps:g})", k,
                       h_dev[k][:3] if k == "V" else h_dev[k], h_ref[k][:3] if k == "V" else h_ref[k])
            break
    _note(case=case, image_elements_off=int(bad.sum()), plain_gradient_frac_off=plain, toggling_gaussians=n_tog, explained=explained)
    assert explained is not None, (case, int(bad.sum()), plain)
    assert explained["fragile_fraction"] <= MAX_FRAGILE, explained


@pytest.mark.parametrize("N,G,K,S,W,H,tracks", [(1200, 1200, 6, 4, 80, 48, 0), (1500, 900, 4, 3, 96, 64, 0), (700, 300, 20, 11, 64, 48, 4)])
def test_every_miss_of_the_blurry_frame_is_a_decision_at_its_threshold(N, G, K, S, W, H, tracks, monkeypatch):
    """The same through the fused path (deform + camera delta + S sub-samples + blend, every leaf): F = the union over the sub-samples
    of their fragile pixels, plus the pixels where the max / min blend channels tie.  tracks = 4: the reference's 17-channel layout
    (3 + mask + 12 + depth; mask <- max, depth <- min)."""
    from deblur4dgs_amd.exposure import render_exposure
    from deblur4dgs_amd.synth import make_scene
    from oracle import scene as oscene
    from tests.test_gpu_exposure import _split

    sc = make_scene(N, G, K, S, W, H, seed=300 + N, dtype=torch.float64, cam_jitter=0.01)
    sc["scales"] = sc["scales"] + 1.2
    fg, bg, bases = _split(sc, torch.float64)
    shared = dict(times=sc["times"].clone().requires_grad_(), RTs=sc["RTs"].clone().requires_grad_(), w2c=sc["viewmat"].clone().requires_grad_())
    calls = []
    orig = raster.project

    opac_all = torch.cat([torch.sigmoid(p["opacities"].detach()) for p in (fg, bg) if p is not None])

    def spy(means, quats, scales, viewmat, K, *a, **kw):
        calls.append((means.detach(), quats.detach(), scales.detach(), opac_all, viewmat.detach(), K.detach()))
        return orig(means, quats, scales, viewmat, K, *a, **kw)

    monkeypatch.setattr(raster, "project", spy)
    tt = torch.linspace(1.0, 6.0, tracks, dtype=torch.float64) if tracks else None
    mask = bool(tracks) or bg is not None
    out = oscene.render_exposure(fg, bg, bases, shared["times"], shared["RTs"], shared["w2c"], sc["K"], (W, H), bg_color=1.0,
                                 return_depth=True, return_mask=mask, target_ts=tt, single=(S == 1))
    assert len(calls) == S
    keys = [k for k in ("img", "mask", "tracks_3d", "depth") if k in out]
    blended_ref = torch.cat([out[k].reshape(1, H, W, -1) for k in keys], -1)[0]
    F_base = torch.zeros(H, W, dtype=torch.bool)
    ms = []
    for s in range(S):
        inf = out["info"][s]
        ms.append(margins.pixel_margins(inf["means2d"].detach(), inf["conics"].detach(), opac_all, inf["depths"].detach(), inf["flatten_ids"], inf["isect_offsets"], W, H))
        F_base |= margins.gaussian_toggle_mask(*calls[s], W, H, eps_px=_eps_px(W, H))[0]
    raw = torch.stack(out["raw_renders"], 0)[:, 0].detach()  # [S,H,W,D']
    stack = torch.cat([raw[:-1], raw.mean(0, keepdim=True)], 0)  # what the reference's max / min see: raw_0..raw_{S-2}, mean
    g = torch.Generator().manual_seed(1)
    w_b = torch.randn(blended_ref.shape, generator=g, dtype=torch.float64)
    w_a = torch.randn(H, W, generator=g, dtype=torch.float64)
    ref_leaves = dict({f"fg.{k}": v for k, v in fg.items()}, **({f"bg.{k}": v for k, v in bg.items()} if bg is not None else {}),
                      **bases, **shared)

    def oracle_grads(keep):
        for v in ref_leaves.values():
            v.grad = None
        ((blended_ref * w_b * keep[..., None]).sum() + (out["acc"][0, ..., 0] * w_a * keep).sum()).backward(retain_graph=True)
        cat = lambda k: torch.cat([p[k].grad for p in (fg, bg) if p is not None], 0)
        r = {k: cat(k) for k in ("means", "quats", "scales", "colors", "opacities")}
        r.update(motion_coefs=fg["motion_coefs"].grad.clone(), rots=bases["rots"].grad.clone(), transls=bases["transls"].grad.clone(),
                 times=shared["times"].grad.clone(), RTs=shared["RTs"].grad.clone(), viewmat=shared["w2c"].grad[:3].clone())
        return r

    dev = torch.device("cuda:0")

    def device_grads(keep):
        cat = lambda k: torch.cat([p[k].detach() for p in (fg, bg) if p is not None], 0).float().to(dev).requires_grad_()
        P = {k: cat(k) for k in ("means", "quats", "scales", "colors", "opacities")}
        L = dict(P, motion_coefs=fg["motion_coefs"].detach().float().to(dev).requires_grad_(),
                 rots=bases["rots"].detach().float().to(dev).requires_grad_(), transls=bases["transls"].detach().float().to(dev).requires_grad_(),
                 times=shared["times"].detach().float().to(dev).requires_grad_(), RTs=shared["RTs"].detach().float().to(dev).requires_grad_(),
                 viewmat=shared["w2c"].detach().float().to(dev).requires_grad_())
        cols, bgc = P["colors"], torch.ones(3, device=dev)
        if mask:
            mk = torch.zeros(N, 1, device=dev)
            mk[:G] = 1.0
            if G == N:
                mk[:] = 1.0
            cols, bgc = torch.cat([cols, mk], -1), torch.cat([bgc, torch.zeros(1, device=dev)])
        if tracks:
            from deblur4dgs_amd.engine import track_points

            tm = track_points(P["means"], L["motion_coefs"], L["rots"], L["transls"], tt.float().to(dev), None)
            cols, bgc = torch.cat([cols, tm.flatten(-2)], -1), torch.cat([bgc, torch.zeros(3 * tracks, device=dev)])
        res = render_exposure(P["means"], P["quats"], P["scales"], P["opacities"], cols, 3, L["motion_coefs"], L["rots"], L["transls"],
                              L["times"], L["RTs"], L["viewmat"], sc["K"].float().to(dev), W, H, background=bgc, return_depth=True, fused=True)
        k = keep.to(dev).float()
        ((res["blended"] * w_b.float().to(dev) * k[..., None]).sum() + (res["acc"] * w_a.float().to(dev) * k).sum()).backward()
        torch.cuda.synchronize()
        r = {n: L[n].grad.cpu() for n in L}
        r["viewmat"] = r["viewmat"][:3]
        return res["blended"].detach().cpu(), res["acc"].detach().cpu(), r

    everything = torch.ones(H, W, dtype=torch.bool)
    bl, acc, g_dev = device_grads(everything)
    g_ref = oracle_grads(everything.double())
    bad = _bad_px(bl, blended_ref.detach()) | _bad_px(acc[..., None], out["acc"][0].detach())
    plain = {k: frac_bad(g_dev[k], g_ref[k], TOL) for k in g_ref}
    case = f"flip cause fused N={N} G={G} K={K} S={S} {W}x{H} channels={blended_ref.shape[-1]}"
    explained = None
    for eps in LADDER:
        F = F_base | margins.blend_tie_mask(stack, eps=eps)
        for mm in ms:
            F = F | margins.fragile_pixels(mm, eps)
        if bool((bad & ~F).any()):
            continue
        keep = ~F
        _, _, h_dev = device_grads(keep)
        h_ref = oracle_grads(keep.double())
        worst = {k: rel_err(h_dev[k], h_ref[k]) for k in h_ref}
        if max(worst.values()) <= TOL:
            explained = dict(eps=eps, fragile_fraction=float(F.float().mean()), worst_masked_rel_err=worst)
            for k in h_ref:
                record(case + f" (cotangents zeroed on the {int(F.sum())} fragile pixels, eps {eps:g})", k, h_dev[k], h_ref[k])
            break
    _note(case=case, image_elements_off=int(bad.sum()), plain_gradient_frac_off=plain, explained=explained)
    assert explained is not None, (case, int(bad.sum()), plain)
    assert explained["fragile_fraction"] <= 4 * MAX_FRAGILE, explained  # (S sub-samples: S times the per-image share)


@pytest.mark.parametrize("name", ["cfg1", "cfg2"])
def test_every_miss_at_full_size_is_a_decision_at_its_threshold(name):
    """BASELINE cfg1 in full (10 k static Gaussians, 288x512) and one whole exposure sub-sample of cfg2 (300 k Gaussians, 6 bases) against the
    scalar-C fp64 oracle - the comparisons whose worst elements sit at 4e-4 ... 1.3e-3 in the parity table - with the same statement:
    every image miss lies on a fragile pixel, and with the cotangents zeroed there every gradient element is within 1e-4."""
    import numpy as np

    from deblur4dgs_amd.rasterization import rasterization
    from deblur4dgs_amd.synth import make_scene
    from oracle import cref, deform

    N, G, K, S, W, H, seed = {"cfg1": (10_000, 0, 1, 1, 512, 288, 1000), "cfg2": (300_000, 300_000, 6, 8, 512, 288, 1001)}[name]
    sc = make_scene(N, G, K, S, W, H, seed=seed, dtype=torch.float64, cam_jitter=0.0 if name == "cfg1" else 0.002)
    s = S // 2
    with torch.no_grad():
        if G > 0:
            m, q = deform.compute_poses_fg(sc["times"][s:s + 1], sc["means"], sc["quats"], sc["motion_coefs"], sc["rots"], sc["transls"])
            m, q = m[:, 0], q[:, 0]
        else:
            m, q = sc["means"], sc["quats"]
        m = deform.camera_delta(m, sc["RTs"][s])
        sc_, op, col = torch.exp(sc["scales"]), torch.sigmoid(sc["opacities"]), torch.sigmoid(sc["colors"])
    bg = np.array([0.9, 0.5, 0.1])
    out, al, ctx = cref.rasterization(m.numpy(), q.numpy(), sc_.numpy(), op.numpy(), col.numpy(), sc["viewmat"].numpy(), sc["K"].numpy(),
                                      W, H, background=bg, render_mode="RGB+ED", dtype=np.float64)
    n = ctx["n_isect"]
    mg = margins.pixel_margins(torch.from_numpy(ctx["m2d"]), torch.from_numpy(ctx["con"]), op, torch.from_numpy(ctx["dep"]),
                               torch.from_numpy(ctx["flat"][:n]).long(), torch.from_numpy(ctx["offs"]).long(), W, H)
    toggles, n_tog = margins.gaussian_toggle_mask(m, q, sc_, op, sc["viewmat"], sc["K"], W, H, eps_px=_eps_px(W, H))
    g = torch.Generator().manual_seed(4)
    wc = torch.randn(H, W, 4, generator=g, dtype=torch.float64)
    wa = torch.randn(H, W, 1, generator=g, dtype=torch.float64)
    names = ("means", "quats", "scales", "opac", "colors")
    dev = torch.device("cuda:0")

    def oracle_grads(keep):
        r = cref.backward(ctx, (wc * keep[..., None]).numpy(), (wa * keep[..., None]).numpy())
        return {k: torch.from_numpy(r[k]) for k in names}

    def device_grads(keep):
        t = {k: v.float().to(dev).requires_grad_() for k, v in dict(means=m, quats=q, scales=sc_, opac=op, colors=col).items()}
        rc, ra, _ = rasterization(t["means"], t["quats"], t["scales"], t["opac"], t["colors"], sc["viewmat"].float().to(dev)[None],
                                  sc["K"].float().to(dev)[None], W, H, backgrounds=torch.tensor(bg, device=dev).float()[None], render_mode="RGB+ED")
        k = keep.to(dev).float()[..., None]
        ((rc[0] * wc.float().to(dev) * k).sum() + (ra[0] * wa.float().to(dev) * k).sum()).backward()
        torch.cuda.synchronize()
        return rc[0].detach().cpu(), ra[0].detach().cpu(), {n_: t[n_].grad.cpu() for n_ in names}

    everything = torch.ones(H, W, dtype=torch.bool)
    rc, ra, g_dev = device_grads(everything)
    g_ref = oracle_grads(everything.double())
    bad = _bad_px(rc, torch.from_numpy(out)) | _bad_px(ra, torch.from_numpy(al))
    plain = {k: frac_bad(g_dev[k], g_ref[k], TOL) for k in g_ref}
    case = f"flip cause {name} " + ("in full" if name == "cfg1" else f"sub-sample {s}") + " vs scalar-C fp64"
    explained = None
    for eps in LADDER:
        F = margins.fragile_pixels(mg, eps) | toggles
        if bool((bad & ~F).any()):
            continue
        _, _, h_dev = device_grads(~F)
        h_ref = oracle_grads((~F).double())
        worst = {k: rel_err(h_dev[k], h_ref[k]) for k in h_ref}
        if max(worst.values()) <= TOL:
            explained = dict(eps=eps, fragile_fraction=float(F.float().mean()), worst_masked_rel_err=worst)
            for k in h_ref:
                record(case + f" (cotangents zeroed on the {int(F.sum())} fragile pixels, eps {eps:g})", k, h_dev[k], h_ref[k])
            break
    _note(case=case, image_elements_off=int(bad.sum()), plain_gradient_frac_off=plain, toggling_gaussians=n_tog, explained=explained)
    assert explained is not None, (case, int(bad.sum()), plain)
    assert explained["fragile_fraction"] <= MAX_FRAGILE, explained
