"""The reference's OWN training shape at full size (missing #4): 40 k dynamic + 100 k static Gaussians, 20 motion bases,
11 exposure sub-samples, 288x512, 3 colour + 1 mask + 12 track channels + depth = 17 channels (run_training_dynamic.py:118-120,
flow3d/scene_model.py:233-296) - bench.py's `refdefault` scene, through `SceneModel.render` (seam S2: the 17-channel instances of
the composite forward / backward, the MFMA flush, k_blend_bwd, the fg + bg concatenation, mask <- max / depth <- min) against the
oracle's restatement of that method (oracle/scene.py) with the scalar-C fp64 rasterizer behind it (oracle/cref.py
`rasterization_torch`, pinned to the torch one in tests/test_oracle_raster.py) and the fp64 camera generator (oracle/camera.py):
the blurry frame, all 11 sub-sample images, every leaf gradient of both Gaussian sets, the bases, every MoveModel parameter, and
the densification side channel (`_current_xys[s].grad`)."""
import os

import pytest
import torch

from deblur4dgs_amd.synth import make_scene
from oracle import camera as ocam
from oracle import cref
from oracle import scene as oscene
from tests.util import check

pytestmark = pytest.mark.gpu
N, G, K, W, H, SEED = 140_000, 40_000, 20, 512, 288, 1010  # == bench.py CONFIGS / SEEDS["refdefault"]


def test_refdefault_full_frame_through_scene_model_against_the_fp64_oracle(monkeypatch):
    from deblur4dgs_amd.scene_model import GaussianParams, MotionBases, SceneModel

    dev = torch.device("cuda:0")
    sc = make_scene(N, G, K, 11, W, H, seed=SEED, dtype=torch.float32)
    keys = ("means", "quats", "scales", "colors", "opacities")
    fgp = GaussianParams(*[sc[k][:G].clone() for k in keys], motion_coefs=sc["motion_coefs"].clone())
    bgp = GaussianParams(*[sc[k][G:].clone() for k in keys])
    model = SceneModel(sc["K"][None].clone(), sc["viewmat"][None].clone(), fgp, MotionBases(sc["rots"].clone(), sc["transls"].clone()),
                       bgp).to(dev)
    torch.manual_seed(SEED)
    with torch.no_grad():  # non-trivial camera deltas and exposure half-widths
        for head in (model.move_model.RT_head0, model.move_model.RT_head1):
            head[-1].bias.copy_(0.004 * torch.randn(6))
        model.move_model.time_params.copy_(torch.tensor([[0.5, 0.3, 0.45, 0.6, 0.2, 0.5, 0.7, 0.5]]))
    t = 3.0
    tt = torch.tensor([1.0, 2.5, 4.0, 6.0])
    g = torch.Generator().manual_seed(3)
    tw = torch.cat([ocam.se3_to_SE3(0.01 * torch.randn(4, 6, generator=g)), torch.tensor([0, 0, 0, 1.0]).expand(4, 1, 4)], 1)

    # ---- oracle: fp64 generator -> oracle/scene.py over the scalar-C rasterizer
    monkeypatch.setattr(oscene.raster, "rasterization", cref.rasterization_torch)
    dd = lambda x: x.detach().double().cpu().clone().requires_grad_()
    fg = {k: dd(v) for k, v in model.fg.params.items()}
    bg = {k: dd(v) for k, v in model.bg.params.items()}
    bases = {k: dd(v) for k, v in model.motion_bases.params.items()}
    sd = {k: v.detach().cpu().double().requires_grad_() for k, v in model.move_model.state_dict().items()}
    w2c = sc["viewmat"].double()
    RTs, times, dT = ocam.forward_start_end_mid(sd, w2c[:3, :3], w2c[:3, 3:4], t, 11, "second")
    ref = oscene.render_exposure(fg, bg, bases, times[0].double(), RTs.double(), w2c, sc["K"].double(), (W, H), bg_color=1.0,
                                 return_depth=True, return_mask=True, target_ts=tt.double(), target_w2cs=tw.double())
    assert ref["exposure_imgs"].shape == (11, 1, H, W, 17)

    # ---- product
    out = model.render(t, sc["viewmat"][None].to(dev), sc["K"][None].to(dev), (W, H), target_ts=tt.to(dev), target_w2cs=tw.to(dev),
                       return_depth=True, return_mask=True, mode="blury", stage="second")
    torch.cuda.synchronize()
    assert out["exposure_imgs"].shape == (11, 1, H, W, 17) and out["tracks_3d"].shape == (1, H, W, 4, 3)
    assert abs(out["deltaT"].item() - dT.item()) < 1e-6
    case = "refdefault (140 k, K=20, S=11, 17 channels) SceneModel.render vs fp64 oracle"
    # 1e-4 relative per tensor; at most 1e-4 of the elements may miss it (discrete alpha / T decisions in fp32)
    for k in ("img", "mask", "depth", "tracks_3d", "acc", "exposure_imgs"):
        check(case, k, out[k].cpu(), ref[k], 1e-4, 1e-4)

    # ---- backward: the reference's losses read img, mask, depth, tracks_3d (trainer.py:575-700) - random cotangents on all of them
    ws = {k: torch.randn(ref[k].shape, generator=g) for k in ("img", "mask", "depth", "tracks_3d", "acc")}
    ref_leaves = list(fg.values()) + list(bg.values()) + list(bases.values()) + list(sd.values())

    def both_backward(keep):
        """d <out, ws * keep> on both sides; keep [H,W] zeroes the cotangents of some pixels."""
        for p in model.parameters():
            p.grad = None
        for v in ref_leaves:
            v.grad = None
        kd = keep.to(dev).float().view(1, H, W, *([1] * 2))
        kr = keep.double().view(1, H, W, *([1] * 2))
        shp = lambda x, k: k.view(1, H, W, *([1] * (x.dim() - 3)))
        o2 = model.render(t, sc["viewmat"][None].to(dev), sc["K"][None].to(dev), (W, H), target_ts=tt.to(dev), target_w2cs=tw.to(dev),
                          return_depth=True, return_mask=True, mode="blury", stage="second")  # (a fresh graph per backward; same bits)
        sum((o2[k] * ws[k].to(dev) * shp(o2[k], kd)).sum() for k in ws).backward()
        sum((ref[k] * ws[k].double() * shp(ref[k], kr)).sum() for k in ws).backward(retain_graph=True)
        torch.cuda.synchronize()
        got, want = {}, {}
        for part, got_p, ref_p in (("fg", model.fg.params, fg), ("bg", model.bg.params, bg)):
            for k in ref_p:
                got[f"{part}.{k}"], want[f"{part}.{k}"] = got_p[k].grad.cpu().clone(), ref_p[k].grad.clone()
        for k in ("rots", "transls"):
            got[f"bases.{k}"], want[f"bases.{k}"] = model.motion_bases.params[k].grad.cpu().clone(), bases[k].grad.clone()
        for name, p in model.move_model.named_parameters():
            want[f"move_model.{name}"] = sd[name].grad.clone() if sd[name].grad is not None else torch.zeros_like(sd[name])
            got[f"move_model.{name}"] = p.grad.cpu().clone() if p.grad is not None else torch.zeros_like(p).cpu()
        for s in (0, 5, 10):  # the densification side channel: d loss / d means2d (flow3d/scene_model.py:456-461, trainer.py:975)
            got[f"_current_xys[{s}].grad"], want[f"_current_xys[{s}].grad"] = model._current_xys[s].grad[0].cpu().clone(), ref["info"][s]["v_means2d"].clone()
        return got, want

    everything = torch.ones(H, W, dtype=torch.bool)
    got, want = both_backward(everything)
    # The plain comparison: recorded for the parity table and bounded loosely - one flipped decision at a pixel of this dense frame
    # moves a per-Gaussian element by up to 6e-3 x max, and the shared leaves (sums over 40 k Gaussians x 11 sub-samples x 147 k pixels)
    # inherit the flips' sum (measured 1.6e-3 on bases.rots).  What is ASSERTED to 1e-4 with no allowance is the comparison below,
    # with the cotangents zeroed on the pixels where the fp64 oracle's decisions sit at their thresholds.
    from tests.util import frac_bad, record, rel_err

    shared = lambda k: k.startswith(("bases.", "move_model."))
    for k in got:
        record(case, f"grad {k}", got[k], want[k])
        if float(want[k].abs().max()) == 0.0:
            assert float(got[k].abs().max()) == 0.0, k
        elif shared(k):
            assert rel_err(got[k], want[k]) <= 1e-2, (k, rel_err(got[k], want[k]))
        else:
            assert frac_bad(got[k], want[k], 1e-4) <= 2e-3, (k, frac_bad(got[k], want[k], 1e-4), rel_err(got[k], want[k]))
    assert len(model._current_xys) == 11
    for s in (0, 5, 10):
        vis_ref = ref["info"][s]["radii"] > 0
        assert ((model._current_radii[s][0].cpu() > 0) != vis_ref).float().mean() < 1e-3

    # ---- and the CAUSE of whatever missed 1e-4 above (tests/test_gpu_flip_cause.py): zero the cotangents on the pixels where, in the
    # fp64 oracle, a decision of one of the 11 sub-samples sits within eps of its threshold or the mask <- max / depth <- min blend
    # ties; then EVERY gradient element must be within 1e-4 - no allowance
    from oracle import margins

    raw = torch.stack(ref["raw_renders"], 0)[:, 0].detach()
    stack = torch.cat([raw[:-1], raw.mean(0, keepdim=True)], 0)
    eps_px = 32 * 6e-8 * max(W, H)
    ms, toggles = [], torch.zeros(H, W, dtype=torch.bool)
    for s in range(11):
        inf = ref["info"][s]
        m_, q_, s_, o_ = inf["inputs"]
        ms.append(margins.pixel_margins(inf["means2d"], inf["conics"], o_, inf["depths"], inf["flatten_ids"], inf["isect_offsets"], W, H))
        toggles |= margins.gaussian_toggle_mask(m_, q_, s_, o_, w2c, sc["K"].double(), W, H, eps_px=eps_px)[0]
    import json
    import time

    eps = 1e-4
    t0 = time.time()
    F = toggles | margins.blend_tie_mask(stack, eps=eps)
    for mm in ms:
        F = F | margins.fragile_pixels(mm, eps)
    got, want = both_backward(~F)
    worst = {k: float((got[k].double() - want[k]).abs().max() / max(float(want[k].abs().max()), 1e-30)) for k in got
             if float(want[k].abs().max()) > 0 or float(got[k].abs().max()) > 0}
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(dict(case=case, eps=eps, fragile_fraction=float(F.float().mean()), fragile_pixels=int(F.sum()), worst_masked_rel_err=worst,
                   fragile_fraction_per_subsample=[float(margins.fragile_pixels(mm, eps).float().mean()) for mm in ms],
                   seconds_masked_stage=time.time() - t0), open("gpurun_out/flip_cause_refdefault.json", "w"), indent=1)
    for k in got:
        record(case + f" (cotangents zeroed on the {int(F.sum())} fragile pixels, eps {eps:g})", f"grad {k}", got[k], want[k])
    # F is the UNION over the eleven sub-samples (each contributes ~1 % of its pixels); outside it every element must be within 1e-4
    assert float(F.float().mean()) <= 0.2, float(F.float().mean())
    assert max(worst.values()) <= 1e-4, {k: v for k, v in worst.items() if v > 1e-4}
