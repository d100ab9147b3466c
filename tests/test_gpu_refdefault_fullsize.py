"""The reference's OWN training shape at full size (missing #4): 40 k dynamic + 100 k static Gaussians, 20 motion bases,
11 exposure sub-samples, 288x512, 3 colour + 1 mask + 12 track channels + depth = 17 channels (run_training_dynamic.py:118-120,
flow3d/scene_model.py:233-296) - bench.py's `refdefault` scene, through `SceneModel.render` (seam S2: the 17-channel instances of
the composite forward / backward, the MFMA flush, k_blend_bwd, the fg + bg concatenation, mask <- max / depth <- min) against the
oracle's restatement of that method (oracle/scene.py) with the scalar-C fp64 rasterizer behind it (oracle/cref.py
`rasterization_torch`, pinned to the torch one in tests/test_oracle_raster.py) and the fp64 camera generator (oracle/camera.py):
the blurry frame, all 11 sub-sample images, every leaf gradient of both Gaussian sets, the bases, every MoveModel parameter, and
the densification side channel (`_current_xys[s].grad`)."""
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
    sum((out[k] * ws[k].to(dev)).sum() for k in ws).backward()
    sum((ref[k] * ws[k].double()).sum() for k in ws).backward()
    torch.cuda.synchronize()
    for part, got_p, ref_p in (("fg", model.fg.params, fg), ("bg", model.bg.params, bg)):
        for k in ref_p:
            check(case, f"grad {part}.{k}", got_p[k].grad.cpu(), ref_p[k].grad, 1e-4, 1e-4)
    for k in ("rots", "transls"):  # sums over 40 k Gaussians x 11 sub-samples: no flip allowance
        check(case, f"grad bases.{k}", model.motion_bases.params[k].grad.cpu(), bases[k].grad, 1e-4, 0.0)
    for name, p in model.move_model.named_parameters():
        want = sd[name].grad if sd[name].grad is not None else torch.zeros_like(sd[name])
        got = p.grad.cpu() if p.grad is not None else torch.zeros_like(p).cpu()
        check(case, f"grad move_model.{name}", got, want, 1e-4, 0.0)
    # the densification side channel: d loss / d means2d of every sub-sample (flow3d/scene_model.py:456-461, trainer.py:975)
    assert len(model._current_xys) == 11
    for s in (0, 5, 10):
        check(case, f"_current_xys[{s}].grad", model._current_xys[s].grad[0].cpu(), ref["info"][s]["v_means2d"], 1e-4, 1e-4)
        vis_ref = ref["info"][s]["radii"] > 0
        assert ((model._current_radii[s][0].cpu() > 0) != vis_ref).float().mean() < 1e-3
