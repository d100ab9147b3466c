"""GPU parity of the pose API of seam S2: `SceneModel.compute_poses_fg / compute_poses_bg /
compute_poses_all / compute_transforms` and `MotionBases.compute_transforms` (reference flow3d/scene_model.py:58-120,
flow3d/params.py:142-180; called by the reference's Trainer at flow3d/trainer.py:303,478,485,701,818 and its Renderer at
flow3d/renderer.py:37) - values and every gradient against fp64 autograd of oracle/deform.py (compute_transforms pinned
by golden F1 incl. clamped / negative t; the roma quaternion half is a restatement: parity unpinned).  Tolerance:
1e-4 * max|ref| per tensor, no flip allowance (nothing discrete on this path except the clamped floor/ceil of t)."""
import pytest
import torch

from deblur4dgs_amd.synth import make_scene
from oracle import deform
from tests.util import check

pytestmark = pytest.mark.gpu
KEYS = ("means", "quats", "scales", "colors", "opacities")


def _model(N, G, K, T, seed, dev, has_bg=True):
    from deblur4dgs_amd.scene_model import GaussianParams, MotionBases, SceneModel

    sc = make_scene(N, G, K, 1, 64, 48, seed=seed, dtype=torch.float32, T=T)
    g = torch.Generator().manual_seed(seed)
    sc["rots"] = sc["rots"] + 0.4 * torch.randn(sc["rots"].shape, generator=g)  # rotations far from identity: all four
    fg = GaussianParams(*[sc[k][:G].clone() for k in KEYS], motion_coefs=sc["motion_coefs"].clone())  # quaternion branches
    bg = GaussianParams(*[sc[k][G:].clone() for k in KEYS]) if has_bg else None
    mb = MotionBases(sc["rots"].clone(), sc["transls"].clone())
    return SceneModel(sc["K"][None].clone(), sc["viewmat"][None].clone(), fg, mb, bg).to(dev), sc


def _dd(x):
    return x.detach().double().cpu().clone().requires_grad_()


def _oracle_leaves(model):
    fg = {k: _dd(v) for k, v in model.fg.params.items()}
    bg = {k: _dd(v) for k, v in model.bg.params.items()} if model.bg is not None else None
    bases = {k: _dd(v) for k, v in model.motion_bases.params.items()}
    return fg, bg, bases


def _zero_grads(model):
    for p in model.parameters():
        p.grad = None


@pytest.mark.parametrize("K,T,ts", [(4, 8, [0.0, 2.37, 5.0, 7.0, 9.5, -1.25]),   # integer, fractional, last, > T-1, < 0
                                    (12, 24, [3.0, 3.5, 11.75]),                 # K > 8: the matrix-pipe column sums
                                    (1, 5, [1.5])])
def test_compute_poses_all_and_fg_match_oracle(K, T, ts):
    dev = torch.device("cuda:0")
    N, G = 700, 450
    model, _ = _model(N, G, K, T, 5 + K, dev)
    tsd = torch.tensor(ts, device=dev)
    B = len(ts)
    fg, bg, bases = _oracle_leaves(model)
    ts64 = torch.tensor(ts, dtype=torch.float64)
    mr, qr = deform.compute_poses_all(ts64, fg, bases, bg)
    m, q = model.compute_poses_all(tsd)
    assert m.shape == (N, B, 3) and q.shape == (N, B, 4) and m.is_contiguous() and q.is_contiguous()
    case = f"poses_all K={K} B={B}"
    check(case, "means", m.cpu(), mr)
    # q and -q are the same rotation, but the reference's sign is part of the contract (roma's branch + Hamilton product)
    check(case, "quats", q.cpu(), qr)
    gen = torch.Generator().manual_seed(1)
    wm, wq = torch.randn(N, B, 3, generator=gen, dtype=torch.float64), torch.randn(N, B, 4, generator=gen, dtype=torch.float64)
    ((mr * wm).sum() + (qr * wq).sum()).backward()
    _zero_grads(model)
    ((m * wm.float().to(dev)).sum() + (q * wq.float().to(dev)).sum()).backward()
    torch.cuda.synchronize()
    for name in ("means", "quats", "motion_coefs"):
        check(case, f"grad fg.{name}", model.fg.params[name].grad.cpu(), fg[name].grad)
    for name in ("means", "quats"):
        check(case, f"grad bg.{name}", model.bg.params[name].grad.cpu(), bg[name].grad)
    for name in ("rots", "transls"):
        check(case, f"grad bases.{name}", model.motion_bases.params[name].grad.cpu(), bases[name].grad)
    assert model.fg.params["scales"].grad is None  # not on this path

    # compute_poses_fg with inds (flow3d/renderer.py:37) and with ts = None
    inds = torch.tensor([3, 0, 17, 17, G - 1], device=dev)
    mf, qf = model.compute_poses_fg(tsd, inds=inds)
    ic = inds.cpu()
    mfr, qfr = deform.compute_poses_fg(ts64, fg["means"][ic], fg["quats"][ic], fg["motion_coefs"][ic], bases["rots"],
                                       bases["transls"])
    check(case, "fg[inds] means", mf.cpu(), mfr.detach())
    check(case, "fg[inds] quats", qf.cpu(), qfr.detach())
    m0, q0 = model.compute_poses_fg(None)
    assert m0.shape == (G, 1, 3) and q0.shape == (G, 1, 4)
    assert torch.equal(m0[:, 0], model.fg.params["means"].detach())
    check(case, "fg[ts=None] quats", q0[:, 0].cpu(), deform.act_quats(fg["quats"]).detach())


def test_compute_poses_bg_and_static_scene():
    dev = torch.device("cuda:0")
    model, _ = _model(300, 120, 3, 6, 11, dev)
    fg, bg, bases = _oracle_leaves(model)
    m, q = model.compute_poses_bg()
    assert m.shape == (180, 3) and q.shape == (180, 4)
    assert torch.equal(m, model.bg.params["means"].detach())
    check("poses_bg", "quats", q.cpu(), deform.act_quats(bg["quats"]).detach())
    w = torch.randn(180, 4, dtype=torch.float64)
    (deform.act_quats(bg["quats"]) * w).sum().backward()
    (q * w.float().to(dev)).sum().backward()
    check("poses_bg", "grad bg.quats", model.bg.params["quats"].grad.cpu(), bg["quats"].grad)
    # a model without background: compute_poses_all == compute_poses_fg
    m2, _ = _model(200, 200, 2, 6, 12, dev, has_bg=False)
    ts = torch.tensor([1.25, 4.0], device=dev)
    a, b = m2.compute_poses_all(ts), m2.compute_poses_fg(ts)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


@pytest.mark.parametrize("K", [5, 20])
def test_compute_transforms_matches_oracle(K):
    """flow3d/trainer.py:701: `compute_transforms(cat(ts - 1, ts, ts + 1))` -> (G, 3n, 3, 4), differentiated into the
    coefficients and the bases (track-smoothness loss); and MotionBases.compute_transforms on activated coefficients."""
    dev = torch.device("cuda:0")
    G, T = 333, 12
    model, _ = _model(G + 50, G, K, T, 21 + K, dev)
    ts = torch.tensor([1.0, 4.0, 9.0])
    tn = torch.cat((ts - 1, ts, ts + 1))
    fg, bg, bases = _oracle_leaves(model)
    ref = deform.compute_transforms(tn.double(), deform.act_coefs(fg["motion_coefs"]), bases["rots"], bases["transls"])
    out = model.compute_transforms(tn.to(dev))
    assert out.shape == (G, 9, 3, 4) and out.is_contiguous()
    case = f"compute_transforms K={K}"
    check(case, "transforms", out.cpu(), ref)
    w = torch.randn(G, 9, 3, 4, dtype=torch.float64, generator=torch.Generator().manual_seed(3))
    (ref * w).sum().backward()
    _zero_grads(model)
    (out * w.float().to(dev)).sum().backward()
    check(case, "grad motion_coefs", model.fg.params["motion_coefs"].grad.cpu(), fg["motion_coefs"].grad)
    check(case, "grad rots", model.motion_bases.params["rots"].grad.cpu(), bases["rots"].grad)
    check(case, "grad transls", model.motion_bases.params["transls"].grad.cpu(), bases["transls"].grad)
    # integer frame indices and an index subset, as the reference's callers pass them
    inds = torch.arange(10, device=dev)
    o2 = model.compute_transforms(torch.arange(T, device=dev), inds=inds)
    r2 = deform.compute_transforms(torch.arange(T).double(), deform.act_coefs(fg["motion_coefs"][:10]), bases["rots"], bases["transls"])
    check(case, "transforms[inds], integer ts", o2.cpu(), r2.detach())
    # the MotionBases method takes ACTIVATED coefficients (params.py:142) and differentiates w.r.t. them
    coefs = torch.softmax(model.fg.params["motion_coefs"].detach(), -1).requires_grad_()
    c64 = coefs.detach().double().cpu().requires_grad_()
    r3 = deform.compute_transforms(tn.double(), c64, bases["rots"].detach(), bases["transls"].detach())
    o3 = model.motion_bases.compute_transforms(tn.to(dev), coefs)
    check(case, "MotionBases.compute_transforms", o3.cpu(), r3.detach())
    (r3 * w).sum().backward()
    (o3 * w.float().to(dev)).sum().backward()
    check(case, "grad activated coefs", coefs.grad.cpu(), c64.grad)


def test_trainer_style_use_of_the_pose_api():
    """The call pattern of Trainer.compute_dynamic_losses (flow3d/trainer.py:478-486,701-706): transposes, splits and an
    einsum on the returned tensors, then one backward through all of it."""
    dev = torch.device("cuda:0")
    N, G, K, T = 500, 300, 6, 10
    model, _ = _model(N, G, K, T, 31, dev)
    ts = torch.tensor([2.0, 5.0], device=dev)
    means, quats = model.compute_poses_all(ts)
    means, quats = means.transpose(0, 1), quats.transpose(0, 1)
    target_means, _ = model.compute_poses_all(torch.tensor([1.0, 3.0, 4.0, 6.0], device=dev))
    target_mean_list = target_means.transpose(0, 1).split(2)
    tsn = torch.cat((ts - 1, ts, ts + 1))
    tf = model.compute_transforms(tsn)
    nbs = torch.einsum("pnij,pj->pni", tf, torch.nn.functional.pad(model.fg.params["means"], (0, 1), value=1.0))
    loss = means.square().mean() + sum(t.abs().mean() for t in target_mean_list) + nbs.reshape(G, 3, -1, 3).diff(dim=1).square().mean()
    loss.backward()
    torch.cuda.synchronize()
    for p in (model.fg.params["means"], model.fg.params["motion_coefs"], model.motion_bases.params["rots"],
              model.motion_bases.params["transls"], model.bg.params["means"]):
        assert p.grad is not None and torch.isfinite(p.grad).all() and p.grad.abs().sum() > 0
    assert quats.shape == (2, N, 4)
