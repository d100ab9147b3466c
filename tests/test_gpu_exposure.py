"""GPU parity (fused seam): deform + camera delta + S sub-samples + exposure blend vs the torch oracle's
restatement of SceneModel.render's loop (oracle/scene.py), forward and every leaf gradient."""
import pytest
import torch

from deblur4dgs_amd.synth import make_scene
from oracle import scene as oscene
from tests.util import check, frac_bad, rel_err

pytestmark = pytest.mark.gpu
# north_star's 1e-4 relative for images and every gradient (measured: profiles/r02_parity_table.md, typically 1e-5);
# FLIPS = fraction of elements that may miss it because one alpha / T decision falls the other way in fp32
GTOL, GFLIPS = 1e-4, 2e-3
STOL = 1e-4                  # shared leaves (bases, times, camera deltas, viewmat): sums over all Gaussians, no allowance


def _split(sc, dtype):
    G = sc["G"]
    keys = ("means", "quats", "scales", "colors", "opacities")
    fg = {k: sc[k][:G].to(dtype).clone().requires_grad_() for k in keys}
    fg["motion_coefs"] = sc["motion_coefs"].to(dtype).clone().requires_grad_()
    bg = {k: sc[k][G:].to(dtype).clone().requires_grad_() for k in keys} if G < sc["N"] else None
    bases = {k: sc[k].to(dtype).clone().requires_grad_() for k in ("rots", "transls")}
    return fg, bg, bases


@pytest.mark.parametrize("N,G,K,S,W,H,mask,depth", [(1500, 900, 4, 3, 96, 64, True, True),
                                                    (1200, 1200, 6, 4, 80, 48, False, True),
                                                    (900, 300, 2, 1, 64, 64, True, False),
                                                    (1000, 600, 5, 2, 72, 40, False, False),
                                                    # the reference's defaults: 20 bases, 11 sub-samples
                                                    (700, 300, 20, 11, 64, 48, True, True),
                                                    (500, 500, 32, 16, 48, 48, False, True),
                                                    # small S: k_project_bwd's sub-group mapping (4 / 2 groups of 64 Gaussians per
                                                    # block), on the VALU (K <= 8) and on the matrix pipe (K > 8), ragged N / G
                                                    (1100, 700, 6, 1, 80, 48, False, True),
                                                    (1000, 1000, 12, 1, 64, 48, True, True),
                                                    (900, 500, 20, 2, 64, 48, False, True)])
def test_exposure_forward_backward(N, G, K, S, W, H, mask, depth):
    from deblur4dgs_amd.exposure import render_exposure

    sc = make_scene(N, G, K, S, W, H, seed=300 + N, dtype=torch.float64, cam_jitter=0.01)
    # make the Gaussians big enough to overlap several pixels
    sc["scales"] = sc["scales"] + 1.2
    fg, bg, bases = _split(sc, torch.float64)
    times = sc["times"].clone().requires_grad_()
    RTs = sc["RTs"].clone().requires_grad_()
    w2c = sc["viewmat"].clone().requires_grad_()
    out = oscene.render_exposure(fg, bg, bases, times, RTs, w2c, sc["K"], (W, H), bg_color=1.0, return_depth=depth,
                                 return_mask=mask, single=(S == 1))
    blended_ref = torch.cat([out[k] for k in ("img", "mask", "depth") if k in out], -1)[0]
    g = torch.Generator().manual_seed(1)
    w_b = torch.randn(blended_ref.shape, generator=g, dtype=torch.float64)
    w_a = torch.randn(out["acc"][0].shape, generator=g, dtype=torch.float64)
    raw_stack = torch.stack(out["raw_renders"], 0)[:, 0]  # [S,H,W,D']
    w_r = 0.1 * torch.randn(raw_stack.shape, generator=g, dtype=torch.float64)
    ((blended_ref * w_b).sum() + (out["acc"][0] * w_a).sum() + (raw_stack * w_r).sum()).backward()

    dev = torch.device("cuda:0")
    cat = lambda k: torch.cat([p[k].detach() for p in (fg, bg) if p is not None], 0).float().to(dev).requires_grad_()
    P = {k: cat(k) for k in ("means", "quats", "scales", "colors", "opacities")}
    coefs = fg["motion_coefs"].detach().float().to(dev).requires_grad_()
    rots = bases["rots"].detach().float().to(dev).requires_grad_()
    transls = bases["transls"].detach().float().to(dev).requires_grad_()
    tms = times.detach().float().to(dev).requires_grad_()
    rts = RTs.detach().float().to(dev).requires_grad_()
    vm = w2c.detach().float().to(dev).requires_grad_()
    colors_in = P["colors"]
    bgc = torch.ones(3, device=dev)
    if mask:
        mk = torch.zeros(N, 1, device=dev)
        mk[:G] = 1.0
        if G == N:
            mk[:] = 1.0
        colors_in = torch.cat([colors_in, mk], -1)
        bgc = torch.cat([bgc, torch.zeros(1, device=dev)])
    res = render_exposure(P["means"], P["quats"], P["scales"], P["opacities"], colors_in, 3, coefs, rots, transls, tms,
                          rts, vm, sc["K"].float().to(dev), W, H, background=bgc, return_depth=depth)
    torch.cuda.synchronize()
    case = f"fused N={N} G={G} K={K} S={S} {W}x{H} mask={mask} depth={depth}"
    check(case, "renders", res["renders"].cpu(), raw_stack, 1e-4, GFLIPS)
    check(case, "blended", res["blended"].cpu(), blended_ref, 1e-4, GFLIPS)
    check(case, "acc", res["acc"].cpu(), out["acc"][0, ..., 0], 1e-4, GFLIPS)
    loss = (res["blended"] * w_b.float().to(dev)).sum() + (res["acc"] * w_a[..., 0].float().to(dev)).sum() + \
        (res["renders"] * w_r.float().to(dev)).sum()
    loss.backward()
    torch.cuda.synchronize()

    def ref_cat(k):
        return torch.cat([p[k].grad for p in (fg, bg) if p is not None], 0)

    for k in ("means", "quats", "scales", "colors", "opacities"):
        check(case, k, P[k].grad.cpu(), ref_cat(k), GTOL, GFLIPS)
    check(case, "motion_coefs", coefs.grad.cpu(), fg["motion_coefs"].grad, GTOL, GFLIPS)
    # shared leaves: every element is a sum over all Gaussians (no isolated elements -> no flip allowance)
    check(case, "rots", rots.grad.cpu(), bases["rots"].grad, STOL)
    check(case, "transls", transls.grad.cpu(), bases["transls"].grad, STOL)
    check(case, "times", tms.grad.cpu(), times.grad, STOL)
    check(case, "RTs", rts.grad.cpu(), RTs.grad, STOL)
    check(case, "viewmat", vm.grad.cpu()[:3], w2c.grad[:3], STOL)


def test_track_points_match_oracle():
    """a11 (scene_model.py:258-289): positions at B target times in B target cameras, forward + all gradients."""
    from deblur4dgs_amd.engine import track_points
    from oracle.camera import se3_to_SE3
    from oracle import deform

    dev = torch.device("cuda:0")
    N, G, K, B = 700, 450, 5, 4
    sc = make_scene(N, G, K, 1, 64, 64, seed=91, dtype=torch.float64)
    fg, bg, bases = _split(sc, torch.float64)
    ts = torch.tensor([0.0, 2.4, 7.0, 30.0], dtype=torch.float64)  # integer, fractional, and clamped (> T-1)
    w2c = torch.cat([se3_to_SE3(0.1 * torch.randn(B, 6)).double(), torch.tensor([0, 0, 0, 1.0]).expand(B, 1, 4).double()], 1)
    m, _ = deform.compute_poses_all(ts, fg, bases, bg)  # [N,B,3]
    ref = torch.einsum("bij,pbj->pbi", w2c[:, :3], torch.nn.functional.pad(m, (0, 1), value=1.0))
    wgt = torch.randn(ref.shape, generator=torch.Generator().manual_seed(2), dtype=torch.float64)
    (ref * wgt).sum().backward()

    means = torch.cat([fg["means"], bg["means"]], 0).detach().float().to(dev).requires_grad_()
    coefs = fg["motion_coefs"].detach().float().to(dev).requires_grad_()
    rots = bases["rots"].detach().float().to(dev).requires_grad_()
    transls = bases["transls"].detach().float().to(dev).requires_grad_()
    out = track_points(means, coefs, rots, transls, ts.float().to(dev), w2c.float().to(dev))
    assert out.shape == (N, B, 3)
    (out * wgt.float().to(dev)).sum().backward()
    torch.cuda.synchronize()
    check("a11 track points", "points", out.cpu(), ref, 1e-5)
    check("a11 track points", "means", means.grad.cpu(), torch.cat([fg["means"].grad, bg["means"].grad], 0), 1e-4)
    check("a11 track points", "motion_coefs", coefs.grad.cpu(), fg["motion_coefs"].grad, 1e-4)
    check("a11 track points", "rots", rots.grad.cpu(), bases["rots"].grad, 1e-4)
    check("a11 track points", "transls", transls.grad.cpu(), bases["transls"].grad, 1e-4)


def test_grad_arena_receives_the_leaf_gradients_without_copies():
    """`grad_arena` (used by the view-sharded multi-GPU driver): the projection backward writes the leaf gradients into
    caller-provided buffers and autograd adopts them as `.grad` - same bits as the default path, same storage."""
    from deblur4dgs_amd.exposure import render_exposure

    dev = torch.device("cuda:0")
    sc = make_scene(3000, 3000, 4, 3, 96, 64, seed=11)
    names = ("means", "quats", "scales", "opacities", "colors", "motion_coefs", "rots", "transls", "times", "RTs", "viewmat")

    def run(arena):
        L = {k: sc[k].to(dev).clone().requires_grad_() for k in names}
        res = render_exposure(L["means"], L["quats"], L["scales"], L["opacities"], L["colors"], 3, L["motion_coefs"],
                              L["rots"], L["transls"], L["times"], L["RTs"], L["viewmat"], sc["K"].to(dev), 96, 64,
                              return_depth=True, grad_arena=arena)
        (res["blended"].square().sum() + res["acc"].sum()).backward()
        return L

    ref = run(None)
    flat = torch.zeros(sum(sc[k].numel() for k in names), device=dev)
    arena, off = {}, 0
    for k in names:
        arena[k] = flat[off:off + sc[k].numel()].view(sc[k].shape)
        off += sc[k].numel()
    got = run(arena)
    for k in names:
        assert got[k].grad.data_ptr() == arena[k].data_ptr(), k
        assert torch.equal(got[k].grad, ref[k].grad), k


def test_leaves_are_version_checked_between_forward_and_backward():
    """the projection backward re-reads the leaf parameters; they travel through save_for_backward, so an
    in-place update after the forward (optimizer.step, a control step) raises in backward, as stock autograd / gsplat
    would, instead of silently differentiating the modified values.  (Writes through `.data`, like the reference's
    `reset_opacities`, bypass every version counter - here as in stock PyTorch.)"""
    from deblur4dgs_amd.exposure import render_exposure

    dev = torch.device("cuda:0")
    sc = make_scene(2000, 1200, 4, 2, 96, 64, seed=3)
    names = ("means", "quats", "scales", "opacities", "colors", "motion_coefs", "rots", "transls", "times", "RTs", "viewmat")
    for victim in ("opacities", "rots", "means"):
        L = {k: sc[k].to(dev).clone().requires_grad_() for k in names}
        res = render_exposure(L["means"], L["quats"], L["scales"], L["opacities"], L["colors"], 3, L["motion_coefs"], L["rots"],
                              L["transls"], L["times"], L["RTs"], L["viewmat"], sc["K"].to(dev), 96, 64, return_depth=True)
        with torch.no_grad():
            L[victim].add_(0.1)
        with pytest.raises(RuntimeError, match="modified by an inplace operation"):
            res["blended"].sum().backward()


@pytest.mark.parametrize("S,H,W,Cn", [(5, 21, 30, 17), (5, 22, 32, 17), (1, 16, 16, 4), (8, 24, 36, 5), (3, 7, 9, 3)])
def test_blend_kernels_match_the_reference_blend(S, H, W, Cn):
    """d4gs_blend_fwd / _bwd against the literal restatement of scene_model.py:386-397 (oracle.scene.blend_exposure, fp64):
    both code paths - 16-byte words when the pixel count is a multiple of 4, scalar otherwise - incl. the
    max{raw_0..raw_{S-2}, mean} quirk, exact ties, S = 1, and the gradient routing to the first winner."""
    from deblur4dgs_amd.exposure import BlendFn, reference_policy
    from oracle import scene as oscene

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(S * 100 + Cn)
    renders = torch.rand(S, H, W, Cn, generator=g)
    if Cn > 3:
        renders[:, 0, 0, 3] = 0.0          # exact ties on the max-policy channel
        renders[S - 1, 1, 1, 3] = 5.0      # the last sub-sample holds the max: the reference sees the mean there
    alphas = torch.rand(S, H, W, generator=g)
    wb, wa = torch.randn(H, W, Cn, generator=g), torch.randn(H, W, generator=g)
    r_ref, a_ref = renders.double().requires_grad_(), alphas.double().requires_grad_()
    out_ref, acc_ref, _ = oscene.blend_exposure([r_ref[s][None] for s in range(S)], [a_ref[s][None] for s in range(S)],
                                                single=(S == 1))
    ((out_ref[0] * wb.double()).sum() + (acc_ref[0] * wa.double()).sum()).backward()
    r, a = renders.to(dev).requires_grad_(), alphas.to(dev).requires_grad_()
    out, acc = BlendFn.apply(r, a, reference_policy(Cn))
    ((out * wb.to(dev)).sum() + (acc * wa.to(dev)).sum()).backward()
    torch.cuda.synchronize()
    for name, got, want in (("out", out, out_ref[0]), ("acc", acc, acc_ref[0]), ("v_renders", r.grad, r_ref.grad),
                            ("v_alphas", a.grad, a_ref.grad)):
        assert float((got.detach().cpu().double() - want.detach()).abs().max()) <= 2e-6 * max(1.0, float(want.abs().max())), name
    # the routing is exact: a max / min channel's gradient lands on ONE sub-sample (or is spread as the mean's)
    assert torch.equal(r.grad.cpu() != 0, r_ref.grad != 0)


def test_backward_builds_its_own_bases_table_when_the_caller_gives_none():
    """D4gsProjOut.blend_bases is optional (include/d4gs.h): a v304 caller leaves it NULL and d4gs_project_bwd blends
    the table behind its partials.  Same bits either way."""
    from deblur4dgs_amd.exposure import render_exposure

    N, G, K, S, W, H = 1300, 800, 6, 4, 80, 48
    sc = make_scene(N, G, K, S, W, H, seed=77, dtype=torch.float32, cam_jitter=0.01)
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    w = torch.randn(H, W, 4, generator=g).to(dev)
    grads = []
    for drop in (False, True):
        L = {k: sc[k].to(dev).clone().requires_grad_() for k in ("means", "quats", "scales", "opacities", "colors", "rots",
                                                                 "transls", "times", "RTs")}
        coefs = sc["motion_coefs"].to(dev).clone().requires_grad_()
        res = render_exposure(L["means"], L["quats"], L["scales"], L["opacities"], L["colors"], 3, coefs, L["rots"], L["transls"],
                              L["times"], L["RTs"], sc["viewmat"].to(dev), sc["K"].to(dev), W, H, return_depth=True)
        if drop:
            assert res["state"].proj_out["blend_bases"] is not None
            res["state"].proj_out["blend_bases"] = None
        (res["blended"] * w).sum().backward()
        torch.cuda.synchronize()
        grads.append({**{k: v.grad.clone() for k, v in L.items()}, "motion_coefs": coefs.grad.clone()})
    for k in grads[0]:
        assert grads[0][k].abs().max() > 0, k
        assert torch.equal(grads[0][k], grads[1][k]), k
