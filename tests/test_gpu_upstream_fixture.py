"""The HIP path against golden vectors made by the REAL gsplat 1.1.1 / roma 1.5.0 / pypose 0.6.8 (tests/golden/gen_upstream_fixture.py).

Neither library exists in the build container or on the MI355X box, so the three fixture files can only be produced on a machine with
the reference's own environment; until someone commits them these tests SKIP and rows a4 / a7 / a12 / c2 of SURVEY 8 stay "parity
unpinned" (the oracle's restatement of those libraries is anchored by closed forms and fp64 autograd only).  With the files present
the comparison is the one `north_star` names: rendered values and gradients within 1e-4 relative of the reference CUDA path, with
the documented flip allowance for elements that sit on a discrete decision (alpha >= 1/255, T <= 1e-4, ceil of the radius)."""
import os

import numpy as np
import pytest
import torch

from tests.util import check

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL, FLIPS = 1e-4, 2e-3


def _load(name):
    # (D4GS_UPSTREAM_DIR: a scratch directory with files of the same layout - scripts/mock_upstream_fixture.py writes oracle-made ones
    # there to exercise this file's mechanics on the GPU box; it pins nothing and is never the committed location)
    path = os.path.join(os.environ.get("D4GS_UPSTREAM_DIR", GOLD), name)
    if not os.path.exists(path):
        pytest.skip(f"{name} not generated yet: run tests/golden/gen_upstream_fixture.py where gsplat / roma / pypose are installed")
    return np.load(path)


def test_rasterization_matches_real_gsplat():
    """K-U1: every case of upstream_gsplat.npz through seam S1 (deblur4dgs_amd.rasterization, flow3d/scene_model.py:360-373)."""
    from deblur4dgs_amd.rasterization import rasterization
    from tests.golden.gen_upstream_fixture import RASTER_CASES, static_inputs

    d = _load("upstream_gsplat.npz")
    dev = torch.device("cuda:0")
    regen = {c[0]: c for c in RASTER_CASES}
    for case in [str(c) for c in d["cases"]]:
        get = lambda k: d[f"{case}|{k}"]
        W, H, mode = int(get("W")), int(get("H")), str(get("mode"))
        if case in regen:  # the stored inputs ARE the seeded scene (a drift of synth.py must not pass silently)
            _, N, W0, H0, seed, D, mul, _, otweak = regen[case]
            want = static_inputs(N, W0, H0, seed, D, mul)
            if otweak is None:
                for k in ("means", "quats", "scales", "opac", "colors"):
                    np.testing.assert_allclose(get("in_" + k), want[k].float().numpy(), rtol=1e-6, atol=1e-7)
        t = {k: torch.from_numpy(get("in_" + k)).to(dev).requires_grad_(k != "K") for k in ("means", "quats", "scales", "opac", "colors", "V", "K")}
        rc, ra, info = rasterization(t["means"], t["quats"], t["scales"], t["opac"], t["colors"], t["V"][None], t["K"][None], W, H,
                                     backgrounds=torch.from_numpy(get("bg")).to(dev)[None], render_mode=mode)
        info["means2d"].retain_grad()
        ((rc * torch.from_numpy(get("w_c")).to(dev)).sum() + (ra * torch.from_numpy(get("w_a")).to(dev)).sum()).backward()
        torch.cuda.synchronize()
        name = f"upstream gsplat {case}"
        check(name, "image", rc.detach().cpu(), torch.from_numpy(get("image")), TOL, FLIPS)
        check(name, "alpha", ra.detach().cpu(), torch.from_numpy(get("alpha")), TOL, FLIPS)
        radii_ref = torch.from_numpy(get("radii")).reshape(-1)
        radii = info["radii"].reshape(-1).cpu()
        vis = radii_ref > 0
        assert float(((radii > 0) == vis).float().mean()) >= 0.999, case  # visibility (a cull decision at fp32 rounding may flip one)
        both = vis & (radii > 0)
        assert float((radii[both] == radii_ref[both]).float().mean()) >= 0.995, case  # ceil(3 sqrt(lambda)): one pixel at a rounding edge
        for key, got in (("means2d", info["means2d"]), ("depths", info["depths"]), ("conics", info["conics"])):
            ref = torch.from_numpy(get(key))
            ref = ref.reshape(-1, *ref.shape[2:]) if ref.dim() > 2 else ref.reshape(-1)
            g = got.detach().cpu().reshape(ref.shape)
            check(name, key, g[both], ref[both], TOL, FLIPS)
        check(name, "means2d.grad", info["means2d"].grad.cpu().reshape(-1, 2), torch.from_numpy(get("g_means2d")).reshape(-1, 2), TOL, FLIPS)
        for k in ("means", "quats", "scales", "opac", "colors"):
            check(name, k + ".grad", t[k].grad.cpu(), torch.from_numpy(get("g_" + k)), TOL, FLIPS)
        check(name, "viewmat.grad", t["V"].grad.cpu()[:3], torch.from_numpy(get("g_V"))[:3], 4 * TOL, 0.0)


def test_pose_compose_matches_real_roma():
    """K-U2: compute_poses_fg's quaternion (flow3d/scene_model.py:94-101) through d4gs_poses_fwd.  The rotations enter as motion
    bases (K one-hot ACTIVATED coefficients select basis k, whose 6-D representation is the first two columns of R: Gram-Schmidt of an
    orthonormal pair returns R), so the device path runs its own rotmat -> quaternion -> product -> normalise chain on them."""
    from deblur4dgs_amd import engine

    d = _load("upstream_roma.npz")
    dev = torch.device("cuda:0")
    R, q_raw, want = torch.from_numpy(d["R"]), torch.from_numpy(d["q_raw"]), torch.from_numpy(d["composed_wxyz"])
    n, Kc = R.shape[0], 16
    got = []
    for a in range(0, n, Kc):
        Rk = R[a:a + Kc]
        k = Rk.shape[0]
        rots = torch.cat([Rk[:, :, 0], Rk[:, :, 1]], -1).view(k, 1, 6)  # cont_6d: [first column | second column]
        transls = torch.zeros(k, 1, 3)
        coefs = torch.eye(k)
        _, quats, _ = engine.poses(torch.zeros(k, 3, device=dev), q_raw[a:a + k].to(dev), coefs.to(dev), rots.to(dev), transls.to(dev),
                                   torch.zeros(1, device=dev), want=(False, True, False), raw_coefs=False)
        got.append(quats[:, 0].cpu())
    got = torch.cat(got, 0)
    sign = torch.sign((got * want).sum(-1, keepdim=True))
    # q and -q are the same rotation; roma's sign is part of what the reference feeds gsplat (which normalises and squares), so the
    # product keeps roma's convention away from the angle ~ pi branch points and is compared up to sign only there
    near_pi = torch.arange(n) < 64
    assert bool((sign[~near_pi] > 0).all()), "quaternion sign convention differs from roma's away from the pi branch"
    err = (got * sign - want).abs().amax(-1)
    assert float(err[~near_pi].max()) <= 1e-5 and float(err[~near_pi].median()) <= 2e-7  # (fp32 rotmat -> quaternion of both sides)
    assert float(err[near_pi].max()) <= 5e-5  # angle within 1e-4 of pi: w = sqrt(1 + trace) / 2 ~ 5e-5 is formed from a cancelling fp32 sum


def test_camera_path_matches_real_pypose():
    """K-U3: se3.Exp -> linear_interpolation -> SE3.Log -> se3_to_SE3 (move_model.py:145-147) through d4gs_camera_path_fwd, and - if the
    fixture was made with --reference - the reference's own MoveModel (F5 weights) through the product MoveModel."""
    from deblur4dgs_amd import move_model as mm

    d = _load("upstream_pypose.npz")
    dev = torch.device("cuda:0")
    S = int(d["S"])
    d0, d1 = torch.from_numpy(d["d0"]), torch.from_numpy(d["d1"])
    if "c0_RTs" in d.files:
        tp = torch.full((1, 8), 0.5, device=dev)
        for i in range(d0.shape[0]):
            RTs, _, _ = mm.CameraPathFn.apply(d0[i:i + 1].to(dev), d1[i:i + 1].to(dev), tp, S, 0, 0.0, False)
            np.testing.assert_allclose(RTs.cpu().numpy(), d[f"c{i}_RTs"].reshape(S, 3, 4), rtol=0, atol=3e-6, err_msg=f"case {i}")
    if "mm_cases" in d.files:
        f5 = np.load(os.path.join(GOLD, "f5_move_model.npz"))
        m = mm.MoveModel(num_fg=7)
        m.load_state_dict({k[3:]: torch.from_numpy(f5[k]) for k in f5.files if k.startswith("sd_")}, strict=True)
        m = m.to(dev)
        for c in range(int(d["mm_cases"])):
            R, T, t = torch.from_numpy(f5[f"c{c}_R"]).to(dev), torch.from_numpy(f5[f"c{c}_T"]).to(dev), float(f5[f"c{c}_t"])
            stage = "first" if int(f5[f"c{c}_stage"]) == 1 else "second"
            RTs, times, dT = m.forward_start_end_mid({"R": R, "T": T, "timestep": t}, num_cameras=S, mode="uniform", stage=stage)
            np.testing.assert_allclose(RTs.detach().cpu().numpy(), d[f"mm{c}_RTs"], rtol=0, atol=3e-6)
            np.testing.assert_allclose(times.detach().cpu().numpy(), d[f"mm{c}_times"], rtol=0, atol=1e-6)
            np.testing.assert_allclose(dT.detach().cpu().numpy().reshape(-1), d[f"mm{c}_deltaT"].reshape(-1), rtol=0, atol=1e-7)
    if "c0_RTs" not in d.files and "mm_cases" not in d.files:
        pytest.skip("upstream_pypose.npz was generated without --reference: it holds the se3.Exp values only")
