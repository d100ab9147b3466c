"""The exit path of the "parity unpinned" half of the oracle: golden vectors from the REAL third-party libraries.

THIS SCRIPT CANNOT RUN IN THE BUILD CONTAINER OR ON THE MI355X BOX (neither has gsplat / roma / pypose, nor a CUDA device, nor a
network to fetch them).  It is what a maintainer with the reference's own environment runs once:

    pip install gsplat==1.1.1 roma==1.5.0 pypose==0.6.8          # the pins of the reference's requirements.txt:137,354,385
    python tests/golden/gen_upstream_fixture.py [--reference /path/to/Deblur4DGS] [--device cuda:0]

and commits the three files it writes next to itself:

    upstream_gsplat.npz   K-U1  gsplat.rendering.rasterization(packed=False) on seeded scenes from deblur4dgs_amd/synth.py and on
                                the closed-form scenes of tests/test_gpu_known_answers.py: image, alpha, means2d, radii, depths,
                                conics, and the gradient of a fixed random linear loss w.r.t. every input and w.r.t. means2d
                                (the call the reference makes: flow3d/scene_model.py:360-373)
    upstream_roma.npz     K-U2  the quaternion compose of compute_poses_fg (flow3d/scene_model.py:94-101):
                                normalize(xyzw_to_wxyz(quat_product(rotmat_to_unitquat(R), wxyz_to_xyzw(q)))), incl. rotations by ~pi
    upstream_pypose.npz   K-U3  pypose se3.Exp -> linear_interpolation -> SE3.Log -> se3_to_SE3 as MoveModel.forward_start_end_mid runs it
                                (flow3d/models/move_model.py:138-166, flow3d/models/utils/spline_utils.py:371-408); with --reference
                                also the reference's own MoveModel loaded with the committed F5 state dict (f5_move_model.npz)

`tests/test_gpu_upstream_fixture.py` compares the HIP path with these files and SKIPS while they are absent; the day they exist,
rows a4 / a7 / a12 / c2 of SURVEY 8 stop being "parity unpinned".  Only data is written: inputs, weights, outputs.  The scenes are
regenerated from the same seeds by the test, and the stored inputs are compared first, so a drift of synth.py cannot pass silently.
"""
import argparse
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

RASTER_CASES = [
    # name, N, W, H, seed, D, scale_mul, render_mode, opacity tweak
    ("small_rgb_ed", 1200, 96, 64, 101, 3, 3.0, "RGB+ED", None),
    ("large_splats_rgb", 3000, 96, 64, 102, 3, 12.0, "RGB", None),
    ("train17_ed", 1500, 64, 48, 103, 16, 4.0, "RGB+ED", None),
    ("opaque_saturating", 4000, 64, 48, 104, 3, 10.0, "RGB+ED", 0.98),  # T crosses 1e-4 inside most tile lists
    ("near_threshold", 1500, 64, 48, 105, 3, 3.0, "RGB", 0.012),         # alphas around 1/255
]


def static_inputs(N, W, H, seed, D, scale_mul):
    """tests/util.py:static_inputs (kept in step with it: the test regenerates and compares)."""
    from deblur4dgs_amd.synth import make_scene

    sc = make_scene(N, 0, 1, 1, W, H, seed, dtype=torch.float64, D=D)
    return dict(means=sc["means"], quats=sc["quats"], scales=torch.exp(sc["scales"]) * scale_mul,
                opac=torch.sigmoid(sc["opacities"]), colors=torch.sigmoid(sc["colors"]), V=sc["viewmat"], K=sc["K"])


def known_answer_scenes():
    """The closed-form scenes (isotropic Gaussians, identity view): a pixel-centre splat, two overlapping splats, an alpha just above /
    below 1/255, a stack whose transmittance crosses 1e-4, splats at the near plane and outside the 1.3 x fov clamp."""
    f, W, H = 60.0, 48, 32
    K = torch.tensor([[f, 0, W / 2], [0, f, H / 2], [0, 0, 1]], dtype=torch.float64)
    out = {}

    def scene(means, s, o, colors):
        n = len(means)
        return dict(means=torch.tensor(means, dtype=torch.float64), quats=torch.tensor([[1.0, 0, 0, 0]] * n, dtype=torch.float64),
                    scales=torch.tensor(s, dtype=torch.float64).view(n, 1).expand(n, 3).contiguous(), opac=torch.tensor(o, dtype=torch.float64),
                    colors=torch.tensor(colors, dtype=torch.float64), V=torch.eye(4, dtype=torch.float64), K=K, W=W, H=H)

    z = 4.0
    cx = lambda px: (px + 0.5 - W / 2) * z / f
    cy = lambda py: (py + 0.5 - H / 2) * z / f
    out["ka_single"] = scene([[cx(20), cy(12), z]], [0.25], [0.8], [[0.9, 0.3, 0.1]])
    out["ka_two_overlap"] = scene([[cx(20), cy(12), 3.0], [cx(21) * 5.0 / z, cy(13) * 5.0 / z, 5.0]], [0.2, 0.4], [0.6, 0.9],
                                  [[1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])
    for tag, o in (("above", 1.0 / 255.0 * 1.02), ("below", 1.0 / 255.0 * 0.98)):
        out[f"ka_alpha_{tag}"] = scene([[cx(20), cy(12), z]], [0.25], [o], [[0.5, 0.5, 0.5]])
    n = 12
    out["ka_T_1e-4"] = scene([[cx(20), cy(12), 2.0 + 0.25 * i] for i in range(n)], [0.3] * n, [0.9] * n,
                             [[(i % 3 == 0) * 1.0, (i % 3 == 1) * 1.0, (i % 3 == 2) * 1.0] for i in range(n)])
    out["ka_near_and_fov"] = scene([[0.0, 0.0, 0.0099], [0.0, 0.0, 0.0101], [3.5 * z * (0.5 * W / f), 0.0, z], [cx(40), cy(20), z]],
                                   [0.002, 0.002, 0.6, 0.3], [0.9, 0.9, 0.9, 0.9], [[1, 1, 1], [1, 0, 1], [0, 1, 0], [0, 1, 1]])
    return out


def run_gsplat(inp, W, H, mode, dev, seed, ed_weight=True):
    from gsplat.rendering import rasterization

    t = {k: v.to(dev, torch.float32).clone().requires_grad_(k != "K") for k, v in inp.items() if torch.is_tensor(v)}
    D = t["colors"].shape[-1]
    bg = torch.linspace(0.1, 0.9, D, device=dev)
    rc, ra, info = rasterization(t["means"], t["quats"], t["scales"], t["opac"], t["colors"], t["V"][None], t["K"][None], W, H,
                                 packed=False, backgrounds=bg[None], render_mode=mode)
    info["means2d"].retain_grad()
    g = torch.Generator().manual_seed(seed)
    w_c = torch.randn(rc.shape, generator=g).to(dev)
    if not ed_weight:  # one-to-twelve-splat scenes: d(sum w z / alpha) / d alpha cancels analytically and 1 / alpha ~ 250 amplifies the
        w_c[..., D:] = 0.0  # fp32 residue into the opacity gradient; the ED VALUES are still compared, its gradient on the seeded scenes
    w_a = torch.randn(ra.shape, generator=g).to(dev)
    ((rc * w_c).sum() + (ra * w_a).sum()).backward()
    torch.cuda.synchronize()
    out = {"in_" + k: v.detach().cpu().numpy() for k, v in t.items()}
    out.update(W=np.int64(W), H=np.int64(H), mode=np.array(mode), bg=bg.cpu().numpy(), w_c=w_c.cpu().numpy(), w_a=w_a.cpu().numpy(),
               image=rc.detach().cpu().numpy(), alpha=ra.detach().cpu().numpy(), means2d=info["means2d"].detach().cpu().numpy(),
               radii=info["radii"].cpu().numpy(), depths=info["depths"].detach().cpu().numpy(),
               conics=info["conics"].detach().cpu().numpy(), g_means2d=info["means2d"].grad.cpu().numpy())
    for k in ("means", "quats", "scales", "opac", "colors", "V"):
        out["g_" + k] = t[k].grad.cpu().numpy()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default=None, help="a checkout of ZcsrenlongZ/Deblur4DGS (optional: adds the reference's own MoveModel run)")
    ap.add_argument("--device", default="cuda:0")
    args = ap.parse_args()
    import gsplat
    import pypose as pp
    import roma

    dev = torch.device(args.device)
    versions = np.array([f"gsplat {gsplat.__version__}", f"roma {getattr(roma, '__version__', '?')}", f"pypose {pp.__version__}",
                         f"torch {torch.__version__}"])

    # ---- K-U1 gsplat -------------------------------------------------------------------------------------------------------
    f = {"versions": versions, "cases": np.array([c[0] for c in RASTER_CASES] + list(known_answer_scenes()))}
    for name, N, W, H, seed, D, mul, mode, otweak in RASTER_CASES:
        inp = static_inputs(N, W, H, seed, D, mul)
        if otweak is not None:
            inp["opac"] = torch.full_like(inp["opac"], otweak) if otweak > 0.5 else inp["opac"] * otweak / inp["opac"].mean()
        for k, v in run_gsplat(inp, W, H, mode, dev, seed).items():
            f[f"{name}|{k}"] = v
    for name, sc in known_answer_scenes().items():
        for k, v in run_gsplat(sc, sc["W"], sc["H"], "RGB+ED", dev, 7, ed_weight=False).items():
            f[f"{name}|{k}"] = v
    np.savez_compressed(os.path.join(HERE, "upstream_gsplat.npz"), **f)

    # ---- K-U2 roma ---------------------------------------------------------------------------------------------------------
    g = torch.Generator().manual_seed(202)
    n = 512
    ax = torch.nn.functional.normalize(torch.randn(n, 3, generator=g, dtype=torch.float64), dim=-1)
    ang = torch.rand(n, generator=g, dtype=torch.float64) * 2 * math.pi
    ang[:64] = math.pi - 1e-4 * torch.rand(64, generator=g, dtype=torch.float64)  # the branchy corner of rotmat_to_unitquat
    ang[64:96] = 1e-5 * torch.rand(32, generator=g, dtype=torch.float64)
    Rm = roma.rotvec_to_rotmat((ax * ang[:, None]).float())
    q = torch.randn(n, 4, generator=g)  # raw wxyz leaves
    qn = torch.nn.functional.normalize(q, p=2, dim=-1)
    rq = roma.rotmat_to_unitquat(Rm)
    out = roma.quat_xyzw_to_wxyz(roma.quat_product(rq, roma.quat_wxyz_to_xyzw(qn)))
    out = torch.nn.functional.normalize(out, p=2, dim=-1)
    np.savez_compressed(os.path.join(HERE, "upstream_roma.npz"), versions=versions, R=Rm.numpy(), q_raw=q.numpy(), rotmat_to_unitquat_xyzw=rq.numpy(),
                        composed_wxyz=out.numpy())

    # ---- K-U3 pypose -------------------------------------------------------------------------------------------------------
    f = {"versions": versions}
    g = torch.Generator().manual_seed(303)
    d0 = 0.02 * torch.randn(24, 6, generator=g)
    d1 = 0.02 * torch.randn(24, 6, generator=g)
    d0[:4], d1[:4] = 0.0, 0.0       # the zero-initialised heads of a fresh MoveModel
    d0[4:8] *= 40.0                 # large rotations
    S = 11
    mm0 = None
    if args.reference:
        sys.path.insert(0, args.reference)
        from flow3d.models.move_model import MoveModel

        mm0 = MoveModel(num_fg=7, camera_mode="linear")
    for i in range(d0.shape[0]):
        s0, s1 = pp.se3(d0[i:i + 1]).Exp(), pp.se3(d1[i:i + 1]).Exp()
        f[f"c{i}_exp0"], f[f"c{i}_exp1"] = s0.tensor().numpy(), s1.tensor().numpy()  # [1,7] tx ty tz qx qy qz qw
        if mm0 is not None:  # the three lines of MoveModel.forward_start_end_mid (move_model.py:145-147), on the reference's own methods
            mid = mm0._interpolate(s0, s1, num_cameras=S, mode="uniform")
            log = mid.Log()
            f[f"c{i}_interp"], f[f"c{i}_log"] = mid.tensor().numpy(), log.tensor().numpy()
            f[f"c{i}_RTs"] = mm0.postprocessPose(log).squeeze(0).detach().numpy()  # [S,3,4]
    f["d0"], f["d1"], f["S"] = d0.numpy(), d1.numpy(), np.int64(S)
    if args.reference:
        f5 = np.load(os.path.join(HERE, "f5_move_model.npz"))
        mm = MoveModel(num_fg=7, camera_mode="linear")
        mm.load_state_dict({k[3:]: torch.from_numpy(f5[k]) for k in f5.files if k.startswith("sd_")})
        mm = mm.to(dev)
        for c in range(int(f5["n_cases"])):
            R, T, t = torch.from_numpy(f5[f"c{c}_R"]).to(dev), torch.from_numpy(f5[f"c{c}_T"]).to(dev), float(f5[f"c{c}_t"])
            stage = "first" if int(f5[f"c{c}_stage"]) == 1 else "second"
            RTs, times, dT = mm.forward_start_end_mid({"R": R, "T": T, "timestep": t}, num_cameras=S, mode="uniform", stage=stage)
            f[f"mm{c}_RTs"], f[f"mm{c}_times"], f[f"mm{c}_deltaT"] = RTs.detach().cpu().numpy(), times.detach().cpu().numpy(), dT.detach().cpu().numpy()
        f["mm_cases"] = f5["n_cases"]
    np.savez_compressed(os.path.join(HERE, "upstream_pypose.npz"), **f)
    print("wrote upstream_gsplat.npz, upstream_roma.npz, upstream_pypose.npz to", HERE, "- versions:", list(versions))


if __name__ == "__main__":
    main()
