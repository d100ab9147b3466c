"""Generate golden vectors by IMPORTING the reference's own Python (build container only).

Run:  python tests/golden/gen_golden.py          (needs /root/reference; writes tests/golden/*.npz)

The reference cannot travel to the GPU box, so the vectors (inputs + reference outputs) are committed
as small .npz fixtures and this script is the record of how they were made.  Only the importable,
CPU-runnable parts of the path are covered:

  F1  MotionBases.compute_transforms + autograd grads     flow3d/params.py:142-180
  F2  GaussianParams activations                          flow3d/params.py:39-43,70-84
  F3  cont_6d_to_rmat (incl. near-parallel inputs)        flow3d/transforms.py:41-53
  F4  SE3_to_se3 / se3_to_SE3                             flow3d/models/utils/spline_utils.py:177-215
  F5  MoveModel.forward with a fixed non-zero state_dict  flow3d/models/move_model.py:112-135
  F6  GaussianParams.densify_params / cull_params / reset_opacities   flow3d/params.py:86-118
  F7  the checkpoint layout: state_dict() of the reference's OWN SceneModel (fg + bg + motion bases + MoveModel), keys, shapes,
      dtypes and values, and of a foreground-only one                 flow3d/scene_model.py:14-36,145-160, flow3d/params.py:9-37,121-141
      (SceneModel imports gsplat / cv2 / roma at module level and calls `.cuda()` on its MoveModel: empty stub modules stand in for
      the three imports - none is called by __init__ / state_dict / init_from_state_dict - and nn.Module.cuda is a no-op here)

`roma`, `pypose`, `jaxtyping` are absent from the image; they are imported by these modules but never
called on the code paths exercised here, so empty stub modules stand in for the import statements.
gsplat (the rasterizer) cannot be imported or stubbed: no golden vectors exist for it (parity unpinned).
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _import_reference():
    for m in ("roma", "pypose", "jaxtyping"):
        sys.modules.setdefault(m, types.ModuleType(m))
    sys.modules["jaxtyping"].Float = object
    sys.modules["pypose"].LieTensor = object
    sys.path.insert(0, REF)
    from flow3d.models.move_model import MoveModel
    from flow3d.models.utils.spline_utils import SE3_to_se3, se3_to_SE3
    from flow3d.params import GaussianParams, MotionBases
    from flow3d.transforms import cont_6d_to_rmat

    return GaussianParams, MotionBases, cont_6d_to_rmat, SE3_to_se3, se3_to_SE3, MoveModel


def main():
    GaussianParams, MotionBases, cont_6d_to_rmat, SE3_to_se3, se3_to_SE3, MoveModel = _import_reference()
    torch.manual_seed(20260928)
    T = 24

    # ---- F1 ------------------------------------------------------------------------------------
    f1 = {}
    case = 0
    for G, K in ((5, 1), (5, 6), (257, 6), (33, 20)):
        for t in (3.0, 3.37, 0.0, float(T - 1), T - 1 + 0.6, -0.4, 22.999):
            gen = torch.Generator().manual_seed(1000 + case)
            rots = (torch.tensor([1.0, 0, 0, 0, 1, 0]) + 0.3 * torch.randn(K, T, 6, generator=gen)).requires_grad_()
            transls = torch.randn(K, T, 3, generator=gen).requires_grad_()
            raw_coefs = torch.randn(G, K, generator=gen).requires_grad_()
            mb = MotionBases(rots.detach().clone(), transls.detach().clone())
            gp = GaussianParams(
                torch.zeros(G, 3), torch.randn(G, 4, generator=gen), torch.zeros(G, 3), torch.zeros(G, 3),
                torch.zeros(G), motion_coefs=raw_coefs.detach().clone(),
            )
            ts = torch.tensor([[t]])
            out = mb.compute_transforms(ts, gp.get_coefs())  # (G,1,3,4)
            wgt = torch.randn(out.shape, generator=gen)
            (out * wgt).sum().backward()
            p = f"c{case}_"
            f1[p + "t"] = np.float32(t)
            f1[p + "rots"] = rots.detach().numpy()
            f1[p + "transls"] = transls.detach().numpy()
            f1[p + "raw_coefs"] = raw_coefs.detach().numpy()
            f1[p + "wgt"] = wgt.numpy()
            f1[p + "out"] = out.detach().numpy()
            f1[p + "g_rots"] = mb.params["rots"].grad.numpy()
            f1[p + "g_transls"] = mb.params["transls"].grad.numpy()
            f1[p + "g_raw_coefs"] = gp.params["motion_coefs"].grad.numpy()
            case += 1
    f1["n_cases"] = np.int64(case)
    np.savez_compressed(os.path.join(OUT, "f1_compute_transforms.npz"), **f1)

    # ---- F2 ------------------------------------------------------------------------------------
    gen = torch.Generator().manual_seed(2)
    G, K = 64, 6
    raw = dict(
        means=torch.randn(G, 3, generator=gen),
        quats=torch.randn(G, 4, generator=gen) * 3,
        scales=torch.randn(G, 3, generator=gen) - 3,
        colors=torch.randn(G, 3, generator=gen) * 2,
        opacities=torch.randn(G, generator=gen) * 3,
        motion_coefs=torch.randn(G, K, generator=gen) * 4,
    )
    raw["quats"][0] = 0.0  # zero quaternion -> F.normalize eps path
    gp = GaussianParams(**{k: v.clone() for k, v in raw.items()})
    f2 = {"raw_" + k: v.numpy() for k, v in raw.items()}
    f2.update(
        quats=gp.get_quats().detach().numpy(),
        colors=gp.get_colors().detach().numpy(),
        scales=gp.get_scales().detach().numpy(),
        opacities=gp.get_opacities().detach().numpy(),
        coefs=gp.get_coefs().detach().numpy(),
    )
    np.savez_compressed(os.path.join(OUT, "f2_activations.npz"), **f2)

    # ---- F3 ------------------------------------------------------------------------------------
    gen = torch.Generator().manual_seed(3)
    r6 = torch.randn(128, 6, generator=gen)
    r6[:8, 3:] = r6[:8, :3] * 1.5 + 1e-3 * torch.randn(8, 3, generator=gen)  # near-parallel
    r6[8:12] *= 1e-3
    r6 = r6.requires_grad_()
    Rm = cont_6d_to_rmat(r6)
    wgt = torch.randn(Rm.shape, generator=gen)
    (Rm * wgt).sum().backward()
    np.savez_compressed(
        os.path.join(OUT, "f3_cont6d.npz"), r6=r6.detach().numpy(), R=Rm.detach().numpy(), wgt=wgt.numpy(),
        g_r6=r6.grad.numpy(),
    )

    # ---- F4 ------------------------------------------------------------------------------------
    gen = torch.Generator().manual_seed(4)
    wu = torch.randn(64, 6, generator=gen)
    wu[:16] *= 1e-3  # near identity
    wu[16:24, :3] *= 2.0  # large angles (< pi after scaling mostly)
    Rt = se3_to_SE3(wu)
    back = SE3_to_se3(Rt)
    np.savez_compressed(os.path.join(OUT, "f4_se3.npz"), wu=wu.numpy(), Rt=Rt.numpy(), back=back.numpy())

    # ---- F5 ------------------------------------------------------------------------------------
    torch.manual_seed(5)
    mm = MoveModel(num_fg=7, camera_mode="linear")
    with torch.no_grad():
        for p in mm.parameters():
            p.add_(0.05 * torch.randn_like(p))
        mm.time_params.copy_(torch.tensor([[0.5, 0.03, 0.47, 1.3, -0.2, 0.5, 0.77, 0.5]]))
    sd = {k: v.detach().numpy() for k, v in mm.state_dict().items()}
    f5 = {"sd_" + k: v for k, v in sd.items()}
    gen = torch.Generator().manual_seed(55)
    poses = se3_to_SE3(0.3 * torch.randn(6, 6, generator=gen))
    case = 0
    for i in range(6):
        for stage in ("first", "second"):
            for t in (0.0, 1.0, 2.5, 3.0, 4.0, 6.0, 7.0):
                R, Tt = poses[i, :, :3], poses[i, :, 3:4]
                d0, d1, t0, t1 = mm(R, Tt, t, stage=stage)
                p = f"c{case}_"
                f5[p + "R"] = R.numpy()
                f5[p + "T"] = Tt.numpy()
                f5[p + "t"] = np.float32(t)
                f5[p + "stage"] = np.int64(1 if stage == "first" else 2)
                f5[p + "d0"] = d0.detach().numpy()
                f5[p + "d1"] = d1.detach().numpy()
                f5[p + "dT0"] = t0.detach().numpy()
                f5[p + "dT1"] = t1.detach().numpy()
                case += 1
    f5["n_cases"] = np.int64(case)
    np.savez_compressed(os.path.join(OUT, "f5_move_model.npz"), **f5)

    # ---- F6 ------------------------------------------------------------------------------------
    f6 = {}
    gen = torch.Generator().manual_seed(77)
    names = ("means", "quats", "scales", "colors", "opacities", "motion_coefs")
    for c, (G, K) in enumerate(((40, 4), (7, 0), (64, 6))):
        raw = dict(means=torch.randn(G, 3, generator=gen), quats=torch.randn(G, 4, generator=gen),
                   scales=torch.randn(G, 3, generator=gen), colors=torch.randn(G, 3, generator=gen),
                   opacities=torch.randn(G, generator=gen))
        if K > 0:
            raw["motion_coefs"] = torch.randn(G, K, generator=gen)
        split = torch.rand(G, generator=gen) < 0.3
        dup = (torch.rand(G, generator=gen) < 0.3) & ~split
        cull = torch.rand(G, generator=gen) < 0.4
        f6[f"c{c}_split"], f6[f"c{c}_dup"], f6[f"c{c}_cull"] = split.numpy(), dup.numpy(), cull.numpy()
        for k, v in raw.items():
            f6[f"c{c}_in_{k}"] = v.numpy()
        mk = lambda: GaussianParams(raw["means"].clone(), raw["quats"].clone(), raw["scales"].clone(), raw["colors"].clone(),
                                    raw["opacities"].clone(), raw["motion_coefs"].clone() if K > 0 else None)
        gp = mk()
        for k, v in gp.densify_params(split, dup).items():
            f6[f"c{c}_densify_{k}"] = v.detach().numpy()
        gp = mk()
        for k, v in gp.cull_params(cull).items():
            f6[f"c{c}_cull_{k}"] = v.detach().numpy()
        gp = mk()
        for k, v in gp.reset_opacities(torch.logit(torch.tensor(0.08))).items():
            f6[f"c{c}_reset_{k}"] = v.detach().numpy()
    f6["n_cases"] = np.int64(3)
    np.savez_compressed(os.path.join(OUT, "f6_control_params.npz"), **f6)
    # ---- F7 ------------------------------------------------------------------------------------
    gs = types.ModuleType("gsplat")
    gs.rendering = types.ModuleType("gsplat.rendering")
    gs.rendering.rasterization = None
    sys.modules.setdefault("gsplat", gs)
    sys.modules.setdefault("gsplat.rendering", gs.rendering)
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    torch.nn.Module.cuda = lambda self, *a, **k: self  # (this generator only: SceneModel.__init__ does MoveModel(...).cuda())
    from flow3d.scene_model import SceneModel

    gen = torch.Generator().manual_seed(7)
    r = lambda *sh: torch.randn(*sh, generator=gen)
    Gf, Gb, K, T, F = 9, 5, 3, 6, 4
    fg = GaussianParams(r(Gf, 3), r(Gf, 4), r(Gf, 3), r(Gf, 3), r(Gf), motion_coefs=r(Gf, K), scene_center=r(3), scene_scale=torch.tensor(1.7))
    bg = GaussianParams(r(Gb, 3), r(Gb, 4), r(Gb, 3), r(Gb, 3), r(Gb), scene_center=r(3), scene_scale=torch.tensor(2.3))
    mb = MotionBases(r(K, T, 6), r(K, T, 3))
    Ks, w2cs = r(F, 3, 3), r(F, 4, 4)
    torch.manual_seed(77)
    full = SceneModel(Ks, w2cs, fg, mb, bg)
    with torch.no_grad():
        for p_ in full.move_model.parameters():
            p_.add_(0.05 * torch.randn_like(p_))
    f7 = {}
    for tag, model in (("full", full), ("fgonly", SceneModel(Ks, w2cs, fg, mb, None))):
        sd = model.state_dict()
        f7[tag + "_keys"] = np.array(list(sd.keys()))  # in the reference's order
        for k, v in sd.items():
            f7[f"{tag}|{k}"] = v.detach().numpy()
        # and the reference's own loader accepts it (what `init_from_state_dict` needs from a checkpoint)
        again = SceneModel.init_from_state_dict({k: v.clone() for k, v in sd.items()})
        assert again.num_fg_gaussians == Gf and again.num_motion_bases == K and again.num_frames == T
        assert (again.bg is None) == (tag == "fgonly")
        # ... with its quirk: the buffers `fg.scene_center` / `fg.scene_scale` / `bg.*` are looked up under `fg.params.*` and never
        # found, so the loaded model is back at centre 0 / scale 1 (and `bg_scene_scale` at 1): recorded as `<tag>_loaded|<key>`
        for k, v in again.state_dict().items():
            if not k.startswith("move_model."):  # (a fresh random MoveModel: its weights come from ckpt["move_model"], trainer.py:126-170)
                f7[f"{tag}_loaded|{k}"] = v.detach().numpy()
    np.savez_compressed(os.path.join(OUT, "f7_state_dict.npz"), **f7)
    print("wrote fixtures to", OUT)


if __name__ == "__main__":
    main()
