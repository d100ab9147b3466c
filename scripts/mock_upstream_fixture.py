"""MOCK of tests/golden/gen_upstream_fixture.py for exercising tests/test_gpu_upstream_fixture.py's mechanics on the GPU box.

The real generator needs gsplat / roma / pypose + CUDA and cannot run here.  This one writes files of the SAME layout from the oracle
(oracle/raster.py, oracle/deform.py, oracle/camera.py, fp64) into a scratch directory:

    python scripts/mock_upstream_fixture.py gpurun_out/mock_upstream
    D4GS_UPSTREAM_DIR=gpurun_out/mock_upstream python -m pytest tests/test_gpu_upstream_fixture.py -q

It pins NOTHING (oracle vs product is what every other parity test already does) and its output must never be committed under
tests/golden/: it only shows that the consumer test reads the layout, runs every comparison and passes on faithful data."""
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import camera as ocam  # noqa: E402
from oracle import deform, raster  # noqa: E402
from tests.golden import gen_upstream_fixture as G  # noqa: E402


def run_oracle(inp, W, H, mode, seed, ed_weight=True):
    t = {k: v.double().clone().requires_grad_(k != "K") for k, v in inp.items() if torch.is_tensor(v)}
    D = t["colors"].shape[-1]
    bg = torch.linspace(0.1, 0.9, D, dtype=torch.float64)
    rc, ra, info = raster.rasterization(t["means"], t["quats"], t["scales"], t["opac"], t["colors"], t["V"], t["K"], W, H, background=bg,
                                        render_mode=mode)
    info["means2d"].retain_grad()
    g = torch.Generator().manual_seed(seed)
    w_c = torch.randn((1, *rc.shape), generator=g)
    if not ed_weight:
        w_c[..., D:] = 0.0
    w_a = torch.randn((1, *ra.shape), generator=g)
    ((rc[None] * w_c.double()).sum() + (ra[None] * w_a.double()).sum()).backward()
    f32 = lambda x: x.detach().float().numpy()
    out = {"in_" + k: f32(v) for k, v in t.items()}
    out.update(W=np.int64(W), H=np.int64(H), mode=np.array(mode), bg=f32(bg), w_c=w_c.numpy(), w_a=w_a.numpy(), image=f32(rc[None]), alpha=f32(ra[None]),
               means2d=f32(info["means2d"][None]), radii=info["radii"][None].int().numpy(), depths=f32(info["depths"][None]),
               conics=f32(info["conics"][None]), g_means2d=f32(info["means2d"].grad[None]))
    for k in ("means", "quats", "scales", "opac", "colors", "V"):
        out["g_" + k] = f32(t[k].grad)
    return out


def main(out_dir):
    os.makedirs(out_dir, exist_ok=True)
    versions = np.array(["MOCK: oracle restatement, not gsplat / roma / pypose"])
    f = {"versions": versions, "cases": np.array([c[0] for c in G.RASTER_CASES] + list(G.known_answer_scenes()))}
    for name, N, W, H, seed, D, mul, mode, otweak in G.RASTER_CASES:
        inp = G.static_inputs(N, W, H, seed, D, mul)
        if otweak is not None:
            inp["opac"] = torch.full_like(inp["opac"], otweak) if otweak > 0.5 else inp["opac"] * otweak / inp["opac"].mean()
        for k, v in run_oracle(inp, W, H, mode, seed).items():
            f[f"{name}|{k}"] = v
    for name, sc in G.known_answer_scenes().items():
        for k, v in run_oracle(sc, sc["W"], sc["H"], "RGB+ED", 7, ed_weight=False).items():
            f[f"{name}|{k}"] = v
    np.savez_compressed(os.path.join(out_dir, "upstream_gsplat.npz"), **f)

    g = torch.Generator().manual_seed(202)
    n = 512
    ax = torch.nn.functional.normalize(torch.randn(n, 3, generator=g, dtype=torch.float64), dim=-1)
    ang = torch.rand(n, generator=g, dtype=torch.float64) * 2 * math.pi
    ang[:64] = math.pi - 1e-4 * torch.rand(64, generator=g, dtype=torch.float64)
    ang[64:96] = 1e-5 * torch.rand(32, generator=g, dtype=torch.float64)
    Rm = ocam.se3_to_SE3(torch.cat([ax * ang[:, None], torch.zeros(n, 3, dtype=torch.float64)], -1))[:, :, :3].float()
    q = torch.randn(n, 4, generator=g)
    qn = torch.nn.functional.normalize(q.double(), p=2, dim=-1)
    rq = deform.rotmat_to_unitquat_xyzw(Rm.double())
    out = torch.nn.functional.normalize(deform.quat_xyzw_to_wxyz(deform.quat_product_xyzw(rq, deform.quat_wxyz_to_xyzw(qn))), p=2, dim=-1)
    np.savez_compressed(os.path.join(out_dir, "upstream_roma.npz"), versions=versions, R=Rm.numpy(), q_raw=q.numpy(),
                        rotmat_to_unitquat_xyzw=rq.float().numpy(), composed_wxyz=out.float().numpy())

    f = {"versions": versions}
    g = torch.Generator().manual_seed(303)
    d0 = 0.02 * torch.randn(24, 6, generator=g)
    d1 = 0.02 * torch.randn(24, 6, generator=g)
    d0[:4], d1[:4] = 0.0, 0.0
    d0[4:8] *= 40.0
    S = 11
    u = torch.linspace(0, 1, S, dtype=torch.float64)
    for i in range(d0.shape[0]):
        P0, P1 = ocam.se3_exp(d0[i:i + 1].double()), ocam.se3_exp(d1[i:i + 1].double())
        X = ocam.linear_interpolation(P0, P1, u)
        f[f"c{i}_exp0"], f[f"c{i}_exp1"] = P0.float().numpy(), P1.float().numpy()
        f[f"c{i}_RTs"] = ocam.se3_to_SE3(ocam.SE3_log(X))[0].float().numpy()
    f["d0"], f["d1"], f["S"] = d0.numpy(), d1.numpy(), np.int64(S)
    f5 = np.load(os.path.join(ROOT, "tests", "golden", "f5_move_model.npz"))
    sd = {k[3:]: torch.from_numpy(f5[k]).double() for k in f5.files if k.startswith("sd_")}
    for c in range(int(f5["n_cases"])):
        R, T, t = torch.from_numpy(f5[f"c{c}_R"]).double(), torch.from_numpy(f5[f"c{c}_T"]).double(), float(f5[f"c{c}_t"])
        stage = "first" if int(f5[f"c{c}_stage"]) == 1 else "second"
        RTs, times, dT = ocam.forward_start_end_mid(sd, R, T, t, S, stage)
        f[f"mm{c}_RTs"], f[f"mm{c}_times"], f[f"mm{c}_deltaT"] = RTs.float().numpy(), times.float().numpy(), dT.float().numpy()
    f["mm_cases"] = f5["n_cases"]
    np.savez_compressed(os.path.join(out_dir, "upstream_pypose.npz"), **f)
    print("MOCK upstream fixtures (oracle-made) written to", out_dir)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "mock_upstream"))
