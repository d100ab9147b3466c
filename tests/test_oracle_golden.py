"""Pin the oracle's restatement of the importable reference code against the golden vectors
generated from the reference's own Python (tests/golden/gen_golden.py)."""
import os

import numpy as np
import torch

from oracle import camera, deform


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_f1_compute_transforms(golden_dir):
    z = _load(golden_dir, "f1_compute_transforms.npz")
    for c in range(int(z["n_cases"])):
        p = f"c{c}_"
        rots = torch.tensor(z[p + "rots"], requires_grad=True)
        transls = torch.tensor(z[p + "transls"], requires_grad=True)
        raw = torch.tensor(z[p + "raw_coefs"], requires_grad=True)
        ts = torch.tensor([[float(z[p + "t"])]])
        out = deform.compute_transforms(ts, deform.act_coefs(raw), rots, transls)
        np.testing.assert_allclose(out.detach().numpy(), z[p + "out"], rtol=1e-6, atol=1e-6)
        (out * torch.tensor(z[p + "wgt"])).sum().backward()
        np.testing.assert_allclose(rots.grad.numpy(), z[p + "g_rots"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(transls.grad.numpy(), z[p + "g_transls"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(raw.grad.numpy(), z[p + "g_raw_coefs"], rtol=1e-5, atol=1e-5)


def test_f2_activations(golden_dir):
    z = _load(golden_dir, "f2_activations.npz")
    t = lambda k: torch.tensor(z[k])
    np.testing.assert_array_equal(deform.act_quats(t("raw_quats")).numpy(), z["quats"])
    np.testing.assert_array_equal(deform.act_colors(t("raw_colors")).numpy(), z["colors"])
    np.testing.assert_array_equal(deform.act_scales(t("raw_scales")).numpy(), z["scales"])
    np.testing.assert_array_equal(deform.act_opacities(t("raw_opacities")).numpy(), z["opacities"])
    np.testing.assert_array_equal(deform.act_coefs(t("raw_motion_coefs")).numpy(), z["coefs"])


def test_f3_cont6d(golden_dir):
    z = _load(golden_dir, "f3_cont6d.npz")
    r6 = torch.tensor(z["r6"], requires_grad=True)
    R = deform.cont_6d_to_rmat(r6)
    np.testing.assert_array_equal(R.detach().numpy(), z["R"])
    (R * torch.tensor(z["wgt"])).sum().backward()
    np.testing.assert_allclose(r6.grad.numpy(), z["g_r6"], rtol=1e-6, atol=1e-6)


def test_f4_se3(golden_dir):
    z = _load(golden_dir, "f4_se3.npz")
    wu = torch.tensor(z["wu"])
    Rt = camera.se3_to_SE3(wu)
    np.testing.assert_allclose(Rt.numpy(), z["Rt"], rtol=1e-6, atol=1e-6)
    back = camera.SE3_to_se3(torch.tensor(z["Rt"]))
    np.testing.assert_allclose(back.numpy(), z["back"], rtol=1e-5, atol=1e-5)


def test_f5_move_model(golden_dir):
    z = _load(golden_dir, "f5_move_model.npz")
    sd = {k[3:]: torch.tensor(z[k]) for k in z.files if k.startswith("sd_")}
    for c in range(int(z["n_cases"])):
        p = f"c{c}_"
        stage = "first" if int(z[p + "stage"]) == 1 else "second"
        d0, d1, t0, t1 = camera.move_model_forward(
            sd, torch.tensor(z[p + "R"]), torch.tensor(z[p + "T"]), float(z[p + "t"]), stage
        )
        np.testing.assert_allclose(d0.numpy(), z[p + "d0"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(d1.numpy(), z[p + "d1"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(t0.numpy(), z[p + "dT0"], rtol=0, atol=0)
        np.testing.assert_allclose(t1.numpy(), z[p + "dT1"], rtol=0, atol=0)
