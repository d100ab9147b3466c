"""Host-side generator (a12): the product's MoveModel vs the oracle's restatement (which is pinned to the
reference's own Python through tests/golden F4/F5), plus group identities for the restated pypose ops."""
import numpy as np
import torch

from deblur4dgs_amd import move_model as mm
from oracle import camera


def _model(seed=0):
    torch.manual_seed(seed)
    m = mm.MoveModel(num_fg=5)
    with torch.no_grad():
        for p in m.parameters():
            p.add_(0.05 * torch.randn_like(p))
        m.time_params.copy_(torch.tensor([[0.5, 0.03, 0.47, 1.3, -0.2, 0.5, 0.77, 0.5]]))
    return m


def test_forward_matches_golden(golden_dir):
    import os

    z = np.load(os.path.join(golden_dir, "f5_move_model.npz"))
    m = mm.MoveModel(num_fg=7)
    m.load_state_dict({k[3:]: torch.tensor(z[k]) for k in z.files if k.startswith("sd_")})
    for c in range(int(z["n_cases"])):
        p = f"c{c}_"
        stage = "first" if int(z[p + "stage"]) == 1 else "second"
        d0, d1, t0, t1 = m(torch.tensor(z[p + "R"]), torch.tensor(z[p + "T"]), float(z[p + "t"]), stage=stage)
        np.testing.assert_allclose(d0.detach().numpy(), z[p + "d0"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(d1.detach().numpy(), z[p + "d1"], rtol=1e-5, atol=1e-6)
        np.testing.assert_array_equal(t0.detach().numpy(), z[p + "dT0"])
        np.testing.assert_array_equal(t1.detach().numpy(), z[p + "dT1"])


def test_start_end_mid_matches_oracle():
    m = _model()
    sd = {k: v.detach() for k, v in m.state_dict().items()}
    pose = mm.se3_to_SE3(0.3 * torch.randn(6))
    for t in (0.0, 2.0, 3.0, 6.0):
        for stage in ("first", "second"):
            info = {"R": pose[:, :3], "T": pose[:, 3:4], "timestep": t}
            RTs, times, dT = m.forward_start_end_mid(info, num_cameras=11, mode="uniform", stage=stage)
            oR, ot, od = camera.forward_start_end_mid(sd, pose[:, :3], pose[:, 3:4], t, 11, stage)
            np.testing.assert_allclose(RTs.detach().numpy(), oR.numpy(), rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(times.detach().numpy(), ot.numpy(), rtol=1e-6, atol=1e-6)
            np.testing.assert_allclose(dT.detach().numpy(), od.numpy(), rtol=0, atol=0)
            assert RTs.shape == (11, 3, 4) and times.shape == (1, 11)


def test_zero_init_heads_give_identity_deltas():
    m = mm.MoveModel(num_fg=3)
    pose = mm.se3_to_SE3(0.2 * torch.randn(6))
    RTs, times, dT = m.forward_start_end_mid({"R": pose[:, :3], "T": pose[:, 3:4], "timestep": 3.0}, num_cameras=11)
    np.testing.assert_allclose(RTs.detach().numpy(), np.tile(np.eye(3, 4, dtype=np.float32), (11, 1, 1)), atol=1e-6)
    np.testing.assert_allclose(times[0].detach().numpy(), np.linspace(2.5, 3.5, 11), atol=1e-6)  # time_params = 0.5


def test_lie_group_identities():
    g = torch.Generator().manual_seed(0)
    xi = 0.7 * torch.randn(32, 6, generator=g, dtype=torch.float64)
    X = mm.se3_Exp(xi)
    np.testing.assert_allclose(mm.SE3_Log(X).numpy(), xi.numpy(), atol=1e-10)  # Log(Exp(xi)) = xi
    q = X[..., 3:]
    np.testing.assert_allclose(q.norm(dim=-1).numpy(), 1.0, atol=1e-12)
    e = mm.SO3_mul(q, mm.SO3_Inv(q))
    np.testing.assert_allclose(e.numpy(), np.tile([0, 0, 0, 1.0], (32, 1)), atol=1e-12)
    # slerp end points and midpoint
    a, b = X[:16], X[16:]
    Y = mm.linear_interpolation(a, b, torch.tensor([0.0, 0.5, 1.0], dtype=torch.float64))
    np.testing.assert_allclose(Y[:, 0].numpy(), a.numpy(), atol=1e-10)
    sgn = torch.sign((Y[:, 2, 3:] * b[:, 3:]).sum(-1, keepdim=True))
    np.testing.assert_allclose((Y[:, 2, 3:] * sgn).numpy(), b[:, 3:].numpy(), atol=1e-9)
    np.testing.assert_allclose(Y[:, 2, :3].numpy(), b[:, :3].numpy(), atol=1e-10)
    # near-zero branch is smooth
    tiny = 1e-9 * torch.randn(4, 6, generator=g, dtype=torch.float64)
    np.testing.assert_allclose(mm.SE3_Log(mm.se3_Exp(tiny)).numpy(), tiny.numpy(), atol=1e-15)
