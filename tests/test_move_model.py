"""Host-side generator (a12) without a GPU.

The arithmetic lives in csrc/camera.hip (GPU tests: tests/test_gpu_camera_path.py).  Here: the oracle's torch
restatement (oracle/camera.py, pinned to the reference's own Python through tests/golden F4/F5 in
tests/test_oracle_golden.py) satisfies the group identities of the pypose ops it restates, and the product module
keeps the reference's parameter layout and refuses CPU tensors (no eager fallback)."""
import os

import numpy as np
import pytest
import torch

from deblur4dgs_amd import move_model as mm
from oracle import camera


def test_product_module_loads_the_reference_state_dict_layout(golden_dir):
    z = np.load(os.path.join(golden_dir, "f5_move_model.npz"))
    sd = {k[3:]: torch.tensor(z[k]) for k in z.files if k.startswith("sd_")}  # keys written by the reference's module
    m = mm.MoveModel(num_fg=7)
    missing, unexpected = m.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    assert sum(p.numel() for p in m.parameters()) == 30036
    assert [tuple(p.shape) for p in m._layer_params()[:4]] == [(64, 66), (64,), (64, 64), (64,)]


def test_product_has_no_cpu_path():
    m = mm.MoveModel(num_fg=3)
    pose = camera.se3_to_SE3(0.2 * torch.randn(6))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.forward_start_end_mid({"R": pose[:, :3], "T": pose[:, 3:4], "timestep": 3.0}, num_cameras=11)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(pose[:, :3], pose[:, 3:4], 3.0)
    src = open(mm.__file__).read()
    for name in ("def se3_to_SE3", "def SE3_to_se3", "def linear_interpolation", "def taylor_A", "def so3_Exp"):
        assert name not in src, f"{name}: the SE(3) chain belongs to csrc/camera.hip (product) and oracle/ (checker)"


def test_oracle_zero_init_heads_give_identity_deltas():
    m = mm.MoveModel(num_fg=3)  # zero-initialised heads (move_model.py:99-102)
    sd = {k: v.detach() for k, v in m.state_dict().items()}
    pose = camera.se3_to_SE3(0.2 * torch.randn(6))
    RTs, times, dT = camera.forward_start_end_mid(sd, pose[:, :3], pose[:, 3:4], 3.0, 11, "second")
    np.testing.assert_allclose(RTs.numpy(), np.tile(np.eye(3, 4, dtype=np.float32), (11, 1, 1)), atol=1e-6)
    np.testing.assert_allclose(times[0].numpy(), np.linspace(2.5, 3.5, 11), atol=1e-6)  # time_params = 0.5
    assert RTs.shape == (11, 3, 4) and times.shape == (1, 11) and dT.shape == (1, 1)


def test_oracle_lie_group_identities():
    g = torch.Generator().manual_seed(0)
    xi = 0.7 * torch.randn(32, 6, generator=g, dtype=torch.float64)
    X = camera.se3_exp(xi)
    np.testing.assert_allclose(camera.SE3_log(X).numpy(), xi.numpy(), atol=1e-10)  # Log(Exp(xi)) = xi
    q = X[..., 3:]
    np.testing.assert_allclose(q.norm(dim=-1).numpy(), 1.0, atol=1e-12)
    e = camera.quat_mul(q, camera.quat_inv(q))
    np.testing.assert_allclose(e.numpy(), np.tile([0, 0, 0, 1.0], (32, 1)), atol=1e-12)
    # slerp end points and midpoint
    a, b = X[:16], X[16:]
    Y = camera.linear_interpolation(a, b, torch.tensor([0.0, 0.5, 1.0], dtype=torch.float64))
    np.testing.assert_allclose(Y[:, 0].numpy(), a.numpy(), atol=1e-10)
    sgn = torch.sign((Y[:, 2, 3:] * b[:, 3:]).sum(-1, keepdim=True))
    np.testing.assert_allclose((Y[:, 2, 3:] * sgn).numpy(), b[:, 3:].numpy(), atol=1e-9)
    np.testing.assert_allclose(Y[:, 2, :3].numpy(), b[:, :3].numpy(), atol=1e-10)
    # near-zero branch is smooth
    tiny = 1e-9 * torch.randn(4, 6, generator=g, dtype=torch.float64)
    np.testing.assert_allclose(camera.SE3_log(camera.se3_exp(tiny)).numpy(), tiny.numpy(), atol=1e-15)


def test_oracle_generator_passes_fp64_gradcheck():
    """The chain the GPU gradient tests differentiate: d (RTs, times, deltaT) / d (heads' last layer, time_params)."""
    torch.manual_seed(1)
    m = mm.MoveModel(num_fg=3).double()
    with torch.no_grad():
        for p in m.parameters():
            p.add_(0.05 * torch.randn_like(p))
        m.time_params.copy_(torch.tensor([[0.5, 0.3, 0.47, 0.6, 0.2, 0.5, 0.7, 0.5]]))
    base = {k: v.detach().clone() for k, v in m.state_dict().items()}
    pose = camera.se3_to_SE3(torch.tensor([0.2, -0.1, 0.3, 0.5, -0.4, 0.1], dtype=torch.float64)).double()

    def f(b0, b1, tp):
        sd = dict(base)
        sd["RT_head0.2.bias"], sd["RT_head1.2.bias"], sd["time_params"] = b0, b1, tp
        RTs, times, dT = camera.forward_start_end_mid(sd, pose[:, :3], pose[:, 3:4], 2.0, 5, "second")
        return RTs, times, dT

    args = [base[k].clone().requires_grad_() for k in ("RT_head0.2.bias", "RT_head1.2.bias", "time_params")]
    assert torch.autograd.gradcheck(f, args, eps=1e-6, atol=1e-6, rtol=1e-4)
