"""a12 on the GPU: the fused camera-path kernels (csrc/camera.hip, through the C ABI) against the ORACLE
(oracle/camera.py, pinned to the reference's Python by tests/golden F4/F5 for the torch half): values in fp32/fp64 and
every gradient - all MoveModel parameters, `time_params`, and the input pose - against fp64 autograd of the oracle.
Tolerances: values 3e-6 abs; gradients 1e-4 relative to the largest entry of the tensor (north_star's figure)."""
import numpy as np
import pytest
import torch

from deblur4dgs_amd import move_model as mm
from oracle import camera
from tests.util import record

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _model(seed=0, scale=0.05):
    torch.manual_seed(seed)
    m = mm.MoveModel(num_fg=5)
    with torch.no_grad():
        for p in m.parameters():
            p.add_(scale * torch.randn_like(p))
        m.time_params.copy_(torch.tensor([[0.5, 0.03, 0.47, 1.3, -0.2, 0.5, 0.77, 0.5]]))
    return m.to(DEV)


def _pose(vec):
    w2c = torch.eye(4)
    w2c[:3] = camera.se3_to_SE3(torch.as_tensor(vec, dtype=torch.float32))
    return w2c


@pytest.mark.parametrize("S", [2, 5, 11, 16])
def test_fused_matches_oracle(S):
    m = _model()
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(S)
    for t in (0.0, 1.0, 2.0, 3.0, 4.0, 6.0, 7.0):  # time_params rows: clamp-low, interior, clamp-high, relu-dead, ...
        for stage in ("first", "second"):
            w2c = _pose(0.4 * torch.randn(6, generator=g)).to(DEV)
            info = {"R": w2c[:3, :3], "T": w2c[:3, 3:4], "timestep": t}  # strided views, as scene_model passes
            RTs, times, dT = m.forward_start_end_mid(info, num_cameras=S, mode="uniform", stage=stage)
            oR, ot, od = camera.forward_start_end_mid(sd, w2c[:3, :3].cpu(), w2c[:3, 3:4].cpu(), t, S, stage)
            assert RTs.shape == oR.shape and times.shape == ot.shape and dT.shape == od.shape
            np.testing.assert_allclose(RTs.detach().cpu().numpy(), oR.numpy(), rtol=0, atol=3e-6)
            np.testing.assert_allclose(times.detach().cpu().numpy(), ot.numpy(), rtol=0, atol=1e-6)
            np.testing.assert_array_equal(dT.detach().cpu().numpy(), od.numpy())
            # the module-interface forward() (move_model.py:112-135) reads the heads back from the same kernels
            # (values only: under autograd it refuses rather than hand out graph-less tensors - forward_start_end_mid is
            # the differentiable entry point, as on the reference's render path)
            with pytest.raises(RuntimeError, match="no autograd graph"):
                m(info["R"], info["T"], t, stage=stage)
            with torch.no_grad():
                d0, d1, t0, t1 = m(info["R"], info["T"], t, stage=stage)
            o0, o1, ot0, ot1 = camera.move_model_forward(sd, w2c[:3, :3].cpu(), w2c[:3, 3:4].cpu(), t, stage)
            np.testing.assert_allclose(d0.cpu().numpy(), o0.numpy(), rtol=0, atol=2e-6)
            np.testing.assert_allclose(d1.cpu().numpy(), o1.numpy(), rtol=0, atol=2e-6)
            np.testing.assert_array_equal(t0.cpu().numpy(), ot0.numpy() + 0.0)
            np.testing.assert_array_equal(t1.cpu().numpy(), ot1.numpy())


@pytest.mark.parametrize("scale", [0.0, 0.05, 0.5])
def test_fused_gradients_match_oracle_autograd(scale):
    """Every parameter gradient and the input-pose gradient of d4gs_move_model_bwd / d4gs_pose_encode_bwd against fp64
    autograd of oracle.camera.forward_start_end_mid.  scale 0 = the zero-initialised heads (theta == 0: the norm and
    the guarded branches sit on their special cases)."""
    S = 11
    for t in (2.0, 3.0, 4.0, 0.0):
        m = _model(seed=3, scale=0.05)
        if scale != 0.05:
            with torch.no_grad():
                for head in (m.RT_head0, m.RT_head1):
                    head[-1].weight.mul_(scale / 0.05)
                    head[-1].bias.mul_(scale / 0.05)
        w2c = _pose([0.2, -0.1, 0.3, 0.5, -0.4, 0.1])
        g = torch.Generator().manual_seed(1)
        wR, wt, wd = torch.randn(S, 3, 4, generator=g), torch.randn(1, S, generator=g), 0.7

        # oracle, fp64 autograd
        sd = {k: v.detach().cpu().double().requires_grad_() for k, v in m.state_dict().items()}
        Ro = w2c[:3, :3].double().clone().requires_grad_()
        To = w2c[:3, 3:4].double().clone().requires_grad_()
        oR, ot, od = camera.forward_start_end_mid(sd, Ro, To, t, S, "second")
        ((oR * wR.double()).sum() + (ot * wt.double()).sum() + wd * od.sum()).backward()

        # product
        Rg = w2c[:3, :3].to(DEV).clone().requires_grad_()
        Tg = w2c[:3, 3:4].to(DEV).clone().requires_grad_()
        m.zero_grad(set_to_none=True)
        RTs, times, dT = m.forward_start_end_mid({"R": Rg, "T": Tg, "timestep": t}, num_cameras=S)
        ((RTs * wR.to(DEV)).sum() + (times * wt.to(DEV)).sum() + wd * dT.sum()).backward()
        torch.cuda.synchronize()
        got = {n: (p.grad if p.grad is not None else torch.zeros_like(p)) for n, p in m.named_parameters()}
        got["input.R"], got["input.T"] = Rg.grad, Tg.grad
        want = {n: (sd[n].grad if sd[n].grad is not None else torch.zeros_like(sd[n])) for n in sd}
        want["input.R"], want["input.T"] = Ro.grad, To.grad
        for n in got:
            ref = want[n].numpy()
            r, _ = record(f"a12 grads scale={scale} t={t}", n, got[n].cpu(), want[n])
            tol = 1e-4 * max(float(np.abs(ref).max()), 1e-6)
            np.testing.assert_allclose(got[n].cpu().numpy(), ref, rtol=0, atol=tol, err_msg=f"{n} t={t} scale={scale}")
        assert (float(got["time_params"].abs().sum()) > 0) == (t == 2.0)  # 0.47 is the only un-clamped interior row


def test_parameters_are_version_checked_between_forward_and_backward():
    """saved parameters go through save_for_backward, so an in-place update before backward raises."""
    m = _model()
    w2c = _pose([0.1, 0.2, -0.1, 0.3, 0.1, 0.2]).to(DEV)
    RTs, times, dT = m.forward_start_end_mid({"R": w2c[:3, :3], "T": w2c[:3, 3:4], "timestep": 2.0}, num_cameras=11)
    with torch.no_grad():
        m.RT_main[0].weight.add_(1.0)
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        RTs.sum().backward()


def test_pose_encode_matches_oracle():
    g = torch.Generator().manual_seed(0)
    for _ in range(20):
        w2c = _pose(torch.randn(6, generator=g) * torch.tensor([0.5, 0.5, 0.5, 2.0, 2.0, 2.0]))
        want = camera.posenc(camera.SE3_to_se3(w2c[:3]).unsqueeze(0))
        w2c = w2c.to(DEV)
        got = mm.pose_encode(w2c[:3, :3], w2c[:3, 3:4])
        # rotations kept away from pi (1/(2A) of SO3_to_so3 is ill-conditioned there); |x| up to ~6 times f = 16:
        # sin/cos of ~100 rad amplify the few-ulp differences of acos / the Taylor sums by 16
        np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=0, atol=3e-5)


def test_c_abi_rejects_bad_arguments():
    from deblur4dgs_amd import _lib as L
    import ctypes as C

    z = C.c_void_p(0)
    assert L.lib().d4gs_camera_path_fwd(z, z, 4, z, 8, 1, 0.0, z, z, z, z, z, z) < 0
    assert b"NULL" in L.lib().d4gs_last_error()
    buf = torch.zeros(64, device=DEV)
    p = C.c_void_p(buf.data_ptr())
    assert L.lib().d4gs_camera_path_fwd(p, p, 0, z, 8, 1, 0.0, p, z, p, p, p, z) < 0
    assert L.lib().d4gs_pose_encode(p, 2, p, 1, p, z) < 0
    assert L.lib().d4gs_pose_encode_bwd(p, 3, p, 1, z, p, p, z) < 0
