"""a12 on the GPU: the fused camera-path kernels (csrc/camera.hip, through the C ABI) against the eager chain that
tests/test_move_model.py pins to the oracle / the reference's golden vectors.  Tolerances: f32 forward 2e-6 abs
(same formulas, different sin/cos/atan evaluation order), Jacobian-vector products 2e-5 relative to the largest
entry."""
import numpy as np
import pytest
import torch

from deblur4dgs_amd import move_model as mm
from oracle import camera

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


def _eager(m, info, S, mode, stage):
    """the reference-style eager chain on the same device (bypasses the fused path)"""
    R, T = info["R"].clone().requires_grad_(), info["T"]  # requires_grad on R disables _fused
    return m.forward_start_end_mid({"R": R, "T": T, "timestep": info["timestep"]}, num_cameras=S, mode=mode, stage=stage)


@pytest.mark.parametrize("S", [2, 5, 11, 16])
def test_fused_matches_eager_and_oracle(S):
    m = _model()
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(S)
    for t in (0.0, 1.0, 2.0, 3.0, 4.0, 6.0, 7.0):  # time_params rows: clamp-low, interior, clamp-high, relu-dead, ...
        for stage in ("first", "second"):
            for mode in ("uniform",):
                w2c = torch.eye(4)
                w2c[:3] = mm.se3_to_SE3(0.4 * torch.randn(6, generator=g))
                w2c = w2c.to(DEV)
                info = {"R": w2c[:3, :3], "T": w2c[:3, 3:4], "timestep": t}  # strided views, as scene_model passes
                RTs, times, dT = m.forward_start_end_mid(info, num_cameras=S, mode=mode, stage=stage)
                eR, et, ed = _eager(m, info, S, mode, stage)
                assert RTs.shape == eR.shape and times.shape == et.shape and dT.shape == ed.shape
                np.testing.assert_allclose(RTs.detach().cpu().numpy(), eR.detach().cpu().numpy(), rtol=0, atol=2e-6)
                np.testing.assert_allclose(times.detach().cpu().numpy(), et.detach().cpu().numpy(), rtol=0, atol=1e-6)
                np.testing.assert_array_equal(dT.detach().cpu().numpy(), ed.detach().cpu().numpy())
                if mode == "uniform":
                    oR, ot, od = camera.forward_start_end_mid(sd, w2c[:3, :3].cpu(), w2c[:3, 3:4].cpu(), t, S, stage)
                    np.testing.assert_allclose(RTs.detach().cpu().numpy(), oR.numpy(), rtol=0, atol=3e-6)
                    np.testing.assert_allclose(times.detach().cpu().numpy(), ot.numpy(), rtol=0, atol=1e-6)


@pytest.mark.parametrize("scale", [0.0, 0.05, 0.5])
def test_fused_gradients_match_eager_autograd(scale):
    """scale 0 = the zero-initialised heads (theta == 0: the norm / guarded branches sit on their special cases)."""
    S = 11
    for t in (2.0, 3.0, 4.0, 0.0):
        m = _model(seed=3, scale=0.05)
        if scale != 0.05:
            with torch.no_grad():
                for head in (m.RT_head0, m.RT_head1):
                    head[-1].weight.mul_(scale / 0.05)
                    head[-1].bias.mul_(scale / 0.05)
        w2c = torch.eye(4)
        w2c[:3] = mm.se3_to_SE3(torch.tensor([0.2, -0.1, 0.3, 0.5, -0.4, 0.1]))
        w2c = w2c.to(DEV)
        info = {"R": w2c[:3, :3], "T": w2c[:3, 3:4], "timestep": t}
        g = torch.Generator().manual_seed(1)
        wR, wt, wd = torch.randn(S, 3, 4, generator=g).to(DEV), torch.randn(1, S, generator=g).to(DEV), 0.7

        def grads(fn):
            m.zero_grad(set_to_none=True)
            RTs, times, dT = fn()
            ((RTs * wR).sum() + (times * wt).sum() + wd * dT.sum()).backward()
            return {n: (p.grad.clone() if p.grad is not None else torch.zeros_like(p)) for n, p in m.named_parameters()}

        gf = grads(lambda: m.forward_start_end_mid(info, num_cameras=S))
        ge = grads(lambda: _eager(m, info, S, "uniform", "second"))
        for n in gf:
            ref = ge[n].cpu().numpy()
            tol = 2e-5 * max(1.0, float(np.abs(ref).max()))
            np.testing.assert_allclose(gf[n].cpu().numpy(), ref, rtol=0, atol=tol, err_msg=f"{n} t={t} scale={scale}")
        assert (float(gf["time_params"].abs().sum()) > 0) == (t == 2.0)  # 0.47 is the only un-clamped interior row


def test_pose_encode_matches_eager():
    g = torch.Generator().manual_seed(0)
    for _ in range(20):
        w2c = torch.eye(4)
        w2c[:3] = mm.se3_to_SE3(torch.randn(6, generator=g) * torch.tensor([0.5, 0.5, 0.5, 2.0, 2.0, 2.0]))
        want = mm._posenc(mm.SE3_to_se3(w2c[:3]).unsqueeze(0))
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
