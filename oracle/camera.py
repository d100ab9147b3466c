"""Oracle: host-side camera-delta / exposure-time generator (a12).

Test infrastructure (see oracle/__init__.py).

* Pure-torch half -- PINNED by tests/golden (reference Python imported in the build container):
  `taylor_A/B/C`, `skew_symmetric`, `SO3_to_so3`, `SE3_to_se3`, `se3_to_SE3`
  (flow3d/models/utils/spline_utils.py:12-54,177-215), the positional embedding and MLP of
  `MoveModel.forward` (flow3d/models/move_model.py:12-63,66-135).
* pypose half -- **parity unpinned** [RECALLED pypose==0.6.8, requirements.txt:354]:
  `se3.Exp`, `SE3.Log`, `SO3.Inv/@/Log`, `so3.Exp`, `bvv` as used by `linear_interpolation`
  (spline_utils.py:371-408) and `forward_start_end_mid` (move_model.py:138-166).
  Conventions: se3 = [tau(3), phi(3)], SE3 = [t(3), q_xyzw(4)].
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- spline_utils.py:12-54
def skew_symmetric(w):
    w0, w1, w2 = w.unbind(dim=-1)
    O = torch.zeros_like(w0)
    return torch.stack(
        [
            torch.stack([O, -w2, w1], dim=-1),
            torch.stack([w2, O, -w0], dim=-1),
            torch.stack([-w1, w0, O], dim=-1),
        ],
        dim=-2,
    )


def _taylor(x, nth, first_denom_step, start):
    ans = torch.zeros_like(x)
    denom = 1.0
    for i in range(nth + 1):
        denom *= first_denom_step(i)
        ans = ans + (-1) ** i * x ** (2 * i) / denom
    return ans


def taylor_A(x, nth=10):  # sin(x)/x
    return _taylor(x, nth, lambda i: (2 * i) * (2 * i + 1) if i > 0 else 1.0, 0)


def taylor_B(x, nth=10):  # (1-cos x)/x^2
    return _taylor(x, nth, lambda i: (2 * i + 1) * (2 * i + 2), 0)


def taylor_C(x, nth=10):  # (x-sin x)/x^3
    return _taylor(x, nth, lambda i: (2 * i + 2) * (2 * i + 3), 0)


# --------------------------------------------------------------------------- spline_utils.py:177-215
def SO3_to_so3(R, eps=1e-7):
    trace = R[..., 0, 0] + R[..., 1, 1] + R[..., 2, 2]
    theta = ((trace - 1) / 2).clamp(-1 + eps, 1 - eps).acos()[..., None, None] % math.pi
    lnR = 1 / (2 * taylor_A(theta) + 1e-8) * (R - R.transpose(-2, -1))
    return torch.stack([lnR[..., 2, 1], lnR[..., 0, 2], lnR[..., 1, 0]], dim=-1)


def SE3_to_se3(Rt, eps=1e-8):
    R, t = Rt.split([3, 1], dim=-1)
    w = SO3_to_so3(R)
    wx = skew_symmetric(w)
    theta = w.norm(dim=-1)[..., None, None]
    I = torch.eye(3, device=w.device, dtype=torch.float32)
    A = taylor_A(theta)
    B = taylor_B(theta)
    invV = I - 0.5 * wx + (1 - A / (2 * B)) / (theta**2 + eps) * wx @ wx
    u = (invV @ t)[..., 0]
    return torch.cat([w, u], dim=-1)


def se3_to_SE3(wu):
    w, u = wu.split([3, 3], dim=-1)
    wx = skew_symmetric(w)
    theta = w.norm(dim=-1)[..., None, None]
    I = torch.eye(3, device=w.device, dtype=torch.float32)
    A = taylor_A(theta)
    B = taylor_B(theta)
    C = taylor_C(theta)
    R = I + A * wx + B * wx @ wx
    V = I + B * wx + C * wx @ wx
    return torch.cat([R, (V @ u[..., None])], dim=-1)


# --------------------------------------------------------------------------- move_model.py:12-63
def posenc(x, num_freqs=5):
    """include_input, log-sampled freqs 2^0..2^(n-1), [sin, cos] per freq -> 6*(1+2n) = 66-d."""
    outs = [x]
    for f in 2.0 ** torch.linspace(0.0, num_freqs - 1, steps=num_freqs):
        outs.append(torch.sin(x * f))
        outs.append(torch.cos(x * f))
    return torch.cat(outs, -1)


def _mlp(x, sd, prefix, n_layers, slope=0.01):
    for li in range(n_layers):
        idx = 2 * li
        x = F.linear(x, sd[f"{prefix}.{idx}.weight"], sd[f"{prefix}.{idx}.bias"])
        if li < n_layers - 1:
            x = F.leaky_relu(x, slope)
    return x


def move_model_forward(sd: dict, R, T, time, stage="second"):
    """MoveModel.forward (move_model.py:112-135).  sd = MoveModel.state_dict()."""
    RT = SE3_to_se3(torch.cat([R, T], dim=-1))[None]
    x = _mlp(posenc(RT), sd, "RT_main", 5)
    d0 = _mlp(x, sd, "RT_head0", 2)
    d1 = _mlp(x, sd, "RT_head1", 2)
    tp = sd["time_params"]
    if stage == "first":
        dT0 = torch.zeros(1, device=RT.device)
        dT1 = torch.zeros(1, device=RT.device)
    else:
        index = int(time)
        if index <= 0 or index >= tp.shape[-1] - 1:
            dT0 = torch.zeros_like(tp[:, 0])
            dT1 = torch.zeros_like(tp[:, 0])
        else:
            dT = F.relu(tp[:, index]).clamp(0.1, 0.9)
            dT0, dT1 = dT * -1.0, dT * 1.0
    return d0, d1, dT0, dT1


# --------------------------------------------------------------------------- pypose 0.6.8 [RECALLED]
def _safe(theta2, big, small_fn, eps=1e-12):
    """where(theta2 > eps, big(theta), small(theta2)) without NaN gradients."""
    ok = theta2 > eps
    th = torch.sqrt(torch.where(ok, theta2, torch.ones_like(theta2)))
    return torch.where(ok, big(th), small_fn(theta2))


def so3_exp(phi):
    """-> unit quaternion xyzw."""
    t2 = (phi * phi).sum(-1, keepdim=True)
    imag = _safe(t2, lambda th: torch.sin(0.5 * th) / th, lambda x: 0.5 - x / 48.0 + x * x / 3840.0)
    real = _safe(t2, lambda th: torch.cos(0.5 * th), lambda x: 1.0 - x / 8.0 + x * x / 384.0)
    return torch.cat([phi * imag, real], -1)


def so3_log(q):
    v, w = q[..., :3], q[..., 3:]
    n2 = (v * v).sum(-1, keepdim=True)
    fac = _safe(
        n2,
        lambda n: 2.0 * torch.atan(n / w) / n,
        lambda x: 2.0 / w - 2.0 * x / (3.0 * w**3),
    )
    return fac * v


def quat_mul(p, q):  # xyzw Hamilton
    pv, pw = p[..., :3], p[..., 3:]
    qv, qw = q[..., :3], q[..., 3:]
    pvb, qvb = torch.broadcast_tensors(pv, qv)
    return torch.cat(
        [pw * qv + qw * pv + torch.linalg.cross(pvb, qvb, dim=-1), pw * qw - (pv * qv).sum(-1, keepdim=True)],
        -1,
    )


def quat_inv(q):
    return torch.cat([-q[..., :3], q[..., 3:]], -1)


def _Jl(phi):
    t2 = (phi * phi).sum(-1, keepdim=True)[..., None]
    K = skew_symmetric(phi)
    c1 = _safe(t2, lambda th: (1 - torch.cos(th)) / th**2, lambda x: 0.5 - x / 24.0)
    c2 = _safe(t2, lambda th: (th - torch.sin(th)) / th**3, lambda x: 1.0 / 6.0 - x / 120.0)
    I = torch.eye(3, dtype=phi.dtype, device=phi.device)
    return I + c1 * K + c2 * (K @ K)


def _Jl_inv(phi):
    t2 = (phi * phi).sum(-1, keepdim=True)[..., None]
    K = skew_symmetric(phi)
    c2 = _safe(
        t2,
        lambda th: (1 - th * torch.cos(0.5 * th) / (2 * torch.sin(0.5 * th))) / th**2,
        lambda x: 1.0 / 12.0 + x / 720.0,
    )
    I = torch.eye(3, dtype=phi.dtype, device=phi.device)
    return I - 0.5 * K + c2 * (K @ K)


def se3_exp(xi):
    """xi = [tau, phi] -> SE3 [t, q_xyzw]."""
    tau, phi = xi[..., :3], xi[..., 3:]
    t = (_Jl(phi) @ tau[..., None])[..., 0]
    return torch.cat([t, so3_exp(phi)], -1)


def SE3_log(X):
    t, q = X[..., :3], X[..., 3:]
    phi = so3_log(q)
    tau = (_Jl_inv(phi) @ t[..., None])[..., 0]
    return torch.cat([tau, phi], -1)


def linear_interpolation(start, end, u):
    """spline_utils.py:371-408.  start/end SE3 [...,7]; u [I] -> SE3 [..., I, 7]."""
    ts, qs = start[..., :3], start[..., 3:]
    te, qe = end[..., :3], end[..., 3:]
    u = u.expand(*start.shape[:-1], -1)
    t = (1 - u)[..., None] * ts[..., None, :] + u[..., None] * te[..., None, :]
    r = so3_log(quat_mul(quat_inv(qs), qe))
    q = quat_mul(qs[..., None, :], so3_exp(u[..., None] * r[..., None, :]))
    return torch.cat([t, q], -1)


def forward_start_end_mid(sd: dict, R, T, time, num_cameras=11, stage="second"):
    """move_model.py:138-166 with mode='uniform', camera_mode='linear'.
    -> RTs [S,3,4], times [1,S], deltaT [1,1]."""
    d0, d1, t0, t1 = move_model_forward(sd, R, T, time, stage)
    P0, P1 = se3_exp(d0), se3_exp(d1)  # [1,7]
    u = torch.linspace(0, 1, num_cameras, device=R.device)
    X = linear_interpolation(P0, P1, u)  # [1,S,7]
    RTs = se3_to_SE3(SE3_log(X))[0]  # pypose [tau,phi] read as [w,u] (reference quirk, reproduced)
    n = t0.shape[0]
    ts = t0[:, None].repeat(1, num_cameras)
    te = t1[:, None].repeat(1, num_cameras)
    wts = (torch.arange(num_cameras) / (num_cameras - 1)).to(RTs.device)[None].repeat(n, 1)
    times = (ts + time) * (1.0 - wts) + (te + time) * wts
    deltaT = torch.abs(te[:, num_cameras - 1 :])
    return RTs, times.reshape(n, num_cameras), deltaT
