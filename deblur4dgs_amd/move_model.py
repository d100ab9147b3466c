"""Host-side camera-delta / exposure-time generator.

Mirror of the reference's `MoveModel` (flow3d/models/move_model.py:66-213) and of the SE(3) helpers it uses
(flow3d/models/utils/spline_utils.py:12-54,177-215,371-408).  Same class / method names, same state_dict keys
(`RT_main.*`, `RT_head0.*`, `RT_head1.*`, `time_params`), so a reference checkpoint's `move_model` entry loads.

The MLP (30k parameters) stays `nn.Linear`.  Everything after it - se3.Exp of the two heads, lerp/slerp, SE3.Log,
se3_to_SE3, the exposure-time lerp - and the pose pre-processing + positional embedding before it are latency-only
work (12 differentiable inputs, 12*S outputs) that eager PyTorch spreads over ~700 forward and ~1500 backward
launches per render (12.6 ms on MI355X, 3 renders per training step).  On a GPU they run as ONE HIP launch each
way (`csrc/camera.hip`: dual-number forward that also emits the Jacobian, mat-vec backward; `CameraPathFn` /
`pose_encode` below) and raise if libd4gs.so is missing.  For CPU tensors the reference-style eager chain below
runs, exactly as the reference's own device-agnostic module does.  The pypose ops the reference calls (se3.Exp, SE3.Log, SO3 Inv/@/Log, so3.Exp, bvv; pypose==0.6.8 is not installable here) are
restated in torch below, including the reference's convention quirk: pypose's Log returns [tau, phi] and the
result is fed to `se3_to_SE3`, which reads it as [w (rotation), u (translation)] (move_model.py:146-147).
"""
from __future__ import annotations

import math

import ctypes as C

import torch
import torch.nn as nn
import torch.nn.functional as F


# ------------------------------------------------------------------ spline_utils.py:12-54,177-215
def skew_symmetric(w):
    w0, w1, w2 = w.unbind(dim=-1)
    O = torch.zeros_like(w0)
    return torch.stack([torch.stack([O, -w2, w1], -1), torch.stack([w2, O, -w0], -1), torch.stack([-w1, w0, O], -1)], -2)


def _series(x, nth, step):
    ans, denom = torch.zeros_like(x), 1.0
    for i in range(nth + 1):
        denom *= step(i)
        ans = ans + (-1) ** i * x ** (2 * i) / denom
    return ans


def taylor_A(x, nth=10):
    return _series(x, nth, lambda i: (2 * i) * (2 * i + 1) if i > 0 else 1.0)


def taylor_B(x, nth=10):
    return _series(x, nth, lambda i: (2 * i + 1) * (2 * i + 2))


def taylor_C(x, nth=10):
    return _series(x, nth, lambda i: (2 * i + 2) * (2 * i + 3))


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
    A, B = taylor_A(theta), taylor_B(theta)
    invV = I - 0.5 * wx + (1 - A / (2 * B)) / (theta**2 + eps) * wx @ wx
    return torch.cat([w, (invV @ t)[..., 0]], dim=-1)


def se3_to_SE3(wu):
    w, u = wu.split([3, 3], dim=-1)
    wx = skew_symmetric(w)
    theta = w.norm(dim=-1)[..., None, None]
    I = torch.eye(3, device=w.device, dtype=torch.float32)
    A, B, C = taylor_A(theta), taylor_B(theta), taylor_C(theta)
    R = I + A * wx + B * wx @ wx
    V = I + B * wx + C * wx @ wx
    return torch.cat([R, V @ u[..., None]], dim=-1)


# ------------------------------------------------------------------ pypose 0.6.8 ops, restated
def _guarded(theta2, big, small, eps=1e-12):
    ok = theta2 > eps
    th = torch.sqrt(torch.where(ok, theta2, torch.ones_like(theta2)))
    return torch.where(ok, big(th), small(theta2))


def so3_Exp(phi):
    t2 = (phi * phi).sum(-1, keepdim=True)
    im = _guarded(t2, lambda th: torch.sin(0.5 * th) / th, lambda x: 0.5 - x / 48.0 + x * x / 3840.0)
    re = _guarded(t2, lambda th: torch.cos(0.5 * th), lambda x: 1.0 - x / 8.0 + x * x / 384.0)
    return torch.cat([phi * im, re], -1)


def SO3_Log(q):
    v, w = q[..., :3], q[..., 3:]
    n2 = (v * v).sum(-1, keepdim=True)
    return _guarded(n2, lambda n: 2.0 * torch.atan(n / w) / n, lambda x: 2.0 / w - 2.0 * x / (3.0 * w**3)) * v


def SO3_mul(p, q):
    pv, pw, qv, qw = p[..., :3], p[..., 3:], q[..., :3], q[..., 3:]
    a, b = torch.broadcast_tensors(pv, qv)
    return torch.cat([pw * qv + qw * pv + torch.linalg.cross(a, b, dim=-1), pw * qw - (pv * qv).sum(-1, keepdim=True)], -1)


def SO3_Inv(q):
    return torch.cat([-q[..., :3], q[..., 3:]], -1)


def _Jl(phi):
    t2 = (phi * phi).sum(-1, keepdim=True)[..., None]
    K = skew_symmetric(phi)
    c1 = _guarded(t2, lambda th: (1 - torch.cos(th)) / th**2, lambda x: 0.5 - x / 24.0)
    c2 = _guarded(t2, lambda th: (th - torch.sin(th)) / th**3, lambda x: 1.0 / 6.0 - x / 120.0)
    return torch.eye(3, dtype=phi.dtype, device=phi.device) + c1 * K + c2 * (K @ K)


def _Jl_inv(phi):
    t2 = (phi * phi).sum(-1, keepdim=True)[..., None]
    K = skew_symmetric(phi)
    c2 = _guarded(t2, lambda th: (1 - th * torch.cos(0.5 * th) / (2 * torch.sin(0.5 * th))) / th**2,
                  lambda x: 1.0 / 12.0 + x / 720.0)
    return torch.eye(3, dtype=phi.dtype, device=phi.device) - 0.5 * K + c2 * (K @ K)


def se3_Exp(xi):
    tau, phi = xi[..., :3], xi[..., 3:]
    return torch.cat([(_Jl(phi) @ tau[..., None])[..., 0], so3_Exp(phi)], -1)


def SE3_Log(X):
    phi = SO3_Log(X[..., 3:])
    return torch.cat([(_Jl_inv(phi) @ X[..., :3, None])[..., 0], phi], -1)


def linear_interpolation(start, end, u):
    """spline_utils.py:371-408: lerp translations, slerp rotations.  start/end [...,7], u [I] -> [...,I,7]."""
    ts, qs, te, qe = start[..., :3], start[..., 3:], end[..., :3], end[..., 3:]
    u = u.expand(*start.shape[:-1], -1)
    t = (1 - u)[..., None] * ts[..., None, :] + u[..., None] * te[..., None, :]
    r = SO3_Log(SO3_mul(SO3_Inv(qs), qe))
    q = SO3_mul(qs[..., None, :], so3_Exp(u[..., None] * r[..., None, :]))
    return torch.cat([t, q], -1)


# ------------------------------------------------------------------ move_model.py:12-63
def _posenc(x, num_freqs=5):
    outs = [x]
    for f in 2.0 ** torch.linspace(0.0, num_freqs - 1, steps=num_freqs):
        outs += [torch.sin(x * f), torch.cos(x * f)]
    return torch.cat(outs, -1)


P_input_ch = 6 * (1 + 2 * 5)  # 66


# ------------------------------------------------------------------ fused HIP path (csrc/camera.hip)
def _stream(t):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def pose_encode(R, T):
    """preprocessPose + positional embedding of one pose in one launch: R [3,3] (any row stride), T [3,1] -> [1,66]."""
    from . import _lib as L

    assert R.is_cuda and R.dtype == torch.float32 and T.dtype == torch.float32 and R.shape == (3, 3)
    if R.stride(1) != 1:
        R = R.contiguous()
    enc = torch.empty(1, P_input_ch, device=R.device, dtype=torch.float32)
    T = T.reshape(3, -1)[:, 0]
    L.check(L.lib().d4gs_pose_encode(_p(R), R.stride(0), _p(T), T.stride(0), _p(enc), _stream(R)), "pose_encode")
    return enc


MLP_ACTS = P_input_ch + 7 * 64


class MoveModelFn(torch.autograd.Function):
    """The whole module for one pose in one C call each way (d4gs_move_model_fwd / _bwd): pose encoding, the 9-layer
    MLP and the camera path.  params = (time_params, w0, b0, .., w8, b8) in the layer order of include/d4gs.h."""

    @staticmethod
    def forward(ctx, R, T, S, index, t, stage_first, time_params, *wb):
        from . import _lib as L

        dev = R.device
        if R.stride(1) != 1:
            R = R.contiguous()
        T = T.reshape(3, -1)[:, 0]
        ws = [x.detach().contiguous() for x in wb[0::2]]
        bs = [x.detach().contiguous() for x in wb[1::2]]
        tp = time_params.detach().contiguous()
        buf = torch.empty(P_input_ch + MLP_ACTS + 12 + S * 12 + S * 144 + 2 * S + 2, device=dev, dtype=torch.float32)
        enc, acts, delta, RTs, jac, times, dtimes, dT = buf.split([P_input_ch, MLP_ACTS, 12, S * 12, S * 144, S, S, 2])
        pp = L.MoveModelParams()
        for l in range(9):
            pp.w[l], pp.b[l] = ws[l].data_ptr(), bs[l].data_ptr()
        pp.time_params, pp.n_time_params = tp.data_ptr(), tp.numel()
        out = L.fill(L.MoveModelOut(), enc=enc, acts=acts, delta=delta, RTs=RTs, jac=jac, times=times, dtimes=dtimes,
                     deltaT=dT)
        L.check(L.lib().d4gs_move_model_fwd(_p(R), R.stride(0), _p(T), T.stride(0), C.byref(pp), S, index, float(t),
                                            int(stage_first), C.byref(out), _stream(R)), "move_model_fwd")
        ctx.keep = (buf, ws, bs, tp)  # non-differentiable buffers (plain attributes: no version checks needed)
        ctx.meta = (S, index, time_params.shape, [x.shape for x in wb])
        return RTs.view(S, 3, 4), times.view(1, S), dT[:1].view(1, 1)

    @staticmethod
    def backward(ctx, v_RTs, v_times, v_dT):
        from . import _lib as L

        buf, ws, bs, tp = ctx.keep
        S, index, tp_shape, shapes = ctx.meta
        enc, acts, delta, RTs, jac, times, dtimes, dT = buf.split([P_input_ch, MLP_ACTS, 12, S * 12, S * 144, S, S, 2])
        cont = lambda v: None if v is None else v.contiguous().float()
        v_RTs, v_times, v_dT = cont(v_RTs), cont(v_times), cont(v_dT)
        sizes = [tp.numel(), 12] + [x.numel() for pair in zip(ws, bs) for x in pair]
        gbuf = torch.empty(sum(sizes), device=buf.device, dtype=torch.float32)
        parts = gbuf.split(sizes)
        pp = L.MoveModelParams()
        gg = L.MoveModelGrads()
        for l in range(9):
            pp.w[l], pp.b[l] = ws[l].data_ptr(), bs[l].data_ptr()
            gg.v_w[l], gg.v_b[l] = parts[2 + 2 * l].data_ptr(), parts[3 + 2 * l].data_ptr()
        pp.time_params, pp.n_time_params = tp.data_ptr(), tp.numel()
        gg.v_time_params, gg.v_delta = parts[0].data_ptr(), parts[1].data_ptr()
        out = L.fill(L.MoveModelOut(), enc=enc, acts=acts, delta=delta, RTs=RTs, jac=jac, times=times, dtimes=dtimes,
                     deltaT=dT)
        L.check(L.lib().d4gs_move_model_bwd(C.byref(pp), C.byref(out), _p(v_RTs), _p(v_times), _p(v_dT), S, index,
                                            C.byref(gg), _stream(buf)), "move_model_bwd")
        grads = [parts[2 + k].view(shapes[k]) for k in range(18)]
        return (None, None, None, None, None, None, parts[0].view(tp_shape), *grads)


class CameraPathFn(torch.autograd.Function):
    """(delta0 [1,6], delta1 [1,6], time_params [1,P]) -> RTs [S,3,4], times [1,S], deltaT [1,1]."""

    @staticmethod
    def forward(ctx, delta0, delta1, time_params, S, index, t, moving):
        from . import _lib as L

        dev = delta0.device
        d0, d1 = delta0.detach().contiguous().float(), delta1.detach().contiguous().float()
        tp = time_params.detach().contiguous().float()
        buf = torch.empty(S * 12 + S * 144 + 2 * S + 2, device=dev, dtype=torch.float32)
        RTs, jac, times, dtimes, dT = buf.split([S * 12, S * 144, S, S, 2])
        L.check(L.lib().d4gs_camera_path_fwd(_p(d0), _p(d1), S, _p(tp if moving else None), tp.numel(), index, float(t),
                                             _p(RTs), _p(jac), _p(times), _p(dtimes), _p(dT), _stream(d0)),
                "camera_path_fwd")
        ctx.save_for_backward(jac, dtimes, dT)
        ctx.meta = (S, index, tp.numel(), time_params.shape)
        return RTs.view(S, 3, 4), times.view(1, S), dT[:1].view(1, 1)

    @staticmethod
    def backward(ctx, v_RTs, v_times, v_dT):
        from . import _lib as L

        jac, dtimes, dT = ctx.saved_tensors
        S, index, P, tp_shape = ctx.meta
        cont = lambda v: None if v is None else v.contiguous().float()
        v_RTs, v_times, v_dT = cont(v_RTs), cont(v_times), cont(v_dT)
        out = torch.empty(12 + P, device=jac.device, dtype=torch.float32)
        L.check(L.lib().d4gs_camera_path_bwd(_p(jac), _p(dtimes), _p(dT), _p(v_RTs), _p(v_times), _p(v_dT), S, index, P,
                                             _p(out[:6]), _p(out[6:12]), _p(out[12:]), _stream(jac)),
                "camera_path_bwd")
        return out[:6].view(1, 6), out[6:12].view(1, 6), out[12:].view(tp_shape), None, None, None, None


class MoveModel(nn.Module):
    """move_model.py:66-213 (camera_mode='linear', the only mode the reference instantiates: scene_model.py:36)."""

    def __init__(self, num_fg, camera_mode="linear"):
        super().__init__()
        assert camera_mode == "linear", "the reference only uses camera_mode='linear' (scene_model.py:36)"
        self.slope = 0.01
        W = 32
        self.camera_mode = camera_mode
        lr = lambda: nn.LeakyReLU(self.slope)
        self.RT_main = nn.Sequential(nn.Linear(P_input_ch, W * 2), lr(), nn.Linear(W * 2, W * 2), lr(),
                                     nn.Linear(W * 2, W * 2), lr(), nn.Linear(W * 2, W * 2), lr(),
                                     nn.Linear(W * 2, W * 2))
        self.RT_head0 = nn.Sequential(nn.Linear(W * 2, W * 2), lr(), nn.Linear(2 * W, 6))
        self.RT_head1 = nn.Sequential(nn.Linear(W * 2, W * 2), lr(), nn.Linear(2 * W, 6))
        self.time_params = nn.Parameter(torch.full((1, 8), 0.5))
        self.relu = nn.ReLU()
        self.zero_initialize()

    def zero_initialize(self):
        for head in (self.RT_head0, self.RT_head1):
            nn.init.constant_(head[-1].weight, 0.0)
            nn.init.constant_(head[-1].bias, 0.0)

    def preprocessPose(self, R, T):
        return SE3_to_se3(torch.cat([R, T], dim=-1))

    def postprocessPose(self, RT):
        return se3_to_SE3(RT)

    def _layer_params(self):
        """(w, b) of the 9 linear layers in the order of include/d4gs.h."""
        lin = [self.RT_main[0], self.RT_main[2], self.RT_main[4], self.RT_main[6], self.RT_main[8], self.RT_head0[0],
               self.RT_head0[2], self.RT_head1[0], self.RT_head1[2]]
        return [p for l in lin for p in (l.weight, l.bias)]

    def _fused(self, R, T):
        return R.is_cuda and not (R.requires_grad or T.requires_grad) and R.dtype == T.dtype == torch.float32

    def _heads(self, R, T):
        enc = pose_encode(R, T) if self._fused(R, T) else _posenc(self.preprocessPose(R, T).unsqueeze(0))
        x = self.RT_main(enc)
        return self.RT_head0(x), self.RT_head1(x)

    def forward(self, R, T, time, stage="second"):
        detaRT0, detaRT1 = self._heads(R, T)
        if stage == "first":
            deltaT0 = torch.zeros(1, device=detaRT0.device)
            deltaT1 = torch.zeros(1, device=detaRT0.device)
        else:
            index = int(time)
            if index <= 0 or index >= self.time_params.shape[-1] - 1:
                deltaT0 = torch.zeros_like(self.time_params[:, 0])
                deltaT1 = torch.zeros_like(self.time_params[:, 0])
            else:
                deltaT = self.relu(self.time_params[:, index]).clamp(0.1, 0.9)
                deltaT0, deltaT1 = deltaT * -1.0, deltaT * 1.0
        return detaRT0, detaRT1, deltaT0, deltaT1

    def _interpolate(self, RT_start, RT_end, num_cameras, mode="uniform"):
        assert mode == "uniform"
        u = torch.linspace(start=0, end=1, steps=num_cameras, device=RT_start.device)
        return linear_interpolation(RT_start, RT_end, u)

    def forward_start_end_mid(self, info, num_cameras=10, mode="uniform", stage="second"):
        R, T, time = info["R"], info["T"], info["timestep"]
        if self._fused(R, T) and num_cameras > 1:
            assert mode == "uniform"  # the only mode the reference calls (scene_model.py:254,272)
            return MoveModelFn.apply(R, T, num_cameras, int(time), float(time), stage == "first", self.time_params,
                                     *self._layer_params())
        RT_start, RT_end, time_start, time_end = self.forward(R, T, time, stage=stage)
        RTs = self._interpolate(se3_Exp(RT_start), se3_Exp(RT_end), num_cameras=num_cameras, mode=mode)  # [1,S,7]
        RTs = self.postprocessPose(SE3_Log(RTs)).squeeze(0)  # [S,3,4]
        num_fg = time_start.shape[0]
        time_start = time_start.unsqueeze(-1).repeat(1, num_cameras)
        time_end = time_end.unsqueeze(-1).repeat(1, num_cameras)
        weights = (torch.arange(num_cameras) / (num_cameras - 1)).to(RTs.device)
        weights = weights.unsqueeze(0).repeat(num_fg, 1)
        times = (time_start + time) * (1.0 - weights) + (time_end + time) * weights
        times = times.reshape(num_fg, num_cameras)
        if mode == "mid":
            times = times[:, (num_cameras // 2):(num_cameras // 2 + 1)]
        elif mode == "start":
            times = times[:, 0:1]
        elif mode == "end":
            times = times[:, num_cameras - 1:]
        deltaT = torch.abs(time_end[:, num_cameras - 1:])
        return RTs, times, deltaT
