"""Host-side camera-delta / exposure-time generator.

The module interface of the reference's `MoveModel` (flow3d/models/move_model.py:66-213): same class / method
names and the same state_dict keys (`RT_main.*`, `RT_head0.*`, `RT_head1.*`, `time_params`), so a reference
checkpoint's `move_model` entry loads.  The parameters stay `nn.Linear` modules (optimizers and checkpoints are
unchanged); ALL arithmetic runs in libd4gs.so (`csrc/camera.hip`):

  d4gs_move_model_fwd   SE3_to_se3 + positional embedding of the pose (spline_utils.py:177-195, move_model.py:12-63),
                        the 9-layer MLP, pypose se3.Exp of the two heads, lerp / slerp (spline_utils.py:371-408),
                        SE3.Log read as [w,u] by se3_to_SE3 (the reference's convention quirk, move_model.py:146-147)
                        and the exposure-time lerp (move_model.py:118-135,151-158) - dual-number forward that also
                        emits the Jacobian;
  d4gs_move_model_bwd   mat-vec through that Jacobian + the MLP backward;
  d4gs_pose_encode_bwd  gradient w.r.t. the input pose (test-time pose refinement, flow3d/validator.py:442-448).

There is no eager / CPU implementation here: CPU tensors raise.  The torch restatement of the same chain lives in
`oracle/camera.py` (test infrastructure) and is what the parity tests differentiate.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

P_input_ch = 6 * (1 + 2 * 5)  # 66 (move_model.py:12-63: include_input + 5 log-sampled frequencies)


def _need_gpu(t: torch.Tensor):
    if not t.is_cuda:
        raise RuntimeError("deblur4dgs_amd.MoveModel runs on an MI355X (ROCm) device only; got a CPU tensor "
                           "(no CPU fallback)")


# ------------------------------------------------------------------ fused HIP path (csrc/camera.hip)
def _stream(t):
    from . import _lib as L

    return C.c_void_p(L.raw_stream(t.device.index))


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
    MLP and the camera path.  params = (time_params, w0, b0, .., w8, b8) in the layer order of include/d4gs.h.
    Gradients: every parameter, and the input pose (R, T) when it requires grad (d4gs_pose_encode_bwd)."""

    @staticmethod
    def forward(ctx, R, T, S, index, t, stage_first, time_params, *wb):
        from . import _lib as L

        dev = R.device
        t_shape = T.shape
        R = R.detach()
        if R.stride(1) != 1:
            R = R.contiguous()
        T = T.detach().reshape(3, -1)[:, 0]
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
        # The parameters go through save_for_backward so that an in-place update between forward and backward
        # (optimizer.step, load_state_dict) raises instead of silently differentiating the wrong weights.
        # ... and so do the detached aliases of the caller's pose: the pose-gradient backward re-reads R / T, and an in-place
        # pose update between forward and backward (test-time pose refinement, flow3d/validator.py:442-448) must raise
        # rather than be differentiated at the wrong point
        ctx.save_for_backward(tp, *ws, *bs, R, T)
        ctx.keep = (buf,)  # the buffer this function created itself
        ctx.meta = (S, index, time_params.shape, [x.shape for x in wb], t_shape)
        return RTs.view(S, 3, 4), times.view(1, S), dT[:1].view(1, 1)

    @staticmethod
    def backward(ctx, v_RTs, v_times, v_dT):
        from . import _lib as L

        tp, *wsbs = ctx.saved_tensors
        ws, bs, (R, T) = wsbs[:9], wsbs[9:18], wsbs[18:]
        (buf,) = ctx.keep
        S, index, tp_shape, shapes, t_shape = ctx.meta
        enc, acts, delta, RTs, jac, times, dtimes, dT = buf.split([P_input_ch, MLP_ACTS, 12, S * 12, S * 144, S, S, 2])
        cont = lambda v: None if v is None else v.contiguous().float()
        v_RTs, v_times, v_dT = cont(v_RTs), cont(v_times), cont(v_dT)
        pose_grad = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        sizes = [tp.numel(), 12] + [x.numel() for pair in zip(ws, bs) for x in pair] + [P_input_ch, 9, 3]
        gbuf = torch.empty(sum(sizes), device=buf.device, dtype=torch.float32)
        parts = gbuf.split(sizes)
        pp = L.MoveModelParams()
        gg = L.MoveModelGrads()
        for l in range(9):
            pp.w[l], pp.b[l] = ws[l].data_ptr(), bs[l].data_ptr()
            gg.v_w[l], gg.v_b[l] = parts[2 + 2 * l].data_ptr(), parts[3 + 2 * l].data_ptr()
        pp.time_params, pp.n_time_params = tp.data_ptr(), tp.numel()
        gg.v_time_params, gg.v_delta = parts[0].data_ptr(), parts[1].data_ptr()
        gg.v_enc = parts[20].data_ptr() if pose_grad else None
        out = L.fill(L.MoveModelOut(), enc=enc, acts=acts, delta=delta, RTs=RTs, jac=jac, times=times, dtimes=dtimes,
                     deltaT=dT)
        L.check(L.lib().d4gs_move_model_bwd(C.byref(pp), C.byref(out), _p(v_RTs), _p(v_times), _p(v_dT), S, index,
                                            C.byref(gg), _stream(buf)), "move_model_bwd")
        v_R = v_T = None
        if pose_grad:
            L.check(L.lib().d4gs_pose_encode_bwd(_p(R), R.stride(0), _p(T), T.stride(0), _p(parts[20]), _p(parts[21]),
                                                 _p(parts[22]), _stream(buf)), "pose_encode_bwd")
            v_R, v_T = parts[21].view(3, 3), parts[22].view(t_shape)
        grads = [parts[2 + k].view(shapes[k]) for k in range(18)]
        return (v_R, v_T, None, None, None, None, parts[0].view(tp_shape), *grads)


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
        self.zero_initialize()

    def zero_initialize(self):
        for head in (self.RT_head0, self.RT_head1):
            nn.init.constant_(head[-1].weight, 0.0)
            nn.init.constant_(head[-1].bias, 0.0)

    def _layer_params(self):
        """(w, b) of the 9 linear layers in the order of include/d4gs.h."""
        lin = [self.RT_main[0], self.RT_main[2], self.RT_main[4], self.RT_main[6], self.RT_main[8], self.RT_head0[0],
               self.RT_head0[2], self.RT_head1[0], self.RT_head1[2]]
        return [p for l in lin for p in (l.weight, l.bias)]

    def forward_start_end_mid(self, info, num_cameras=10, mode="uniform", stage="second"):
        """-> RTs [S,3,4] camera deltas, times [1,S] exposure times, deltaT [1,1] (move_model.py:138-166)."""
        R, T, time = info["R"], info["T"], info["timestep"]
        _need_gpu(R)
        assert mode == "uniform", "the only mode the reference calls (scene_model.py:254,272)"
        assert num_cameras > 1, "S == 1 divides by S - 1 = 0 upstream (move_model.py:154); render() always passes 11"
        return MoveModelFn.apply(R.float(), T.float(), num_cameras, int(time), float(time), stage == "first",
                                 self.time_params, *self._layer_params())

    def forward(self, R, T, time, stage="second"):
        """The two head outputs and the signed exposure half-widths (move_model.py:112-135).  The render path does not
        call this (forward_start_end_mid is one fused launch); it exists for the reference's module interface and reads
        the head outputs back from the same fused forward."""
        _need_gpu(R)
        from . import _lib as L

        if torch.is_grad_enabled() and (R.requires_grad or T.requires_grad or any(p.requires_grad for p in self.parameters())):
            # (the reference's forward is differentiable; nothing on the render path calls it - silently returning
            # graph-less tensors would hand a caller zero / None gradients)
            raise RuntimeError("deblur4dgs_amd MoveModel.forward returns values only (no autograd graph): differentiate through "
                               "forward_start_end_mid(), or call it under torch.no_grad()")
        lay = [p.detach().contiguous() for p in self._layer_params()]
        tp = self.time_params.detach().contiguous()
        S = 2
        buf = torch.empty(P_input_ch + MLP_ACTS + 12 + S * 12 + 2 * S + 2, device=R.device, dtype=torch.float32)
        enc, acts, delta, RTs, times, dtimes, dT = buf.split([P_input_ch, MLP_ACTS, 12, S * 12, S, S, 2])
        pp = L.MoveModelParams()
        for l in range(9):
            pp.w[l], pp.b[l] = lay[2 * l].data_ptr(), lay[2 * l + 1].data_ptr()
        pp.time_params, pp.n_time_params = tp.data_ptr(), tp.numel()
        out = L.fill(L.MoveModelOut(), enc=enc, acts=acts, delta=delta, RTs=RTs, jac=None, times=times, dtimes=dtimes,
                     deltaT=dT)
        Rc = R.detach().float().contiguous()
        Tc = T.detach().float().reshape(3, -1)[:, 0].contiguous()
        L.check(L.lib().d4gs_move_model_fwd(_p(Rc), Rc.stride(0), _p(Tc), Tc.stride(0), C.byref(pp), S, int(time),
                                            float(time), int(stage == "first"), C.byref(out), _stream(Rc)),
                "move_model_fwd")
        half = dT[:1].clone()
        return delta[:6].view(1, 6).clone(), delta[6:].view(1, 6).clone(), -half, half
