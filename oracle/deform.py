"""Oracle: per-Gaussian activations, motion-basis deformation, pose compose, camera delta.

Test infrastructure (see oracle/__init__.py).  Plain torch, dtype-generic (fp32 / fp64),
differentiable by autograd.  Every function cites the reference lines it restates.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


# ---------------------------------------------------------------------------------------------
# a1 - activations                                              flow3d/params.py:39-43,70-84
# ---------------------------------------------------------------------------------------------
def act_quats(raw: torch.Tensor) -> torch.Tensor:
    """`F.normalize(x, dim=-1, p=2)` (params.py:39); eps = 1e-12 clamp on the norm."""
    return F.normalize(raw, dim=-1, p=2)


def act_colors(raw: torch.Tensor) -> torch.Tensor:
    return torch.sigmoid(raw)  # params.py:40


def act_scales(raw: torch.Tensor) -> torch.Tensor:
    return torch.exp(raw)  # params.py:41


def act_opacities(raw: torch.Tensor) -> torch.Tensor:
    return torch.sigmoid(raw)  # params.py:42


def act_coefs(raw: torch.Tensor) -> torch.Tensor:
    return F.softmax(raw, dim=-1)  # params.py:43


# ---------------------------------------------------------------------------------------------
# a3 - 6-D rotation -> matrix                                   flow3d/transforms.py:41-53
# ---------------------------------------------------------------------------------------------
def cont_6d_to_rmat(r6: torch.Tensor) -> torch.Tensor:
    """Gram-Schmidt; x, y, z stacked as COLUMNS (`dim=-1`)."""
    x1 = r6[..., 0:3]
    y1 = r6[..., 3:6]
    x = F.normalize(x1, dim=-1)
    y = F.normalize(y1 - (y1 * x).sum(dim=-1, keepdim=True) * x, dim=-1)
    z = torch.linalg.cross(x, y, dim=-1)
    return torch.stack([x, y, z], dim=-1)


# ---------------------------------------------------------------------------------------------
# a2 - MotionBases.compute_transforms                           flow3d/params.py:142-180
# ---------------------------------------------------------------------------------------------
def time_lerp_indices(ts: torch.Tensor, num_frames: int):
    """floor/ceil frame indices clamped to [0, T-1] and the lerp weight w = ts - clamp(floor(ts))
    (params.py:152-153,173; note w uses the CLAMPED floor)."""
    f = torch.floor(ts).clamp(0.0, num_frames - 1).int()
    c = torch.ceil(ts).clamp(0.0, num_frames - 1).int()
    w = ts - f
    return f.long(), c.long(), w


def compute_transforms(
    ts: torch.Tensor, coefs: torch.Tensor, rots: torch.Tensor, transls: torch.Tensor
) -> torch.Tensor:
    """ts (B,) or (1,B); coefs (G,K) activated; rots (K,T,6); transls (K,T,3) -> (G,B,3,4)."""
    if ts.dim() == 1:
        ts = ts[None]
    f, c, w = time_lerp_indices(ts, transls.shape[1])
    tr_f = torch.einsum("pk,kni->pni", coefs, transls[:, f[0]])
    r6_f = torch.einsum("pk,kni->pni", coefs, rots[:, f[0]])
    tr_c = torch.einsum("pk,kni->pni", coefs, transls[:, c[0]])
    r6_c = torch.einsum("pk,kni->pni", coefs, rots[:, c[0]])
    w = w.expand(coefs.shape[0], -1)[..., None]
    tr = (1.0 - w) * tr_f + w * tr_c
    r6 = (1.0 - w) * r6_f + w * r6_c
    return torch.cat([cont_6d_to_rmat(r6), tr[..., None]], dim=-1)


# ---------------------------------------------------------------------------------------------
# roma 1.5.0 restatement (XYZW storage)       call sites flow3d/scene_model.py:94-101
# [RECALLED - parity unpinned: roma is absent from /root/reference and from this image]
# ---------------------------------------------------------------------------------------------
def quat_wxyz_to_xyzw(q: torch.Tensor) -> torch.Tensor:
    return torch.cat([q[..., 1:], q[..., :1]], dim=-1)


def quat_xyzw_to_wxyz(q: torch.Tensor) -> torch.Tensor:
    return torch.cat([q[..., 3:], q[..., :3]], dim=-1)


def quat_product_xyzw(p: torch.Tensor, q: torch.Tensor) -> torch.Tensor:
    """Hamilton product p (x) q, XYZW."""
    pv, pw = p[..., :3], p[..., 3:]
    qv, qw = q[..., :3], q[..., 3:]
    pvb, qvb = torch.broadcast_tensors(pv, qv)
    vec = pw * qv + qw * pv + torch.linalg.cross(pvb, qvb, dim=-1)
    last = pw * qw - (pv * qv).sum(dim=-1, keepdim=True)
    return torch.cat([vec, last], dim=-1)


def rotmat_to_unitquat_xyzw(R: torch.Tensor) -> torch.Tensor:
    """Shepperd / scipy 4-way branch on argmax(R00, R11, R22, trace), then normalise."""
    batch = R.shape[:-2]
    m = R.reshape(-1, 3, 3)
    diag = torch.stack([m[:, 0, 0], m[:, 1, 1], m[:, 2, 2]], dim=-1)
    tr = diag.sum(dim=-1)
    choice = torch.cat([diag, tr[:, None]], dim=-1).argmax(dim=-1)
    cands = []
    for i in range(3):
        j, k = (i + 1) % 3, (i + 2) % 3
        comp = [None] * 4
        comp[i] = 1 - tr + 2 * m[:, i, i]
        comp[j] = m[:, j, i] + m[:, i, j]
        comp[k] = m[:, k, i] + m[:, i, k]
        comp[3] = m[:, k, j] - m[:, j, k]
        cands.append(torch.stack(comp, dim=-1))
    cands.append(
        torch.stack(
            [m[:, 2, 1] - m[:, 1, 2], m[:, 0, 2] - m[:, 2, 0], m[:, 1, 0] - m[:, 0, 1], 1 + tr],
            dim=-1,
        )
    )
    allq = torch.stack(cands, dim=1)  # (n, 4 choices, 4)
    q = allq[torch.arange(m.shape[0]), choice]
    q = q / torch.linalg.norm(q, dim=-1, keepdim=True)
    return q.reshape(*batch, 4)


def quat_wxyz_to_rotmat(q: torch.Tensor) -> torch.Tensor:
    """Rotation matrix of a (not necessarily unit) wxyz quaternion after normalisation."""
    q = q / torch.linalg.norm(q, dim=-1, keepdim=True)
    w, x, y, z = q.unbind(-1)
    return torch.stack(
        [
            torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], -1),
            torch.stack([2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)], -1),
            torch.stack([2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1),
        ],
        dim=-2,
    )


# ---------------------------------------------------------------------------------------------
# a4/a5 - pose compose                                          flow3d/scene_model.py:58-120
# ---------------------------------------------------------------------------------------------
def compute_poses_fg(
    ts: torch.Tensor,
    means: torch.Tensor,
    raw_quats: torch.Tensor,
    raw_coefs: torch.Tensor,
    rots: torch.Tensor,
    transls: torch.Tensor,
):
    """-> means (G,B,3), quats wxyz (G,B,4).  scene_model.py:76-106."""
    quats = act_quats(raw_quats)
    tf = compute_transforms(ts, act_coefs(raw_coefs), rots, transls)  # (G,B,3,4)
    m = torch.einsum("pnij,pj->pni", tf, F.pad(means, (0, 1), value=1.0))
    q = quat_xyzw_to_wxyz(
        quat_product_xyzw(rotmat_to_unitquat_xyzw(tf[..., :3, :3]), quat_wxyz_to_xyzw(quats[:, None]))
    )
    return m, F.normalize(q, p=2, dim=-1)


def compute_poses_all(ts, fg: dict, bases: dict, bg: dict | None):
    """fg/bg: dicts of RAW leaf params (`means`,`quats`,...).  scene_model.py:108-120."""
    m, q = compute_poses_fg(ts, fg["means"], fg["quats"], fg["motion_coefs"], bases["rots"], bases["transls"])
    if bg is not None:
        B = m.shape[1]
        m = torch.cat([m, bg["means"][:, None].expand(-1, B, -1)], dim=0)
        q = torch.cat([q, act_quats(bg["quats"])[:, None].expand(-1, B, -1)], dim=0)
    return m, q


def camera_delta(means: torch.Tensor, RT: torch.Tensor) -> torch.Tensor:
    """means'' = (transR @ means^T + transT)^T; quats are NOT rotated (scene_model.py:352-353)."""
    return (RT[:3, :3] @ means.permute(1, 0) + RT[:3, 3:4]).permute(1, 0)
