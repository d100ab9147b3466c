"""Seeded synthetic scenes for tests and bench.py (SURVEY.md section 8d "Synthetic inputs").

No dataset or checkpoint is reachable (no network); every measurement and parity test runs on these
distributions.  All tensors are created on CPU with a `torch.Generator` and moved by the caller.
"""
from __future__ import annotations

import math

import torch


def make_scene(N: int, G: int, K: int, S: int, W: int, H: int, seed: int, T: int = 24, D: int = 3,
               frame_t: float = 3.0, delta_t: float = 0.5, cam_jitter: float = 0.002, dtype=torch.float32):
    """N Gaussians of which the first G are dynamic (foreground), K motion bases, S exposure sub-samples.
    Returns a dict of RAW leaf parameters + per-sub-sample host inputs."""
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.rand(*s, generator=g, dtype=torch.float64)
    n = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)
    fx = fy = float(W)
    z = 2.0 + 8.0 * r(N)
    x = z * (1.1 * r(N) - 0.55) * (W / fx)
    y = z * (1.1 * r(N) - 0.55) * (H / fy)
    means = torch.stack([x, y, z], -1)
    quats = n(N, 4)
    scales = math.log(0.01) + 0.5 * n(N, 3)
    opacities = 1.5 * n(N)
    colors = n(N, D)
    sc = dict(means=means, quats=quats, scales=scales, opacities=opacities, colors=colors)
    if G > 0:
        sc["motion_coefs"] = n(G, K)
        dirs = torch.nn.functional.normalize(n(K, 3), dim=-1)
        tau = torch.arange(T, dtype=torch.float64)
        sc["transls"] = 0.05 * tau[None, :, None] * dirs[:, None, :] + 0.01 * n(K, T, 3)
        sc["rots"] = torch.tensor([1.0, 0, 0, 0, 1, 0], dtype=torch.float64) + 0.05 * n(K, T, 6)
    if S > 1:
        sc["times"] = torch.linspace(frame_t - delta_t, frame_t + delta_t, S, dtype=torch.float64)
    else:
        sc["times"] = torch.tensor([frame_t], dtype=torch.float64)
    xi = cam_jitter * n(S, 6)
    sc["RTs"] = _se3_exp(xi)
    sc["viewmat"] = torch.eye(4, dtype=torch.float64)
    sc["K"] = torch.tensor([[fx, 0, W / 2], [0, fy, H / 2], [0, 0, 1]], dtype=torch.float64)
    out = {k: v.to(dtype) for k, v in sc.items()}
    out.update(N=N, G=G, Kb=K, S=S, W=W, H=H, T=T, D=D)
    return out


def _se3_exp(wu: torch.Tensor) -> torch.Tensor:
    """[w(3) rot, u(3) trans] -> [S,3,4] (Rodrigues); used only to synthesise small camera deltas."""
    w, u = wu[:, :3], wu[:, 3:]
    th = w.norm(dim=-1)[:, None, None].clamp(min=1e-12)
    O = torch.zeros_like(w[:, 0])
    wx = torch.stack([torch.stack([O, -w[:, 2], w[:, 1]], -1), torch.stack([w[:, 2], O, -w[:, 0]], -1),
                      torch.stack([-w[:, 1], w[:, 0], O], -1)], -2)
    I = torch.eye(3, dtype=wu.dtype)[None]
    A, B, C = torch.sin(th) / th, (1 - torch.cos(th)) / th**2, (th - torch.sin(th)) / th**3
    R = I + A * wx + B * wx @ wx
    V = I + B * wx + C * wx @ wx
    return torch.cat([R, V @ u[..., None]], -1)
