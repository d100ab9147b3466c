"""Seam S1: drop-in for `gsplat.rendering.rasterization` as the reference calls it
(flow3d/scene_model.py:360-373: packed=False, C=1, render_mode in {"RGB","RGB+ED"}).

Same argument names / meaning / return triple as gsplat 1.1.1:
    render_colors [C,H,W,D(+1)], render_alphas [C,H,W,1], info
`info["means2d"]` is an autograd intermediate ([C,N,2], pixel units) on which callers may `.retain_grad()`;
`info["radii"]` is int32 [C,N], > 0 <=> visible.
"""
from __future__ import annotations

import torch

from . import _lib as L
from . import engine
from .engine import RenderCfg, render_instances

_MODES = {"RGB": L.DEPTH_NONE, "RGB+ED": L.DEPTH_ED, "RGB+D": L.DEPTH_D}


def rasterization(
    means: torch.Tensor,  # [N,3]
    quats: torch.Tensor,  # [N,4] wxyz
    scales: torch.Tensor,  # [N,3]
    opacities: torch.Tensor,  # [N]
    colors: torch.Tensor,  # [N,D]
    viewmats: torch.Tensor,  # [C,4,4]
    Ks: torch.Tensor,  # [C,3,3]
    width: int,
    height: int,
    near_plane: float = 0.01,
    far_plane: float = 1e10,
    radius_clip: float = 0.0,
    eps2d: float = 0.3,
    sh_degree=None,
    packed: bool = False,
    tile_size: int = 16,
    backgrounds: torch.Tensor | None = None,  # [C,D]
    render_mode: str = "RGB",
    sparse_grad: bool = False,
    absgrad: bool = False,
    rasterize_mode: str = "classic",
    channel_chunk: int = 32,
    exact_cull: bool = True,
    lazy_sort: bool | None = None,
    near_target: int = 0,
    exact_tiles: bool | None = None,
    **_ignored,
):
    if viewmats.shape[0] != 1 or Ks.shape[0] != 1:
        raise ValueError("C must be 1 (the reference asserts it: flow3d/scene_model.py:249)")
    if sh_degree is not None or packed or absgrad or sparse_grad or rasterize_mode != "classic" or tile_size != 16:
        raise NotImplementedError("only the configuration the reference uses is implemented: "
                                  "sh_degree=None, packed=False, classic, tile_size=16")
    if render_mode not in _MODES:
        raise ValueError(f"render_mode {render_mode!r} not supported (RGB, RGB+ED, RGB+D)")
    N = means.shape[0]
    assert quats.shape == (N, 4) and scales.shape == (N, 3) and opacities.shape == (N,) and colors.shape[0] == N
    bg = None if backgrounds is None else backgrounds[0]
    # any channel count: the engine composites it in chunks of <= 16 channels over one projection / one set of sorted
    # tile lists (engine.channel_chunks), like gsplat's `channel_chunk`
    if lazy_sort is None:  # `info["flatten_ids"]` is part of this seam: lazy lists (unsorted behind a tile's last contributor) only on request
        lazy_sort = engine.LAZY_SORT == "1"
    if exact_tiles is None:  # likewise `tiles_per_gauss` / the lists: the per-tile ellipse test only on request (or D4GS_EXACT_TILES=1)
        exact_tiles = engine.EXACT_TILES == "1"
    cfg = RenderCfg(N=N, G=0, K=0, T=0, S=1, D=colors.shape[-1], width=width, height=height,
                    depth_mode=_MODES[render_mode], flags=0, near_plane=near_plane, far_plane=far_plane, eps2d=eps2d,
                    radius_clip=radius_clip, exact_cull=exact_cull, lazy_sort=lazy_sort, near_target=near_target,
                    exact_tiles=exact_tiles)
    rc, ra, means2d, radii, st = render_instances(cfg, means, quats, scales, opacities, colors, None, None, None,
                                                  None, None, viewmats[0], Ks[0], bg)
    tw, th = cfg.tiles
    info = {
        "means2d": means2d,
        "radii": radii,
        "depths": st.proj_out["depths"],
        "conics": st.proj_out["conics"],
        "opacities": st.proj_out["opac_act"][None],
        "tiles_per_gauss": st.proj_out["tiles_touched"].view(1, N),
        "flatten_ids": st.isect["sorted_gid"][: st.n_isect],
        "isect_offsets": st.proj_out["tile_offsets"][:-1].view(1, th, tw),
        "last_ids": st.raster["last_ids"],
        "n_isect_dev": st.proj_out.get("n_isect"),  # device int64[4]: {count, longest tile list, sampled entries, sampled live entries} (include/d4gs.h)
        "width": width, "height": height, "tile_size": 16, "tile_width": tw, "tile_height": th, "n_cameras": 1,
        "n_isect": st.n_isect,
    }
    return rc, ra, info
