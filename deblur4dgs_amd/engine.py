"""Autograd plumbing around libd4gs.so: two `torch.autograd.Function`s that own allocation and enqueue the
HIP kernels on torch's current stream.

  ProjectFn   leaf params (+ bases, times, camera deltas, viewmat)  ->  means2d, conics, depths  [S,N,...],
              activated opacities / colour table.   fwd: d4gs_project_fwd     bwd: d4gs_project_bwd
  RasterFn    those per-instance tensors                            ->  render_colors [S,H,W,D'], alphas
              fwd: d4gs_bin_sort + d4gs_raster_fwd                   bwd: d4gs_raster_bwd

`means2d` is a real autograd intermediate between the two, so `means2d.retain_grad()` / `.grad` behave exactly as
with gsplat's `info["means2d"]` (reference flow3d/scene_model.py:456-461, flow3d/trainer.py:975).

PyTorch here is plumbing (device memory, streams, autograd graph); all arithmetic happens in the HIP library.
There is no eager / CPU fallback: tensors must live on a ROCm device.
"""
from __future__ import annotations

import os

import collections
import ctypes as C
import threading
from dataclasses import dataclass, field, replace

import torch

from . import _lib as L


@dataclass
class RenderCfg:
    N: int
    G: int
    K: int
    T: int
    S: int
    D: int
    width: int
    height: int
    depth_mode: int = L.DEPTH_NONE
    flags: int = 0
    n_sigmoid: int = 0
    near_plane: float = 0.01
    far_plane: float = 1e10
    eps2d: float = 0.3
    radius_clip: float = 0.0
    exact_cull: bool = True
    optimistic_sizes: bool = True  # size the intersection lists from the previous call's count, verify afterwards
    deferred_size_check: bool = False  # never wait for the device-side count in this call: size the lists from the
    #                                  previous call's count (+25 %), let every kernel check it on the device, and look
    #                                  at the count at the NEXT call of this shape (or `check_deferred()`); an overflow
    #                                  raises there, like an asynchronous device error.  No host stall, and the whole
    #                                  render (forward + backward) can be captured in a HIP graph after one warm-up call.
    grad_arena: dict | None = None  # optional {"means": tensor, ...}: leaf gradients are written THERE (e.g. views of
    #                                  a flat all-reduce buffer) instead of fresh tensors; shapes / dtype must match
    control_stats: dict | None = None  # optional densification-statistics sink, updated by the backward's gather epilogue
    #                                  (SURVEY 8f-1): {"xys_grad_norm_acc" f32[N], "vis_count" i64[N], "max_radii" f32[N],
    #                                  "batch_size" int, "update_max_radii" bool}
    lazy_sort: bool | None = None  # D4GS_LAZY_SORT (include/d4gs.h): near / far partition of the tile lists, far parts sorted only
    #                                  for tiles that did not saturate within the near part.  Same image and gradients bit for bit, but the tail of a list behind
    #                                  its tile's last contributor is then left UNSORTED in `flatten_ids`.  None -> resolved once per
    #                                  render by `resolve_lazy` (D4GS_LAZY_SORT=auto, the default: on when the previous render of the
    #                                  shape measured < 50 % live rows and lists of >= 1000 keys on average)
    near_target: int = 0  # keys the near part of a list is aimed at (0: the library's 1024)
    exact_tiles: bool | None = None  # D4GS_EXACT_TILES (include/d4gs.h): bin a splat only into the tiles of its tight rectangle that its
    #                                  alpha >= 1/255 ellipse reaches.  Same image and gradients bit for bit, shorter lists; the test costs the
    #                                  projection ~1 instruction per candidate pair and lane, so None (D4GS_EXACT_TILES=auto) turns it on from
    #                                  EXACT_TILES_FROM intersections per instance, measured by the previous render of the shape

    @property
    def DP(self) -> int:
        return (self.D + 3) // 4 * 4

    @property
    def NCH(self) -> int:
        return self.D + (1 if self.depth_mode != L.DEPTH_NONE else 0)

    @property
    def tiles(self) -> tuple[int, int]:
        return (self.width + L.TILE - 1) // L.TILE, (self.height + L.TILE - 1) // L.TILE

    def dims(self) -> L.Dims:
        return L.Dims(self.N, self.G, self.K, self.T, self.S, self.D, self.width, self.height, self.depth_mode,
                      self.flags | (L.EXACT_CULL if self.exact_cull else 0) | (L.LAZY_SORT if self.lazy_sort else 0)
                      | (L.EXACT_TILES if (self.exact_tiles and self.exact_cull) else 0), self.n_sigmoid, self.near_plane, self.far_plane,
                      self.eps2d, self.radius_clip)


@dataclass
class State:
    """Non-differentiable buffers shared by the stages (all owned by torch).  One projection + one binning / depth sort
    serve every channel chunk of a render (`raster` is the first chunk's composite state; `last_ids`, `final_T` and the
    alphas are the same for all chunks)."""
    cfg: RenderCfg
    proj_in: dict = field(default_factory=dict)
    proj_out: dict = field(default_factory=dict)
    isect: dict = field(default_factory=dict)
    raster: dict = field(default_factory=dict)
    n_isect: int = -1
    max_tile: int = -1
    binned: bool = False
    # one-call path (FrameFn)
    ws: object = None
    ws_ptr: int = 0
    ws_bytes: int = 0
    want_grad: bool = True  # False: the render was issued under torch.no_grad() - FrameFn allocates the forward workspace only
    ws_cap: tuple = (0, 0)
    frame_io: dict = field(default_factory=dict)
    policy: object = None
    xys_sink: list | None = None
    v_means2d: object = None


# "auto" | "dense" | "sparse": gradient-row mode of the composite backward (D4gsRasterGrads.row_mode); "auto" lets the library
# choose from the list capacity per instance.  Module-level so tests can force it; D4GS_BWD_ROWS (read once, at import) runs a whole
# test session in one mode.
BWD_ROWS = os.environ.get("D4GS_BWD_ROWS", "auto")
assert BWD_ROWS in ("auto", "dense", "sparse"), f"D4GS_BWD_ROWS={BWD_ROWS!r}"

SUPPORTED_D = (1, 2, 3, 4, 5, 8, 16)  # colour-channel instantiations of the composite kernels (+ optional depth)


def channel_chunks(D: int) -> list[tuple[int, int, int]]:
    """[(first channel, end channel, kernel width)]: any channel count is rendered as chunks of <= 16 channels that
    share the projection and the sorted tile lists (gsplat's `channel_chunk` idea); a ragged last chunk is zero-padded
    to the next instantiated width.  The depth channel, if any, rides on the last chunk."""
    out, c = [], 0
    while D - c > SUPPORTED_D[-1]:
        out.append((c, c + SUPPORTED_D[-1], SUPPORTED_D[-1]))
        c += SUPPORTED_D[-1]
    out.append((c, D, next(d for d in SUPPORTED_D if d >= max(D - c, 1))))
    return out


raw_stream = L.raw_stream


def _stream():
    return C.c_void_p(raw_stream())


# (device, S, W, H, N bucket) -> (list capacity, bound on the longest tile list).  N is bucketed to its two leading bits
# so that densification (N changes every control step) reuses the entry; bounded LRU; guarded by a lock because the
# reference's viewer thread renders concurrently with training (flow3d/trainer.py:204-207).
_SIZE_GUESS: "collections.OrderedDict" = collections.OrderedDict()
_SIZE_GUESS_MAX = 64
_SIZE_LOCK = threading.Lock()
_SIZE_STATS = {"calls": 0, "relaunched": 0}
_PINNED: dict = {}


def _sort_class(max_tile: int) -> int:
    """Upper bound on the longest tile list handed to d4gs_bin_sort (with 50 % headroom), rounded to a sort size class;
    0 = unknown (every class is launched)."""
    for c in (512, 2048, 4096, 8192, 16384):
        if 3 * max_tile <= 2 * c:
            return c
    return 0


def _size_key(dev, S, N, W, H):
    shift = max(N.bit_length() - 2, 0)
    return (dev.index, S, W, H, (N >> shift) << shift)


def _list_key(dev, cfg):
    """Key of the list-size guess, the deferred records and the graph watch records of a render: the shape's size key plus the
    RESOLVED exact-tiles flag.  Lists binned with D4GS_EXACT_TILES are 15-40 % shorter than the rectangles' (and may fall in a
    smaller sort class), so a capacity measured under one flag says nothing about the other (a shared key let an
    `auto` on -> off flip launch a deferred render at the stale, too small capacity).  The live fraction stays per shape."""
    return _size_key(dev, cfg.S, cfg.N, cfg.width, cfg.height) + (bool(cfg.exact_tiles and cfg.exact_cull),)


def _guess_drop(key):
    with _SIZE_LOCK:
        _SIZE_GUESS.pop(key, None)


def _guess_get(key):
    with _SIZE_LOCK:
        g = _SIZE_GUESS.get(key)
        if g is not None:
            _SIZE_GUESS.move_to_end(key)
        return g


def _guess_put(key, val):
    with _SIZE_LOCK:
        _SIZE_GUESS[key] = val
        _SIZE_GUESS.move_to_end(key)
        while len(_SIZE_GUESS) > _SIZE_GUESS_MAX:
            _SIZE_GUESS.popitem(last=False)


# Live fraction of the intersection lists, per size key: the forward composite counts the list entries at or in front of
# their tile's last contributor, on a fixed sample of tiles (D4gsProjOut.n_isect[2..3]) - the rows the backward will replay.  Occluded / large-footprint
# scenes replay a small part of their lists and want the backward's SPARSE row mode, whatever their rows per instance
# (cfg2 with 2x splats: 16 % live at 3.2 rows per instance, sparse 7 % faster; cfg3: 91 % live at 4.2, dense 5 % faster).  The
# count arrives with the list sizes, so the choice is made from the previous render of that shape.
_LIVE_FRAC: collections.OrderedDict = collections.OrderedDict()
LIVE_SPARSE_BELOW = 0.5


def _live_put(key, live, sampled):
    if sampled <= 0:  # nothing sampled (empty lists): no evidence, keep what is known
        return
    key = key[:5]  # (a list key: the shape's size key + the exact-tiles flag)
    with _SIZE_LOCK:
        _LIVE_FRAC[key] = live / sampled  # (a wide render's channel chunks count the same tiles once each: the ratio is unchanged)
        _LIVE_FRAC.move_to_end(key)
        while len(_LIVE_FRAC) > _SIZE_GUESS_MAX:
            _LIVE_FRAC.popitem(last=False)


# "auto" (default): by the measured live fraction, "1" always, "0" never.  `rasterization()` - the gsplat seam, whose `info` exposes
# the sorted lists - only follows "1" or an explicit lazy_sort=True (behind a tile's last contributor a lazy list is unsorted)
LAZY_SORT = os.environ.get("D4GS_LAZY_SORT", "auto")
assert LAZY_SORT in ("0", "1", "auto"), f"D4GS_LAZY_SORT={LAZY_SORT!r}"
# "auto" turns it on below this live-row fraction and from this many keys per tile list, both measured by the previous render of the
# shape (A/B hook: D4GS_LAZY_AUTO="0.5,1000").  profiles/r04t_lazy_auto_stats.txt: cfg2 (95 % live, 940 keys) and cfg3 (88 %) and the
# reference's training shape (100 %) stay off; cfg2 with 2x splats (16 %, 1 670) 1.113 -> 1.095 ms, with 4x (4 %, 3 870) 1.52 -> 1.11,
# cfg5 (20 %, 1 150) 9.7 -> 9.08, the training shape with 4x splats (9 %, 1 840) 1.79 -> 1.73 - no workload measured loses
LAZY_AUTO_LIVE, LAZY_AUTO_KEYS = (float(x) for x in os.environ.get("D4GS_LAZY_AUTO", "0.5,1000").split(","))


# "auto" (default) | "0" | "1"
EXACT_TILES = os.environ.get("D4GS_EXACT_TILES", "auto")
assert EXACT_TILES in ("0", "1", "auto"), f"D4GS_EXACT_TILES={EXACT_TILES!r}"
EXACT_TILES_FROM = float(os.environ.get("D4GS_EXACT_TILES_FROM", "3.5"))  # intersections per (sub-sample, Gaussian) instance
EXACT_TILES_MIN_LIVE = 0.5
EXACT_TILES_LIVE_HYST = 0.1
_XT_ON: dict = {}  # size key -> bool: the shape's current choice (hysteresis: the test itself shortens the lists it is decided from)


def resolve_lazy(cfg, dev):
    """Fix cfg.lazy_sort / cfg.near_target / cfg.exact_tiles for one render (once: forward and backward must agree on D4gsDims.flags)."""
    if cfg.exact_tiles is None:
        cfg.exact_tiles = EXACT_TILES == "1"
        if EXACT_TILES == "auto":  # the list capacity of the shape is 1.25 x the previous render's count (+ 4096)
            key = _size_key(dev, cfg.S, cfg.N, cfg.width, cfg.height)
            with _SIZE_LOCK:
                was = _XT_ON.get(key, False)
                live = _LIVE_FRAC.get(key)
            guess = _guess_get(key + (was,))  # the count measured under the flag the shape's last render ran with
            per_inst = ((guess[0] - 4096) / 1.25 / (cfg.S * max(cfg.N, 1))) if guess else 0.0
            # on from 3.5 per instance, off again below 2.1 (the test itself shortens the lists it is decided from)
            on = per_inst >= (0.6 * EXACT_TILES_FROM if was else EXACT_TILES_FROM)
            # ... but never for few-tile launches (their depth-segmented backward places its hand-offs by list length: shorter
            # lists would move gradient bits between renders of one scene) and only where most list entries are alive - the
            # composites must gain more than the test costs the projection (profiles/r05_ab_exact_tiles.txt: cfg3, 88 % live,
            # -1.1 %; the training shape at 720p -1.8 %; cfg5, 20 % live: +-0; cfg2 with 2x / 4x splats, 16 / 4 % live: +1.5 / +12 %).
            # Hysteresis on the live fraction too (on from 0.5, off below 0.4): a scene hovering at the line must not flap.
            min_live = EXACT_TILES_MIN_LIVE - (EXACT_TILES_LIVE_HYST if was else 0.0)
            on = on and seg_state_elems(cfg) == 0 and (live is None or live >= min_live)
            if was and not on:
                # on -> off: the rectangles' lists are longer by an unbounded factor (a thin diagonal splat), and whatever the
                # off key still holds is from before the flag came on.  No guess = the next render of the shape COUNTS first
                # (one host round trip; under stream capture: the explanatory error of _sized_launch) instead of launching blind.
                _guess_drop(key + (False,))
            with _SIZE_LOCK:
                if len(_XT_ON) > 4 * _SIZE_GUESS_MAX:
                    _XT_ON.clear()
                _XT_ON[key] = on
            cfg.exact_tiles = on
    if cfg.lazy_sort is None:
        cfg.lazy_sort = LAZY_SORT == "1"
        if LAZY_SORT == "auto":
            key = _size_key(dev, cfg.S, cfg.N, cfg.width, cfg.height)
            with _SIZE_LOCK:
                f = _LIVE_FRAC.get(key)
            guess = _guess_get(_list_key(dev, cfg))
            tw, th = cfg.tiles
            avg = (guess[0] / 1.25 / max(cfg.S * tw * th, 1)) if guess else 0.0
            cfg.lazy_sort = f is not None and f < LAZY_AUTO_LIVE and avg >= LAZY_AUTO_KEYS
    if cfg.lazy_sort and cfg.near_target <= 0:
        key = _size_key(dev, cfg.S, cfg.N, cfg.width, cfg.height)
        with _SIZE_LOCK:
            f = _LIVE_FRAC.get(key)
        guess = _guess_get(_list_key(dev, cfg))
        tw, th = cfg.tiles
        if f is not None and guess:  # about 2.5 x the entries a tile consumed on the previous render of the shape
            cfg.near_target = max(256, int(2.5 * f * guess[0] / 1.25 / max(cfg.S * tw * th, 1)))
    return cfg


def _lazy_ws(cfg, dev):
    if not cfg.lazy_sort:
        return None
    z = L.Sizes()
    L.check(L.lib().d4gs_query_sizes(C.byref(cfg.dims()), C.byref(z)), "d4gs_query_sizes")
    return torch.empty(int(z.lazy_ws), dtype=torch.int32, device=dev)


def row_mode_for(cfg, dev):
    """D4gsRasterGrads.row_mode for a render of this shape: the forced mode, or - "auto" - sparse / dense from the live
    fraction its previous render measured, or the library's capacity heuristic while there is no measurement."""
    if BWD_ROWS != "auto":
        return {"dense": L.ROWS_DENSE, "sparse": L.ROWS_SPARSE}[BWD_ROWS]
    with _SIZE_LOCK:
        f = _LIVE_FRAC.get(_size_key(dev, cfg.S, cfg.N, cfg.width, cfg.height))
    if f is None:
        return L.ROWS_AUTO
    return L.ROWS_SPARSE if f < LIVE_SPARSE_BELOW else L.ROWS_DENSE


_DEFERRED: dict = {}  # size key -> [(pinned int64[4], event, capacity, max-tile hint), ...]: EVERY unchecked render of that
#                       shape, oldest first (a training step issues several renders of one shape before any count lands)


_PINNED_FREE: list = []  # pinned int64[4] buffers whose deferred count has been consumed (guarded by _SIZE_LOCK)


def _deferred_poll(key, block: bool = False, keep_last: int = 0):
    """Look at the counts of the earlier deferred renders of this shape that have arrived (all of them when `block`; all but the
    newest `keep_last` when `block` and keep_last > 0).
    Every record is verified - none is dropped when the host runs ahead of the device; an overflow raises after the
    whole batch has been read and the size guess updated.  -> (n, max_tile) of the newest record read | None."""
    with _SIZE_LOCK:
        recs = _DEFERRED.get(key)
        if not recs:
            return None
        ready = []
        while recs and ((block and len(recs) > keep_last) or recs[0][1].query()):  # stream order: an older copy lands before a newer one
            ready.append(recs.pop(0))
        if not recs:
            _DEFERRED.pop(key, None)
    last, bad = None, None
    for host_n, ev, cap, hint in ready:
        ev.synchronize()
        n, max_tile, sampled, live = host_n.tolist()
        if n <= cap and not (hint > 0 and max_tile > hint):  # (an overflowed render composited nothing)
            _live_put(key, live, sampled)
        with _SIZE_LOCK:  # the copy has landed and been read: the pinned pair can carry the next count
            if len(_PINNED_FREE) < 64:
                _PINNED_FREE.append(host_n)
        last = (n, max_tile)
        if (n > cap or (hint > 0 and max_tile > hint)) and bad is None:
            bad = (n, max_tile, cap, hint)
    if last is not None:
        _guess_put(key, (last[0] + last[0] // 4 + 4096, _sort_class(last[1])))
    if bad is not None:
        if bad[0] + bad[0] // 4 + 4096 > (last[0] + last[0] // 4 + 4096):
            _guess_put(key, (bad[0] + bad[0] // 4 + 4096, _sort_class(max(bad[1], last[1]))))
        n, max_tile, cap, hint = bad
        raise RuntimeError(f"deblur4dgs_amd: a render with deferred_size_check needed {n} intersections (longest tile "
                           f"list {max_tile}) but its lists were sized for {cap} (class {hint}): that render's output "
                           "was INVALID (its kernels skipped the work).  Re-run the step; the size guess is updated.")
    return last


_WATCH_TLS = threading.local()  # .watch: the GraphWatch whose capture is in progress ON THIS THREAD (GraphWatch.capturing());
#                                 a capture is a per-thread, per-stream affair - a render captured on another thread must not
#                                 attach its count copy to this one's watch
_WARNED_UNWATCHED = False


def _graph_watch():
    return getattr(_WATCH_TLS, "watch", None)


class GraphWatch:
    """Size checks for a step replayed from a HIP graph.  A captured render keeps the list capacities of its capture, and its
    kernels skip their work when the device-side count exceeds them - nobody would notice: the loss would be computed on
    stale images and the optimizer would step on zero raster gradients.  Capture inside `with watch.capturing():` and every
    deferred render of the step leaves a device-to-pinned copy of its counts IN the graph; `watch.replayed()` after each
    `graph.replay()` and `watch.check()` before the next one (no stall in the steady state: the previous replay has long
    finished) raises RuntimeError - and updates the size guesses - if a render of the previous replay overflowed, so the
    caller re-captures (after an eager step or two, which size the lists from the new counts)."""

    def __init__(self, max_renders: int = 32):
        self.recs = []  # (size key, pinned int64[4], capacity, max-tile hint) per captured render
        self.event = None
        self._pool = [torch.zeros(4, dtype=torch.int64).pin_memory() for _ in range(max_renders)]  # pinned before the capture

    def take(self):
        if not self._pool:
            raise RuntimeError("GraphWatch: more renders in the captured step than max_renders")
        return self._pool.pop()

    def capturing(self):
        import contextlib

        @contextlib.contextmanager
        def cm():
            assert _graph_watch() is None, "nested GraphWatch captures"
            _WATCH_TLS.watch = self
            try:
                yield self
            finally:
                _WATCH_TLS.watch = None

        return cm()

    def replayed(self):
        self.event = torch.cuda.Event()
        self.event.record()

    def check(self):
        if self.event is None:
            return
        self.event.synchronize()
        self.event = None
        bad = None
        for key, host_n, cap, hint in self.recs:
            n, max_tile, sampled, live = host_n.tolist()
            if n > cap or (hint > 0 and max_tile > hint):
                _guess_put(key, (n + n // 4 + 4096, _sort_class(max_tile)))
                bad = bad or (n, max_tile, cap, hint)
            else:
                _live_put(key, live, sampled)
        if bad is not None:
            n, max_tile, cap, hint = bad
            raise RuntimeError(f"deblur4dgs_amd: a render replayed from a HIP graph needed {n} intersections (longest tile list "
                               f"{max_tile}) but the graph was captured with lists for {cap} (class {hint}): that replay's output "
                               "was INVALID (its kernels skipped the work).  Re-capture the step; the size guess is updated.")


def check_deferred(keep_last: int = 0):
    """Wait for and verify every outstanding deferred size check (call once per training step, e.g. where the loss is
    read back anyway).  Raises RuntimeError if any render since the last call overflowed its intersection lists.
    keep_last = n: do not wait for the newest n renders of each shape (they are verified by a later call) - the host then runs up to
    n renders ahead of the device instead of waiting for this step's forward, so a host stall of a few milliseconds does not reach the GPU."""
    err = None
    for key in list(_DEFERRED):
        try:
            _deferred_poll(key, block=True, keep_last=keep_last)
        except RuntimeError as e:  # keep draining the other shapes: their records must not outlive this call
            err = err or e
    if err is not None:
        raise err


def _pinned_counts(dev):
    """One pinned int64[4] per (thread, device, stream): every use is followed by an event wait before the next one on
    that stream, and two streams of one thread never share a buffer."""
    k = (threading.get_ident(), dev.index, raw_stream(dev.index))
    with _SIZE_LOCK:
        if k not in _PINNED:
            if len(_PINNED) > 256:
                _PINNED.clear()
            _PINNED[k] = torch.empty(4, dtype=torch.int64).pin_memory()
        return _PINNED[k]


def _need_gpu(t: torch.Tensor):
    if not t.is_cuda:
        raise RuntimeError("deblur4dgs_amd runs on an MI355X (ROCm) device only; got a CPU tensor (no CPU fallback)")


def _f32c(t):
    return None if t is None else t.detach().to(torch.float32).contiguous()


def _proj_structs(st: State):
    pin = L.fill(L.ProjIn(), **st.proj_in)
    pout = L.fill(L.ProjOut(), **st.proj_out)
    return pin, pout


_PROJ_IN = ("means", "quats", "scales", "opacities", "colors", "motion_coefs", "rots", "transls", "times", "RTs", "viewmat",
            "Kmat")


class ProjectFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, st: State, means, quats, scales, opacities, colors, motion_coefs, rots, transls, times, RTs,
                viewmat, Kmat):
        cfg = st.cfg
        _need_gpu(means)
        dev = means.device
        S, N = cfg.S, cfg.N
        tw, th = cfg.tiles
        f32 = dict(dtype=torch.float32, device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        st.proj_in = dict(means=_f32c(means), quats=_f32c(quats), scales=_f32c(scales), opacities=_f32c(opacities),
                          colors=_f32c(colors), motion_coefs=_f32c(motion_coefs), rots=_f32c(rots),
                          transls=_f32c(transls), times=_f32c(times), RTs=_f32c(RTs), viewmat=_f32c(viewmat),
                          Kmat=_f32c(Kmat))
        lib = L.lib()
        st.proj_out = dict(
            means2d=torch.empty(S, N, 2, **f32), depths=torch.empty(S, N, **f32), conics=torch.empty(S, N, 3, **f32),
            radii=torch.empty(S, N, **i32), opac_act=torch.empty(N, **f32), ctab=torch.empty(N, cfg.DP, **f32),
            geom=torch.empty(S * N, L.GEOM_STRIDE, **f32), tile_rects=torch.empty(S * N, 2, **i32),
            tiles_touched=torch.empty(S * N, **i32),
            isect_offsets=torch.empty(S * N, **i32), lazy_ws=_lazy_ws(cfg, dev),
            tile_counts=torch.empty(2 * S * tw * th, **i32),
            tile_offsets=torch.empty(S * tw * th + 1, **i32), n_isect=torch.empty(4, dtype=torch.int64, device=dev),
            scan_ws=torch.empty(lib.d4gs_scan_ws_elems(S * N), **i32),
            blend_bases=torch.empty(S * cfg.K * 16, **f32) if cfg.G > 0 else None,  # the backward's scalar-load table (include/d4gs.h)
            tile_masks=torch.empty(S * N, dtype=torch.int64, device=dev) if (cfg.exact_tiles and cfg.exact_cull) else None,
        )
        dims = cfg.dims()
        pin, pout = _proj_structs(st)
        L.check(lib.d4gs_project_fwd(C.byref(dims), C.byref(pin), C.byref(pout), _stream()), "d4gs_project_fwd")
        ctx.st = st
        # The backward re-reads the leaves.  `_f32c` hands out detached ALIASES of f32-contiguous leaves, which share the
        # leaf's version counter: routing them through save_for_backward makes an in-place update between forward and
        # backward (optimizer.step, a control step) raise, as stock autograd / gsplat would, instead of silently
        # differentiating modified parameters.  (Writes through `.data` bypass version counters everywhere.)
        ctx.save_for_backward(*[st.proj_in[k] for k in _PROJ_IN])
        ctx.needs = [t is not None and t.requires_grad for t in
                     (means, quats, scales, opacities, colors, motion_coefs, rots, transls, times, RTs, viewmat)]
        o = st.proj_out
        # return ALIASES of the buffers: autograd attaches grad_fn (-> ctx -> st) to the returned objects, and st must
        # not hold those objects or every call leaks a reference cycle until the cyclic GC runs
        outs = tuple(o[k].view(o[k].shape) for k in ("means2d", "conics", "depths", "opac_act", "ctab", "radii"))
        ctx.mark_non_differentiable(outs[5])
        return outs

    @staticmethod
    def backward(ctx, v_means2d, v_conics, v_depths, v_opac_act, v_ctab, _v_radii):
        st: State = ctx.st
        cfg = st.cfg
        o = st.proj_out
        dev = o["means2d"].device
        f32 = dict(dtype=torch.float32, device=dev)
        z = lambda g, ref: torch.zeros_like(ref) if g is None else g.to(torch.float32).contiguous()
        v_means2d, v_conics, v_depths = z(v_means2d, o["means2d"]), z(v_conics, o["conics"]), z(v_depths, o["depths"])
        v_opac_act, v_ctab = z(v_opac_act, o["opac_act"]), z(v_ctab, o["ctab"])
        lib = L.lib()
        dims = cfg.dims()
        pi = dict(zip(_PROJ_IN, ctx.saved_tensors))  # raises if a leaf was modified in place since the forward
        dyn = cfg.G > 0
        arena = cfg.grad_arena or {}

        def buf(name, *shape):
            t = arena.get(name)
            if t is not None and t.shape == torch.Size(shape) and t.dtype == torch.float32 and t.is_contiguous():
                return t.view(shape)  # a fresh alias: autograd takes it over as the leaf's .grad without a copy
            return torch.empty(*shape, **f32)

        g = dict(
            v_means=buf("means", cfg.N, 3), v_quats=buf("quats", cfg.N, 4),
            v_scales=buf("scales", cfg.N, 3), v_opacities=buf("opacities", cfg.N),
            v_colors=buf("colors", cfg.N, cfg.D),
            v_motion_coefs=buf("motion_coefs", cfg.G, cfg.K) if dyn else None,
            v_rots=buf("rots", cfg.K, cfg.T, 6) if dyn else None,
            v_transls=buf("transls", cfg.K, cfg.T, 3) if dyn else None,
            v_times=buf("times", cfg.S) if dyn else None,
            v_RTs=buf("RTs", cfg.S, 3, 4) if pi["RTs"] is not None else None,
            v_viewmat=buf("viewmat", 4, 4),
            partials=torch.empty(lib.d4gs_bwd_partials_elems(C.byref(dims)), **f32),
        )
        pin = L.fill(L.ProjIn(), **pi)
        pout = L.fill(L.ProjOut(), **st.proj_out)
        lg = L.fill(L.LeafGrads(), **g)
        vp = lambda t: C.c_void_p(L.ptr(t))  # bare Python ints would be truncated to 32-bit C ints
        L.check(lib.d4gs_project_bwd(C.byref(dims), C.byref(pin), C.byref(pout), vp(v_means2d), vp(v_conics),
                                     vp(v_depths), vp(v_opac_act), vp(v_ctab), C.byref(lg), _stream()),
                "d4gs_project_bwd")
        outs = [g["v_means"], g["v_quats"], g["v_scales"], g["v_opacities"], g["v_colors"], g["v_motion_coefs"],
                g["v_rots"], g["v_transls"], g["v_times"], g["v_RTs"], g["v_viewmat"]]
        outs = [x if need else None for x, need in zip(outs, ctx.needs)]
        return (None, *outs, None)


def _check_stats(cs: dict, N: int) -> dict:
    acc, vis, mr = cs["xys_grad_norm_acc"], cs["vis_count"], cs["max_radii"]
    if not (acc.dtype == torch.float32 and vis.dtype == torch.int64 and mr.dtype == torch.float32 and
            acc.shape == vis.shape == mr.shape == (N,) and acc.is_contiguous() and vis.is_contiguous() and mr.is_contiguous()):
        raise ValueError(f"control_stats: need contiguous xys_grad_norm_acc f32 / vis_count i64 / max_radii f32 of shape ({N},)")
    if int(cs["batch_size"]) <= 0:
        raise ValueError("control_stats: batch_size must be positive")
    return cs


def _sized_launch(cfg: "RenderCfg", dev, n_isect_dev, launch, count_needs_launch: bool = False, fused_counts: bool = False):
    """The intersection-list size protocol shared by the staged and the one-call path.  `launch(capacity, max_tile_hint)`
    allocates the lists and enqueues binning + rasterization (every kernel checks the device-side count against the
    capacity); `n_isect_dev()` is the device int64[4] {count, longest tile list, live-row sample x 2} of the launch just made (staged path:
    already there, the projection ran before; one-call path, `count_needs_launch`: a capacity-0 call whose list kernels
    all return at once does the counting).  `fused_counts`: `launch` takes a third argument, the address of a pinned int64[4] its
    LAST kernel stores the counts to (D4gsFrameIO.counts_pinned: no d4gs_copy_counts launch behind it; 0 = none wanted).
    -> (capacity or exact count, max-tile value) the backward must use."""
    key = _list_key(dev, cfg)
    if cfg.deferred_size_check:
        capturing = torch.cuda.is_current_stream_capturing()
        if not capturing:
            _deferred_poll(key)  # raises if an earlier render of this shape overflowed
        guess = _guess_get(key)
        if guess is not None:
            # deferred check: launch at the guessed capacity, never wait.  The count travels to pinned memory behind
            # the launches and is looked at by a later call (not under stream capture: a graph replays this shape).
            watch = _graph_watch()
            host_n = None
            if fused_counts:  # the pinned buffer is chosen BEFORE the launch: the frame's last kernel stores the counts there
                if capturing and watch is not None:
                    host_n = watch.take()
                elif not capturing:
                    with _SIZE_LOCK:  # pinned pairs are recycled once their count has been read (pin_memory() costs ~0.1 ms)
                        host_n = _PINNED_FREE.pop() if _PINNED_FREE else None
                    if host_n is None:
                        host_n = torch.empty(4, dtype=torch.int64).pin_memory()
                launch(*guess, 0 if host_n is None else host_n.data_ptr())
            else:
                launch(*guess)
            stored = host_n is not None
            if capturing and watch is None:
                global _WARNED_UNWATCHED
                if not _WARNED_UNWATCHED:
                    _WARNED_UNWATCHED = True
                    import warnings

                    warnings.warn("deblur4dgs_amd: a render is being captured in a HIP graph outside `GraphWatch.capturing()`: its "
                                  "replays keep the captured list capacities and skip their work silently if the scene outgrows them "
                                  "(stale images, zero raster gradients).  Capture inside `with engine.GraphWatch().capturing():` and "
                                  "call watch.replayed() / watch.check() around graph.replay().", RuntimeWarning, stacklevel=3)
            if capturing and watch is not None:
                # the copy of the counts becomes a node of the graph: every replay leaves them in this record's pinned buffer
                # (a raw hipMemcpyAsync into memory pinned BEFORE the capture: neither an allocation nor torch's host-allocator
                # bookkeeping may happen on a capturing stream)
                if not stored:
                    host_n = watch.take()
                    L.check(L.lib().d4gs_copy_counts(L.ptr(n_isect_dev()), host_n.data_ptr(), raw_stream(dev.index)), "copy_counts")
                watch.recs.append((key, host_n, guess[0], guess[1]))
            if not capturing:
                if not stored:
                    with _SIZE_LOCK:  # pinned pairs are recycled once their count has been read (pin_memory() costs ~0.1 ms)
                        host_n = _PINNED_FREE.pop() if _PINNED_FREE else None
                    if host_n is None:
                        host_n = torch.empty(4, dtype=torch.int64).pin_memory()
                    # (a kernel stores the counts into the pinned buffer: a device-to-host copy here is a DMA-engine trip the
                    # backward's kernels would wait for)
                    L.check(L.lib().d4gs_copy_counts(L.ptr(n_isect_dev()), host_n.data_ptr(), raw_stream(dev.index)), "copy_counts")
                ev = torch.cuda.Event()
                ev.record()
                with _SIZE_LOCK:
                    _DEFERRED.setdefault(key, []).append((host_n, ev, guess[0], guess[1]))
            _SIZE_STATS["calls"] += 1
            return guess
    if torch.cuda.is_current_stream_capturing():  # what follows WAITS for the counts on the host: impossible under capture
        raise RuntimeError("deblur4dgs_amd: a render of a shape that has no list-size guess yet (or without deferred_size_check) "
                           "cannot be captured in a HIP graph - its intersection counts would have to be read on the host.  Run one "
                           "eager step of the same shapes with RenderCfg.deferred_size_check=True first, then capture.")
    # The list sizes live on the device.  Read them back through pinned memory; when a previous call of the same
    # shape left a guess, launch binning + rasterization FIRST (sized by the guess, checked on the device) and
    # wait for the counts afterwards, so the GPU never idles on the host round trip (70 us per render).
    guess = _guess_get(key) if cfg.optimistic_sizes else None
    if guess is not None:
        launch(*guess)
    elif count_needs_launch:
        launch(-1, 0)  # capacity < 0: d4gs_forward returns after the projection + counting (no binning / composite / blend launches)
    host_n = _pinned_counts(dev)
    host_n.copy_(n_isect_dev(), non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    ev.synchronize()
    n, max_tile, sampled, live = host_n.tolist()
    if guess is not None and n <= guess[0] and not (guess[1] > 0 and max_tile > guess[1]):
        _live_put(key, live, sampled)  # (the launch in front of this readback composited the frame)
    if guess is None or n > guess[0] or (guess[1] > 0 and max_tile > guess[1]):
        if guess is not None:
            _SIZE_STATS["relaunched"] += 1
        launch(n, max_tile)
    _SIZE_STATS["calls"] += 1
    _guess_put(key, (n + n // 4 + 4096, _sort_class(max_tile)))
    return n, max_tile


_SEG_ELEMS: dict = {}


def seg_state_elems(cfg: RenderCfg) -> int:
    """D4gsSizes.seg_state of the configuration (0: the composite does not use depth segments for it)."""
    key = (cfg.S, cfg.width, cfg.height, cfg.D, cfg.depth_mode)
    n = _SEG_ELEMS.get(key)
    if n is None:
        z = L.Sizes()
        L.check(L.lib().d4gs_query_sizes(C.byref(cfg.dims()), C.byref(z)), "d4gs_query_sizes")
        n = _SEG_ELEMS[key] = int(z.seg_state)
    return n


class RasterFn(torch.autograd.Function):
    """Composite one channel chunk.  `cfg` is the chunk's configuration (its D = the kernel width, depth mode set on
    the last chunk only); the projection outputs and the sorted tile lists come from the shared `st`.  The first chunk
    of a render bins + sorts (st.binned)."""

    @staticmethod
    def forward(ctx, st: State, cfg: RenderCfg, means2d, conics, depths, opac_act, ctab, background):
        dev = means2d.device
        S, H, W = cfg.S, cfg.height, cfg.width
        f32 = dict(dtype=torch.float32, device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        lib = L.lib()
        ctab = ctab.detach().to(torch.float32).contiguous()
        assert ctab.shape == (cfg.N, cfg.DP)
        rst = dict(background=_f32c(background), render_colors=torch.empty(S, H, W, cfg.NCH, **f32),
                   render_alphas=torch.empty(S, H, W, **f32), last_ids=torch.empty(S, H, W, **i32),
                   final_T=torch.empty(S, H, W, **f32))
        dims = cfg.dims()
        n_seg = seg_state_elems(cfg)  # few-tile launches: per-pixel states at depth-segment boundaries (D4gsRaster.seg_state)
        rst["seg_state"] = None
        pout = L.fill(L.ProjOut(), **{**st.proj_out, "ctab": ctab})
        ras = L.fill(L.Raster(), **rst)
        hint_used = [0]
        seg_env = os.environ.get("D4GS_SEG")

        def raster(cap, max_hint):
            hint_used[0] = max_hint
            # the library replays in depth segments only when the longest list spans more than one (max_hint > 256; D4GS_SEG=1 forces
            # it, =0 forbids it): the boundary-state buffer - 85 MB for a 288x512 S = 1 17-channel render - is allocated only then
            # (the forward and the backward of the render see the same pointer: ctx.rst)
            want_seg = n_seg > 0 and seg_env != "0" and (max_hint > 256 or seg_env == "1")
            if want_seg and rst["seg_state"] is None:
                rst["seg_state"] = torch.empty(n_seg, **f32)
            elif not want_seg:
                rst["seg_state"] = None
            ras.seg_state = L.ptr(rst["seg_state"])
            isect = L.fill(L.Isect(), **st.isect)
            isect.n_isect, isect.max_tile_count, isect.near_target = max(cap, 1), max_hint, cfg.near_target
            L.check(lib.d4gs_raster_fwd(C.byref(dims), C.byref(pout), C.byref(isect), C.byref(ras), _stream()),
                    "d4gs_raster_fwd")

        def launch(cap, max_hint):
            m = max(cap, 1)
            st.isect = dict(keys=torch.empty(m, dtype=torch.int64, device=dev), gid_of_emit=torch.empty(m, **i32),
                            sorted_gid=torch.empty(m, **i32), sorted_emit=torch.empty(m, **i32))
            isect = L.fill(L.Isect(), **st.isect)
            isect.n_isect, isect.max_tile_count, isect.near_target = m, max_hint, cfg.near_target
            L.check(lib.d4gs_bin_sort(C.byref(dims), C.byref(pout), C.byref(isect), _stream()), "d4gs_bin_sort")
            raster(cap, max_hint)

        if st.binned:
            raster(st.n_isect, st.max_tile)
        else:
            st.n_isect, st.max_tile = _sized_launch(cfg, dev, lambda: st.proj_out["n_isect"], launch)
            st.binned = True
            st.raster = rst
        ctx.st, ctx.cfg, ctx.rst, ctx.ctab = st, cfg, rst, ctab
        ctx.max_hint = hint_used[0]  # the longest-list bound the forward composited under: the backward must see the same one
        return rst["render_colors"].view(S, H, W, cfg.NCH), rst["render_alphas"].unsqueeze(-1)

    @staticmethod
    def backward(ctx, v_colors, v_alphas):
        st: State = ctx.st
        cfg, rst = ctx.cfg, ctx.rst
        dev = rst["render_colors"].device
        f32 = dict(dtype=torch.float32, device=dev)
        lib = L.lib()
        S, N = cfg.S, cfg.N
        if v_colors is None:
            v_colors = torch.zeros_like(rst["render_colors"])
        v_colors = v_colors.to(torch.float32).contiguous()
        v_alphas = None if v_alphas is None else v_alphas.to(torch.float32).contiguous()
        g = dict(
            v_render_colors=v_colors, v_render_alphas=v_alphas,
            isect_grad=torch.empty(max(st.n_isect, 1), 6 + cfg.NCH, **f32),
            isect_live=torch.empty((max(st.n_isect, 1) + 3) // 4 * 4, dtype=torch.uint8, device=dev),
            v_means2d=torch.empty(S, N, 2, **f32), v_conics=torch.empty(S, N, 3, **f32),
            v_depths=torch.empty(S, N, **f32), v_opac_act=torch.empty(N, **f32), v_ctab=torch.empty(N, cfg.DP, **f32),
        )
        dims = cfg.dims()
        pout = L.fill(L.ProjOut(), **{**st.proj_out, "ctab": ctx.ctab})
        isect = L.fill(L.Isect(), **st.isect)
        isect.n_isect, isect.max_tile_count = st.n_isect, ctx.max_hint  # (segments on / off follow it: D4gsRaster.seg_state)
        ras = L.fill(L.Raster(), **rst)
        rg = L.fill(L.RasterGrads(), **g)
        rg.row_mode = row_mode_for(cfg, dev)
        if cfg.control_stats is not None:  # fused statistics: the gather epilogue owns v_means2d and reads radii
            cs = _check_stats(cfg.control_stats, N)
            rg.stats_grad_norm_acc, rg.stats_vis_count = L.ptr(cs["xys_grad_norm_acc"]), L.ptr(cs["vis_count"])
            rg.stats_max_radii = L.ptr(cs["max_radii"])
            rg.stats_batch_size, rg.stats_update_max_radii = int(cs["batch_size"]), int(bool(cs.get("update_max_radii", False)))
        L.check(lib.d4gs_raster_bwd(C.byref(dims), C.byref(pout), C.byref(isect), C.byref(ras), C.byref(rg), _stream()),
                "d4gs_raster_bwd")
        return None, None, g["v_means2d"], g["v_conics"], g["v_depths"], g["v_opac_act"], g["v_ctab"], None


class FrameFn(torch.autograd.Function):
    """The whole path as ONE autograd node over the one-call C entry points (d4gs_forward / d4gs_backward):
    leaves -> (blended [H,W,D'] | None, acc [H,W] | None, renders [S,H,W,D'], alphas [S,H,W]), all differentiable (the
    reference puts losses on the blurry frame AND on the sub-sample images, trainer.py:575-618).  One workspace tensor,
    one ctypes call each way - the staged ProjectFn / RasterFn / BlendFn chain costs ~30 allocations, five calls and three
    Python autograd nodes per direction, which is what bounds small scenes (BASELINE cfg1) and the sharded step.  Bit-identical
    to the staged chain (same kernels, same order).  `means2d` is not an autograd intermediate here: its gradient - the
    densification side channel (flow3d/scene_model.py:456-461, trainer.py:975) - is deposited by the backward into
    `st.xys_sink` (a list of S leaf tensors [1,N,2], see SceneModel.render) and kept as `st.v_means2d`."""

    @staticmethod
    def forward(ctx, st: State, blend_policy, means, quats, scales, opacities, colors, motion_coefs, rots, transls, times, RTs,
                viewmat, Kmat, background):
        cfg = st.cfg
        _need_gpu(means)
        dev = means.device
        S, N, H, W, NCH = cfg.S, cfg.N, cfg.height, cfg.width, cfg.NCH
        f32 = dict(dtype=torch.float32, device=dev)
        lib = L.lib()
        st.proj_in = dict(means=_f32c(means), quats=_f32c(quats), scales=_f32c(scales), opacities=_f32c(opacities),
                          colors=_f32c(colors), motion_coefs=_f32c(motion_coefs), rots=_f32c(rots),
                          transls=_f32c(transls), times=_f32c(times), RTs=_f32c(RTs), viewmat=_f32c(viewmat),
                          Kmat=_f32c(Kmat))
        blended = blend_policy is not None
        io = dict(blended=torch.empty(H, W, NCH, **f32) if blended else None, acc=torch.empty(H, W, **f32) if blended else None,
                  renders=torch.empty(S, H, W, NCH, **f32), alphas=torch.empty(S, H, W, **f32),
                  means2d=torch.empty(S, N, 2, **f32), radii=torch.empty(S, N, dtype=torch.int32, device=dev),
                  n_isect=torch.empty(4, dtype=torch.int64, device=dev), background=_f32c(background))
        dims = cfg.dims()
        pin = L.fill(L.ProjIn(), **st.proj_in)
        fio = L.fill(L.FrameIO(), **io)
        fio.near_target = cfg.near_target
        pol = (C.c_int32 * NCH)(*blend_policy) if blended else None
        if blended:
            fio.policy = pol

        # a forward nobody will differentiate (torch.no_grad(): validation, the viewer's render_view) needs none of the backward's
        # scratch - about half of the workspace, pinned by the returned state for as long as the caller keeps it
        ws_bytes_of = lib.d4gs_frame_workspace_bytes if st.want_grad else lib.d4gs_frame_workspace_bytes_fwd

        def launch(cap, max_hint, counts_ptr=0):
            nbytes = ws_bytes_of(C.byref(dims), cap)
            st.ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=dev)
            st.ws_ptr = (st.ws.data_ptr() + 255) & ~255
            st.ws_bytes = nbytes
            st.ws_cap = (cap, max_hint)  # the workspace layout follows the capacity: the backward must carve it the same way
            fio.counts_pinned = counts_ptr or None  # (deferred size check: the frame's last kernel reports the list sizes itself)
            L.check(lib.d4gs_forward(C.byref(dims), C.byref(pin), C.byref(fio), C.c_void_p(st.ws_ptr), nbytes, cap, max_hint,
                                     _stream()), "d4gs_forward")

        st.n_isect, st.max_tile = _sized_launch(cfg, dev, lambda: io["n_isect"], launch, count_needs_launch=True, fused_counts=True)
        st.binned = True
        st.frame_io, st.policy = io, pol
        ctx.st = st
        ctx.set_materialize_grads(False)  # an output the loss does not use must arrive as None, not as a zero-filled stack
        ctx.save_for_backward(*[st.proj_in[k] for k in _PROJ_IN])  # version-checked, like ProjectFn
        ctx.needs = [t is not None and t.requires_grad for t in
                     (means, quats, scales, opacities, colors, motion_coefs, rots, transls, times, RTs, viewmat)]
        ra = (io["renders"].view(S, H, W, NCH), io["alphas"].view(S, H, W))
        if blended:
            return (io["blended"].view(H, W, NCH), io["acc"].view(H, W), *ra)
        return (None, None, *ra)

    @staticmethod
    def backward(ctx, v_blended, v_acc, v_renders, v_alphas):
        st: State = ctx.st
        cfg, io = st.cfg, st.frame_io
        dev = io["renders"].device
        f32 = dict(dtype=torch.float32, device=dev)
        lib = L.lib()
        dims = cfg.dims()
        pi = dict(zip(_PROJ_IN, ctx.saved_tensors))
        dyn = cfg.G > 0
        arena = cfg.grad_arena or {}
        blended = io["blended"] is not None
        c32 = lambda t: None if t is None else t.to(torch.float32).contiguous()
        v_blended, v_acc, v_renders, v_alphas = c32(v_blended), c32(v_acc), c32(v_renders), c32(v_alphas)
        if not blended and v_renders is None:
            v_renders = torch.zeros_like(io["renders"])
        if blended and v_blended is None and v_acc is None and v_renders is None and v_alphas is None:
            v_blended = torch.zeros_like(io["blended"])

        def buf(name, *shape):
            t = arena.get(name)
            if t is not None and t.shape == torch.Size(shape) and t.dtype == torch.float32 and t.is_contiguous():
                return t.view(shape)
            return torch.empty(*shape, **f32)

        g = dict(
            v_means=buf("means", cfg.N, 3), v_quats=buf("quats", cfg.N, 4), v_scales=buf("scales", cfg.N, 3),
            v_opacities=buf("opacities", cfg.N), v_colors=buf("colors", cfg.N, cfg.D),
            v_motion_coefs=buf("motion_coefs", cfg.G, cfg.K) if dyn else None,
            v_rots=buf("rots", cfg.K, cfg.T, 6) if dyn else None, v_transls=buf("transls", cfg.K, cfg.T, 3) if dyn else None,
            v_times=buf("times", cfg.S) if dyn else None, v_RTs=buf("RTs", cfg.S, 3, 4) if pi["RTs"] is not None else None,
            v_viewmat=buf("viewmat", 4, 4), partials=None)
        v_m2d = torch.empty(cfg.S, cfg.N, 2, **f32)
        fg = L.FrameGrads()
        fg.v_blended, fg.v_acc, fg.v_renders, fg.v_alphas = L.ptr(v_blended), L.ptr(v_acc), L.ptr(v_renders), L.ptr(v_alphas)
        fg.v_means2d = L.ptr(v_m2d)
        fg.row_mode = row_mode_for(cfg, dev)
        if cfg.control_stats is not None:
            cs = _check_stats(cfg.control_stats, cfg.N)
            fg.stats_grad_norm_acc, fg.stats_vis_count = L.ptr(cs["xys_grad_norm_acc"]), L.ptr(cs["vis_count"])
            fg.stats_max_radii = L.ptr(cs["max_radii"])
            fg.stats_batch_size, fg.stats_update_max_radii = int(cs["batch_size"]), int(bool(cs.get("update_max_radii", False)))
        fio = L.fill(L.FrameIO(), **io)
        fio.near_target = cfg.near_target
        if blended:
            fio.policy = st.policy
        L.check(lib.d4gs_backward(C.byref(dims), C.byref(L.fill(L.ProjIn(), **pi)), C.byref(fio), C.byref(fg),
                                  C.byref(L.fill(L.LeafGrads(), **g)), C.c_void_p(st.ws_ptr), st.ws_bytes, st.ws_cap[0], st.ws_cap[1],
                                  _stream()), "d4gs_backward")
        st.v_means2d = v_m2d
        if st.xys_sink is not None:  # the `_current_xys[i].grad` side channel
            for s, x in enumerate(st.xys_sink):
                x.grad = v_m2d[s:s + 1]
        outs = [g["v_means"], g["v_quats"], g["v_scales"], g["v_opacities"], g["v_colors"], g["v_motion_coefs"],
                g["v_rots"], g["v_transls"], g["v_times"], g["v_RTs"], g["v_viewmat"]]
        outs = [x if need else None for x, need in zip(outs, ctx.needs)]
        return (None, None, *outs, None, None)


def frame_supported(cfg: RenderCfg) -> bool:
    """One kernel width, no channel chunking: what the one-call path covers."""
    return cfg.N > 0 and cfg.D in SUPPORTED_D


class PosesFn(torch.autograd.Function):
    """The pose API of the S2 seam (flow3d/scene_model.py:58-120) and the a11 track channels (:258-289) on the HIP path:
    d4gs_poses_fwd / d4gs_poses_bwd.  For the B times `ts` -> any of
      means  [N,B,3]   R_b means + t_b for the first G = len(motion_coefs) rows, the raw means for the static rest (then,
                       for the track channels, expressed in the cameras `w2cs34 [B,3,4]`),
      quats  [N,B,4]   wxyz, unit: normalize(rotmat_to_unitquat(R_b) (x) normalize(quats)),
      tfs    [G,B,3,4] MotionBases.compute_transforms(ts, softmax(motion_coefs)).
    `g_major` picks the memory layout of the reference's tensors ((G,B,...) contiguous); time-major otherwise."""

    @staticmethod
    def forward(ctx, means, quats, motion_coefs, rots, transls, ts, w2cs34, want: tuple, g_major: bool, raw_coefs: bool = True):
        _need_gpu(means)
        dev = means.device
        N, B = means.shape[0], ts.shape[0]
        G = 0 if motion_coefs is None else motion_coefs.shape[0]
        K, T = (rots.shape[0], rots.shape[1]) if G > 0 else (0, 0)
        want_m, want_q, want_t = want
        cfg = RenderCfg(N=N, G=G, K=K, T=T, S=B, D=1, width=16, height=16, flags=L.RAW_PARAMS if raw_coefs else 0,
                        exact_cull=False)
        pin = dict(means=_f32c(means), quats=_f32c(quats) if want_q else None, scales=None, opacities=None, colors=None,
                   motion_coefs=_f32c(motion_coefs), rots=_f32c(rots), transls=_f32c(transls), times=_f32c(ts),
                   RTs=_f32c(w2cs34), viewmat=None, Kmat=None)
        f32 = dict(dtype=torch.float32, device=dev)
        shp = (lambda n, *r: (n, B, *r)) if g_major else (lambda n, *r: (B, n, *r))
        out = dict(means=torch.empty(shp(N, 3), **f32) if want_m else None,
                   quats=torch.empty(shp(N, 4), **f32) if want_q else None,
                   transforms=torch.empty(shp(G, 3, 4), **f32) if want_t and G > 0 else None)
        po = L.fill(L.Poses(), **out)
        po.g_major = int(g_major)
        dims = cfg.dims()
        if any(v is not None for v in out.values()):
            L.check(L.lib().d4gs_poses_fwd(C.byref(dims), C.byref(L.fill(L.ProjIn(), **pin)), C.byref(po), _stream()),
                    "d4gs_poses_fwd")
        ctx.cfg, ctx.g_major, ctx.want = cfg, bool(g_major), want
        ctx.set_materialize_grads(False)  # unused outputs arrive as None (no zero-filled [N,B,...] gradients to stream)
        ctx.keys = [k for k, v in pin.items() if v is not None]
        ctx.save_for_backward(*[pin[k] for k in ctx.keys])  # version-checked, like ProjectFn
        ctx.needs = [t is not None and t.requires_grad for t in (means, quats, motion_coefs, rots, transls, ts, w2cs34)]
        if want_t and out["transforms"] is None:
            out["transforms"] = torch.zeros(shp(0, 3, 4), **f32)
        return out["means"], out["quats"], out["transforms"]

    @staticmethod
    def backward(ctx, v_means, v_quats, v_tfs):
        cfg = ctx.cfg
        pin = dict.fromkeys(_PROJ_IN)
        pin.update(zip(ctx.keys, ctx.saved_tensors))
        dev = pin["means"].device
        f32 = dict(dtype=torch.float32, device=dev)
        lib = L.lib()
        dims = cfg.dims()
        dyn = cfg.G > 0
        c = lambda t: None if t is None else t.to(torch.float32).contiguous()
        if not dyn:
            v_tfs = None
        # (keep the contiguous copies referenced until the call below has been enqueued: a temporary that dies earlier hands
        # its block back to the caching allocator, and the very next torch.empty - the gradient buffers - would alias it)
        vin = dict(means=c(v_means), quats=c(v_quats) if pin["quats"] is not None else None, transforms=c(v_tfs))
        vo = L.fill(L.Poses(), **vin)
        vo.g_major = int(ctx.g_major)
        g = dict(v_means=torch.empty(cfg.N, 3, **f32), v_quats=torch.empty(cfg.N, 4, **f32) if vo.quats else None,
                 v_scales=None, v_opacities=None, v_colors=None,
                 v_motion_coefs=torch.empty(cfg.G, cfg.K, **f32) if dyn else None,
                 v_rots=torch.empty(cfg.K, cfg.T, 6, **f32) if dyn else None,
                 v_transls=torch.empty(cfg.K, cfg.T, 3, **f32) if dyn else None,
                 v_times=torch.empty(cfg.S, **f32) if dyn else None,
                 v_RTs=torch.empty(cfg.S, 3, 4, **f32) if pin["RTs"] is not None else None,
                 v_viewmat=None,
                 partials=torch.empty(lib.d4gs_bwd_partials_elems(C.byref(dims)), **f32))
        L.check(lib.d4gs_poses_bwd(C.byref(dims), C.byref(L.fill(L.ProjIn(), **pin)), C.byref(vo),
                                   C.byref(L.fill(L.LeafGrads(), **g)), _stream()), "d4gs_poses_bwd")
        del vin
        outs = [g["v_means"], g["v_quats"], g["v_motion_coefs"], g["v_rots"], g["v_transls"], g["v_times"], g["v_RTs"]]
        return (*(x if need else None for x, need in zip(outs, ctx.needs)), None, None, None)


def poses(means, quats, motion_coefs, rots, transls, ts, want=(True, True, False), raw_coefs=True):
    """-> (means [N,B,3] | None, quats [N,B,4] | None, transforms [G,B,3,4] | None) in the reference's (G,B,...) layout
    (flow3d/scene_model.py:58-120); the first G = len(motion_coefs) Gaussians are dynamic.  raw_coefs=False: the
    coefficients are already activated (MotionBases.compute_transforms' own signature, params.py:142)."""
    return PosesFn.apply(means, quats, motion_coefs, rots, transls, ts, None, tuple(want), True, raw_coefs)


def track_points(means, motion_coefs, rots, transls, target_ts, target_w2cs=None):
    """-> [N,B,3] positions of all Gaussians (the first G = len(motion_coefs) deformed) at `target_ts`, expressed in
    the cameras `target_w2cs [B,4,4]` (world frame if None)."""
    w34 = None if target_w2cs is None else target_w2cs[:, :3, :]
    pts, _, _ = PosesFn.apply(means, None, motion_coefs, rots, transls, target_ts, w34, (True, False, False), False)
    return pts.permute(1, 0, 2)


def render_instances(cfg: RenderCfg, means, quats, scales, opacities, colors, motion_coefs, rots, transls, times, RTs,
                     viewmat, Kmat, background):
    """Deform + project + bin + sort + composite all S sub-samples.
    -> render_colors [S,H,W,D'], render_alphas [S,H,W,1], means2d [S,N,2], radii int32 [S,N], state."""
    drop0 = cfg.D == 0  # e.g. return_color=False with depth only (scene_model.py:221-225): carry one zero channel
    if drop0:
        colors = torch.zeros(cfg.N, 1, dtype=torch.float32, device=means.device)
        background = None if background is None else torch.zeros(1, dtype=torch.float32, device=means.device)
        cfg = replace(cfg, D=1, n_sigmoid=0, flags=cfg.flags & ~L.RAW_COLORS)
        rc, ra, m2d, radii, st = render_instances(cfg, means, quats, scales, opacities, colors, motion_coefs, rots,
                                                  transls, times, RTs, viewmat, Kmat, background)
        return rc[..., 1:], ra, m2d, radii, st
    resolve_lazy(cfg, means.device)
    st = State(cfg)
    if cfg.N == 0:  # an empty scene (e.g. everything culled): the image is the background, nothing to launch
        _need_gpu(means)
        dev, S, H, W = means.device, cfg.S, cfg.height, cfg.width
        rc = torch.zeros(S, H, W, cfg.NCH, device=dev)
        if background is not None:
            rc[..., :cfg.D] = background.to(torch.float32).reshape(-1)[:cfg.D]
        rc = rc + 0.0 * means.sum()  # keeps the autograd graph connected (all-zero gradients)
        st.n_isect, st.max_tile = 0, 0
        st.proj_out = dict(tile_offsets=torch.zeros(S * cfg.tiles[0] * cfg.tiles[1] + 1, dtype=torch.int32, device=dev),
                           depths=torch.zeros(S, 0, device=dev), conics=torch.zeros(S, 0, 3, device=dev),
                           opac_act=torch.zeros(0, device=dev), tiles_touched=torch.zeros(S * 0, dtype=torch.int32, device=dev))
        st.isect = dict(sorted_gid=torch.zeros(0, dtype=torch.int32, device=dev))
        st.raster = dict(last_ids=torch.zeros(S, H, W, dtype=torch.int32, device=dev))
        return (rc, torch.zeros(S, H, W, 1, device=dev), torch.zeros(S, 0, 2, device=dev) + 0.0 * means.sum(),
                torch.zeros(S, 0, dtype=torch.int32, device=dev), st)
    means2d, conics, depths, opac_act, ctab, radii = ProjectFn.apply(
        st, means, quats, scales, opacities, colors, motion_coefs, rots, transls, times, RTs, viewmat, Kmat)
    chunks = channel_chunks(cfg.D)
    if len(chunks) == 1 and chunks[0][2] == cfg.D:  # the common case: one kernel width, nothing to slice or pad
        rc, ra = RasterFn.apply(st, cfg, means2d, conics, depths, opac_act, ctab, background)
        return rc, ra, means2d, radii, st
    # any other channel count: chunks of <= 16 channels composited from the SAME projection + sorted tile lists
    # (gsplat renders wide feature vectors in `channel_chunk`-sized passes the same way); autograd sums the chunks'
    # per-instance gradients.  Only the first chunk's alpha is differentiated (they are all the same numbers).
    F = torch.nn.functional
    if cfg.control_stats is not None and means2d.requires_grad:
        # the statistics need the TOTAL means2d gradient (autograd sums the chunks): separate pass on that sum
        cs = _check_stats(cfg.control_stats, cfg.N)

        def _stats(gsum, cs=cs, radii=radii, cfg=cfg):
            with torch.no_grad():
                L.check(L.lib().d4gs_control_stats(cfg.S, cfg.N, L.ptr(gsum.to(torch.float32).contiguous()), L.ptr(radii),
                                                   cfg.width, cfg.height, int(cs["batch_size"]), L.ptr(cs["xys_grad_norm_acc"]),
                                                   L.ptr(cs["vis_count"]), L.ptr(cs["max_radii"]),
                                                   int(bool(cs.get("update_max_radii", False))), _stream()), "d4gs_control_stats")

        means2d.register_hook(_stats)
    outs, ra = [], None
    for k, (c0, c1, Dk) in enumerate(chunks):
        last = k == len(chunks) - 1
        ck = replace(cfg, D=Dk, depth_mode=cfg.depth_mode if last else L.DEPTH_NONE, grad_arena=None, control_stats=None)
        tab = F.pad(ctab[:, c0:c1], (0, ck.DP - (c1 - c0)))
        bgk = None if background is None else F.pad(background.to(torch.float32).reshape(-1)[c0:c1], (0, Dk - (c1 - c0)))
        rck, rak = RasterFn.apply(st, ck, means2d, conics, depths, opac_act, tab, bgk)
        if ra is None:
            ra = rak
        outs.append(rck[..., :c1 - c0])
        if last and cfg.depth_mode != L.DEPTH_NONE:
            outs.append(rck[..., Dk:Dk + 1])
    return torch.cat(outs, -1), ra, means2d, radii, st
