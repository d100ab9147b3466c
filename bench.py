#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on its config: "Gaussians/s fwd+bwd, 288x512, N_exposure=8".

A step = ONE blurry frame, forward + backward, through the one-call entry points d4gs_forward / d4gs_backward behind
one autograd node (`--staged`: the five staged entry points behind three nodes; same kernels): raw leaf params -> activations -> motion-basis deformation of all
S exposure sub-samples -> camera delta -> projection -> tile binning + per-tile depth sort -> composite ->
exposure blend -> loss = <blended, Wimg> + <acc, Wacc> -> gradients to every leaf (means, quats, scales,
opacities, colours, motion coefficients, bases, times, camera deltas, viewmat).  Inputs are resident in HBM
before the timed region.  value = Gaussians / t_frame (whole job); `instances_per_s` = N*S / t_frame.
The line also carries `roofline` (+ `roofline_hbm`, `roofline_streaming`) for the dominant kernel of WHATEVER `--config` was run - live
launch durations from HIP events, counters from the hash-stamped files under profiles/ - the measured ceilings of the box, `cpu_baseline`
(default config only), and `sustained`: the same step function run for `--sustain` seconds (default 6) right after the timed region,
reported beside `value`, never as it.  Eager steps verify their intersection-list sizes at most CHECK_LAG steps late and all of them
before the clock stops (`config.size_check`).

Configs (`--config`): cfg1 / cfg2 (headline, default) / cfg3 / cfg5 = BASELINE.json's, `refdefault[720]` = the
reference's own training shape (40 k dynamic + 100 k static Gaussians, 20 bases, 11 sub-samples, 17 channels:
run_training_dynamic.py:118-120, scene_model.py:233-296).  `--scale-mul F` multiplies every Gaussian's extent by F
(SURVEY 8d's distribution has sub-2-pixel splats; real scenes have larger footprints).

N GPUs (torchrun, one rank per GPU, RCCL): the default (`--shard auto`) is BASELINE config 4 in the strict sense at every N <= S -
`--shard exposure`: the S sub-samples of the SAME frame are split over the ranks, the blended frame is REDUCED (SUM all-reduce of colours + alpha, MAX all-reduce of
the max / min policy channels; HIP kernels d4gs_blend_shard_* around the collectives), the backward needs one MIN
all-reduce of the winning sub-sample, leaf gradients are all-reduced; the whole step, collectives included, is replayed
from ONE HIP graph (`--no-graph`: eager).  Total work is fixed -> "scaling": "strong", value = N / t, and the N = 1
value equals the single-GPU line.  The view-sharded (data-parallel) throughput - every rank renders its own full frame, "weak" scaling,
value = world * N / t - is measured right after the timed region and reported as the secondary object
`views_weak_scaling`, and from N = 4 the (N/2) views x 2-way exposure mesh as `views_x_exposure_mesh` (value = views * N / t, weak).
`--shard views` / `--shard mesh --mesh VxE` make one of them the primary line instead.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (N, G, K, S, W, H)
    "cfg1": (10_000, 0, 1, 1, 512, 288),
    "cfg2": (300_000, 300_000, 6, 8, 512, 288),
    "cfg3": (300_000, 300_000, 6, 8, 1280, 720),
    "cfg5": (1_000_000, 1_000_000, 12, 16, 1280, 720),
    "refdefault": (140_000, 40_000, 20, 11, 512, 288),
    "refdefault720": (140_000, 40_000, 20, 11, 1280, 720),
    "tiny": (20_000, 12_000, 4, 4, 256, 144),
}
SEEDS = {"cfg1": 1000, "cfg2": 1001, "cfg3": 1002, "cfg5": 1004, "refdefault": 1010, "refdefault720": 1011, "tiny": 1099}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
F32_PEAK_TFLOPS = 157.3  # fp32 vector peak == fp32-input dense MFMA peak (MI355X_MICROARCH.md)
N_CU, N_SIMD, N_XCD = 256, 1024, 8
# ALGORITHMIC FP32 ops per (splat, pixel) of the reference composite (gsplat rasterize_to_pixels fwd / bwd at 4
# channels), SURVEY.md section 8(d): ~30 forward, ~90 backward.  The HIP kernels execute fewer.
CHECK_LAG = 4  # eager steps whose deferred list-size checks may still be pending while the host queues the next one
FLOPS_PER_PAIR_BWD = 90.0
FLOPS_PER_PAIR_FWD = 30.0
FLOPS_INVALID_PAIR = 12.0  # what a pair that fails the alpha test needs at minimum: delta, sigma, exp, alpha, the test


def flops_per_pair(bwd: bool, n_ch: int) -> float:
    """SURVEY 8(d)'s ~30 / ~90 FP32 ops per (splat, pixel) are quoted at 4 composited channels (RGB + depth).  Per extra channel the
    reference composite adds 2 forward (out += c * fac) and 8 backward (v_c = fac * v_out; v_alpha += (c * T - buffer * ra) * v_out;
    buffer += c * fac): 22 + 2 C forward, 58 + 8 C backward - 30 / 90 at C = 4, 56 / 194 at the training shape's C = 17."""
    return (58.0 + 8.0 * n_ch) if bwd else (22.0 + 2.0 * n_ch)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--pre-roll", type=int, default=30,
                    help="untimed steps between the warm-up steps (+ the graph capture, if any) and the timed region (reported in config.pre_roll_steps): the first ~10 steps of a process run "
                         "3-7 %% slow (list-size guesses, row mode and lazy-sort decisions settle within 3; the device then needs ~15 ms of "
                         "load to reach its sustained state - profiles/r04t_bench_ramp.txt), so with a short --warmup the mean of K timed "
                         "steps is the start-up ramp, not the frame time a training run sees")
    ap.add_argument("--sustain", type=float, default=6.0,
                    help="N = 1: seconds of the same step run after the timed region (reported as `sustained`, never as `value`); 0 = off")
    ap.add_argument("--config", default="cfg2", choices=list(CONFIGS))
    ap.add_argument("--shard", default="auto", choices=["auto", "exposure", "views", "mesh"],
                    help="N > 1: `exposure` = BASELINE config 4 (the S sub-samples of ONE frame over all ranks; strong scaling), `views` = one "
                         "full frame of its own camera view per rank (data parallel; weak scaling), `mesh` = V views x E-way exposure "
                         "(--mesh VxE).  `auto` (default): exposure (the strict cfg4) at every N <= S; "
                         "the line's secondary objects carry the views-only and, from N = 4, the (N/2)x2 mesh numbers as well")
    ap.add_argument("--mesh", default=None, metavar="VxE", help="with --shard mesh: V views x E-way exposure sharding, V * E == --gpus")
    ap.add_argument("--graph-timeout", type=float, default=90.0,
                    help="N > 1: seconds each phase after the first (eager) measurement may take - the HIP-graph capture / replays with RCCL "
                         "collectives inside, the secondary measurements - before a watchdog prints the line measured so far and exits 0 "
                         "(the eager measurement itself gets 3x this; if IT hangs there is no line and the exit code is 3)")
    ap.add_argument("--channels", type=int, default=None, choices=[3, 16],
                    help="colour channels before depth: 3 = RGB+ED (headline), 16 = the reference's dynamic-training "
                         "shape (rgb + mask + 4x3 track channels + depth = 17, scene_model.py:233-296; default of "
                         "--config refdefault*)")
    ap.add_argument("--scale-mul", type=float, default=1.0, help="multiply every Gaussian's extent (footprint sensitivity)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-full", action="store_true",
                    help="BASELINE.md section 3's whole plan (cfg1 x 20 iterations, cfg2 x 3, torch and scalar C); minutes")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--sync-size-check", action="store_true",
                    help="every render waits on the host for its intersection counts before the backward is issued (the "
                         "library's default); without the flag the counts are verified once per step behind the launches "
                         "(`deferred_size_check` + `engine.check_deferred()`, as examples/train_dynamic_step.py does), so "
                         "the host may run a frame ahead and a slow host does not stall the GPU")
    ap.add_argument("--staged", action="store_true",
                    help="drive the five staged C entry points through three autograd nodes (ProjectFn / RasterFn / BlendFn) "
                         "instead of the one-call d4gs_forward / d4gs_backward node (same kernels, more host work)")
    ap.add_argument("--no-graph", action="store_true",
                    help="with --gpus N > 1: time the eager step (host-bound: ~40 launches + 3 collectives per step) "
                         "instead of the captured one")
    ap.add_argument("--graph", action="store_true",
                    help="capture one step (render forward + backward, deferred size check) in a HIP graph and time its "
                         "replays (with --gpus N > 1 / --force-dist the RCCL collectives are captured too: verified at world size 1 only)")
    ap.add_argument("--force-dist", action="store_true", help="exercise the RCCL code path with world_size 1")
    ap.add_argument("--dry-run", action="store_true",
                    help="run the N > 1 control flow of this script on CPU tensors over gloo (tests/test_parallel_gloo.py): "
                         "process-group set-up, the sharded step's collectives, the graph-capture fallback agreement, max-over-ranks "
                         "timing, ONE JSON line - with the render replaced by a synthetic differentiable image.  Measures nothing.")
    ap.add_argument("--lazy-sort", action="store_true",
                    help="secondary measurement: D4GS_LAZY_SORT (near / far partition of the tile lists, far parts sorted only for tiles "
                         "that did not saturate in the near part; for occluded / large-footprint scenes, e.g. --scale-mul 4)")
    ap.add_argument("--no-peaks", action="store_true", help="skip d4gs_measure_peaks (profiler passes: keeps its kernels out of the trace)")
    ap.add_argument("--spatial-order", nargs="?", const="view", default=None, choices=["view", "3d"],
                    help="secondary measurement: the same Gaussians in the order deblur4dgs_amd.control.spatial_order_step leaves them in "
                         "(Morton curve over the camera's image plane, per set; `3d`: over world space) instead of the generator's random "
                         "order; the headline keeps the random order")
    ap.add_argument("--share", type=int, default=1, metavar="P",
                    help="diagnostic (not the headline): render only rank 0's share {s : s %% P == 0} of the exposure sub-samples, no "
                         "collectives - the device work ONE rank of an exposure-sharded frame at world size P executes "
                         "(scripts/shard_floor.py; lets the profiling scripts look at the S / P kernels on one GPU)")
    return ap.parse_args()


def to_dev(sc, dev):
    return {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in sc.items()}


LEAVES = ("means", "quats", "scales", "opacities", "colors", "motion_coefs", "rots", "transls", "times", "RTs", "viewmat")


def scene_of(name, seed_offset=0, channels=3, scale_mul=1.0):
    import math

    from deblur4dgs_amd.synth import make_scene

    N, G, K, S, W, H = CONFIGS[name]
    sc = make_scene(N, G, K, S, W, H, seed=SEEDS[name] + seed_offset, D=channels,
                    cam_jitter=0.0 if name == "cfg1" else 0.002)  # identity camera delta for cfg1
    if scale_mul != 1.0:
        sc["scales"] = sc["scales"] + math.log(scale_mul)
    return sc


def make_inputs(name, dev, seed_offset=0, channels=3, scale_mul=1.0):
    N, G, K, S, W, H = CONFIGS[name]
    sc = scene_of(name, seed_offset, channels, scale_mul)
    d = to_dev(sc, dev)
    leaves = {k: d[k].clone().requires_grad_() for k in LEAVES if k in d and (G > 0 or k not in ("times",))}
    g = torch.Generator().manual_seed(7)
    wimg = torch.randn(H, W, channels + 1, generator=g).to(dev)
    wacc = torch.randn(H, W, generator=g).to(dev)
    return sc, d, leaves, wimg, wacc


# ------------------------------------------------------------------------------------------------ CPU baseline
def _cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _cpu_frame_scalar_c(sc, name, threads):
    """One full blurry frame, forward + backward, through the scalar-C restatement (oracle/raster_ref.c) - the S
    sub-samples in parallel threads (ctypes releases the GIL), deformation by the torch restatement.  -> seconds."""
    import concurrent.futures as cf

    import numpy as np

    from oracle import cref, deform

    N, G, K, S, W, H = CONFIGS[name]
    D = sc["colors"].shape[-1]
    rng = np.random.default_rng(0)
    w_out = rng.standard_normal((H, W, D + 1)).astype(np.float32)
    w_al = rng.standard_normal((H, W, 1)).astype(np.float32)
    t0 = time.perf_counter()
    with torch.no_grad():
        scales, opac, cols = torch.exp(sc["scales"]), torch.sigmoid(sc["opacities"]), torch.sigmoid(sc["colors"])
        quat_static = torch.nn.functional.normalize(sc["quats"], dim=-1)

    def sub(s):
        with torch.no_grad():
            if G > 0:
                fg = {k: sc[k][:G] for k in ("means", "quats")}
                m, q = deform.compute_poses_fg(sc["times"][s:s + 1], fg["means"], fg["quats"], sc["motion_coefs"], sc["rots"],
                                               sc["transls"])
                m, q = m[:, 0], q[:, 0]
                if G < N:
                    m, q = torch.cat([m, sc["means"][G:]], 0), torch.cat([q, quat_static[G:]], 0)
            else:
                m, q = sc["means"], quat_static
            m = deform.camera_delta(m, sc["RTs"][s])
        out, al, ctx = cref.rasterization(m.numpy(), q.numpy(), scales.numpy(), opac.numpy(), cols.numpy(),
                                          sc["viewmat"].numpy(), sc["K"].numpy(), W, H, background=np.ones(D, np.float32),
                                          render_mode="RGB+ED", dtype=np.float32)
        cref.backward(ctx, w_out, w_al)
        return ctx["n_isect"]

    with cf.ThreadPoolExecutor(max_workers=threads) as ex:
        n_isect = sum(ex.map(sub, range(S)))
    return time.perf_counter() - t0, n_isect


TORCH_CPU_THREADS = 16  # measured on the GPU box's 256-thread EPYC 9575F host: with set_num_threads(256) one cfg1 frame
#                         of the torch restatement takes 205-831 s (thousands of tiny ops, thread oversubscription) against
#                         ~3 s on 8-16 threads - so the torch leg pins 16 threads and says so


def _cpu_frame_torch(sc, name):
    """The same frame through the vectorised torch restatement (oracle/scene.py), fp32, TORCH_CPU_THREADS threads."""
    from oracle import scene as oscene

    torch.set_num_threads(min(TORCH_CPU_THREADS, os.cpu_count()))
    N, G, K, S, W, H = CONFIGS[name]
    keys = ("means", "quats", "scales", "colors", "opacities")
    fg = bg = bases = None
    if G > 0:
        fg = {k: sc[k][:G].clone().requires_grad_() for k in keys}
        fg["motion_coefs"] = sc["motion_coefs"].clone().requires_grad_()
        bases = {k: sc[k].clone().requires_grad_() for k in ("rots", "transls")}
    if G < N:
        bg = {k: sc[k][G:].clone().requires_grad_() for k in keys}
    t0 = time.perf_counter()
    ref = oscene.render_exposure(fg, bg, bases, sc["times"], sc["RTs"], sc["viewmat"], sc["K"], (W, H), bg_color=1.0,
                                 return_depth=True, single=(S == 1))
    (ref["img"].sum() + ref["depth"].sum() + ref["acc"].sum()).backward()
    return time.perf_counter() - t0


def _stats(ts, N, S):
    return {"iters": len(ts), "s_per_frame_median": statistics.median(ts), "s_per_frame_min": min(ts),
            "s_per_frame_max": max(ts), "gaussians_per_s": N / statistics.median(ts),
            "instances_per_s": N * S / statistics.median(ts)}


def _cpu_twin(cfgname="cfg1", iters=5):
    """A BASELINE config through the PRODUCT's CPU twin (d4gs_forward_cpu / d4gs_backward_cpu, scalar fp32, one thread) -
    reported beside the oracle's legs; it is product code, so it is not the checker and not `value`."""
    from deblur4dgs_amd.cpu_twin import render_exposure_cpu

    N, G, K, S, W, H = CONFIGS[cfgname]
    sc = scene_of(cfgname, channels=3)
    keys = ("means", "quats", "scales", "colors", "opacities") + (("motion_coefs", "rots", "transls") if G > 0 else ())
    P = {k: sc[k].float().clone().requires_grad_() for k in keys}
    ts = []
    for _ in range(iters):
        for v in P.values():
            v.grad = None
        t0 = time.perf_counter()
        r = render_exposure_cpu(P["means"], P["quats"], P["scales"], P["opacities"], P["colors"], 3, P.get("motion_coefs"), P.get("rots"),
                                P.get("transls"), sc["times"].float() if G > 0 else None,
                                sc["RTs"].float(), sc["viewmat"].float(), sc["K"].float(), W, H, background=torch.ones(3),
                                return_depth=True)
        (r["blended"].sum() + r["acc"].sum()).backward()
        ts.append(time.perf_counter() - t0)
    out = _stats(ts, N, S)
    out.update(cores=1, entry="d4gs_forward_cpu + d4gs_backward_cpu (csrc/cpu_twin.hip)", config=cfgname)
    return out


def cpu_baseline(name, channels, full=False):
    """The build's CPU restatement of the path (the reference has NO CPU path: flow3d/scene_model.py:36,360), timed on
    this box's host cores as BASELINE.md section 3 plans: the same seeded frame, forward + backward, (i) through the
    scalar-C restatement with the S sub-samples on parallel threads and (ii) through the all-thread torch restatement;
    cfg1 in full, the headline config on a bounded number of iterations (default run: ~30 s of CPU work; the whole plan
    with --cpu-baseline-full, committed under profiles/)."""
    host = os.cpu_count()
    out = {"unit": "Gaussians/s", "kind": "port", "cpu_model": _cpu_model(), "host_cores": host, "runs": {}}
    # default run (bounded to well under a minute of CPU work): scalar C on cfg1 x 20 + the benched config
    # x 3, ONE frame of the torch restatement on cfg1 (16 threads; a cfg2 frame of it is minutes and stays in the full plan, committed
    # under profiles/), and the product's own CPU twin on cfg1 x 5 and on cfg2 x 1
    plan = [("cfg1", "scalar_c", 20), ("cfg1", "torch", 20 if full else 1)]
    if name != "cfg1":
        plan += [(name, "scalar_c", 3 if full or name in ("cfg2", "tiny", "refdefault") else 1)]
        if full:
            plan += [(name, "torch", 3 if name == "cfg2" else 1)]
    for cfgname, impl, iters in plan:
        N, G, K, S, W, H = CONFIGS[cfgname]
        sc = scene_of(cfgname, channels=channels if cfgname == name else 3)
        threads = min(S, host)
        ts, n_isect = [], None
        for _ in range(iters):
            if impl == "scalar_c":
                t, n_isect = _cpu_frame_scalar_c(sc, cfgname, threads)
            else:
                t = _cpu_frame_torch(sc, cfgname)
            ts.append(t)
        r = _stats(ts, N, S)
        r["cores"] = threads if impl == "scalar_c" else min(TORCH_CPU_THREADS, host)
        if n_isect is not None:
            r["n_isect"] = n_isect
        out["runs"][f"{cfgname}/{impl}"] = r
    out["product_cpu_twin"] = _cpu_twin("cfg1", 5)
    if name == "cfg2":
        out["product_cpu_twin_cfg2"] = _cpu_twin("cfg2", 1)
    head = out["runs"][f"{name}/scalar_c"]
    N, G, K, S, W, H = CONFIGS[name]
    out.update(value=head["gaussians_per_s"], cores=head["cores"],
               sample=f"{head['iters']} full frame(s) of {name} ({N} Gaussians, {W}x{H}, S={S}), forward + backward through "
                      f"oracle/raster_ref.c (scalar C, fp32) with the {S} sub-samples on {head['cores']} parallel threads + torch "
                      f"deformation; median {head['s_per_frame_median']:.2f} s per frame (min {head['s_per_frame_min']:.2f}, max "
                      f"{head['s_per_frame_max']:.2f}); `runs` also holds cfg1 in full (20 iterations of scalar C) and one cfg1 frame of the torch "
                      f"restatement on {min(TORCH_CPU_THREADS, host)} threads; `product_cpu_twin[_cfg2]` = the product's own scalar CPU entry points; the "
                      f"torch restatement on cfg2 (minutes per frame) runs with --cpu-baseline-full (profiles/r03_cpu_baseline_full.json)")
    return out


# ------------------------------------------------------------------------------------------------ roofline helpers
def _load_json(*rel):
    try:
        return json.load(open(os.path.join(ROOT, *rel)))
    except Exception:
        return None


_LIB_SHA = None


def _lib_sha():
    """sha256 of the libd4gs.so this run loads: the committed counter files (profiles/pmc_traffic.json,
    profiles/<round>_pmc_sq_cfg2.json) carry the hash of the build they were measured on and are only quoted while it
    matches - a stale profile must not dress up a changed kernel."""
    global _LIB_SHA
    if _LIB_SHA is None:
        import hashlib

        from deblur4dgs_amd import _lib as L

        _LIB_SHA = hashlib.sha256(open(L.LIB_PATH, "rb").read()).hexdigest()
    return _LIB_SHA


def _counters_current(doc):
    return bool(doc) and doc.get("lib_sha256") == _lib_sha()


def _traffic(name, kernel):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (separate --pmc FETCH_SIZE / WRITE_SIZE runs, KB
    units; MI355X_MICROARCH.md: on gfx950 FETCH_SIZE under-counts wide coalesced reads up to 2x, so the read term is
    reported both ways)."""
    doc = _load_json("profiles", "pmc_traffic.json")
    if not _counters_current(doc):
        return None  # measured on another build of the kernels
    pm = doc.get(name, {}).get("kernels", {}).get(kernel)
    if not pm:
        return None
    f, w = pm["FETCH_SIZE"] * 1024.0, pm["WRITE_SIZE"] * 1024.0
    return {"fetch_1x": f, "fetch_2x": 2 * f, "write": w, "total_1x": f + w, "total_2x": 2 * f + w}


class Watchdog:
    """N > 1: the eager sharded step is measured FIRST; everything after it (the HIP-graph capture with RCCL collectives inside,
    its replays, the secondary measurements) runs under this watchdog.  If a phase does not finish in time - a collective that hangs
    in a capture is the realistic first-contact failure at N = 8, and RCCL's own watchdog takes 600 s and prints no line - rank 0
    prints the line measured so far (`config.launch` says what happened) and every rank leaves with exit code 0.  A daemon thread:
    torch's collectives, synchronize(), graph replay and .item() all release the GIL while they wait; `faulthandler` (a C thread that
    needs no GIL) is the last resort behind it."""

    def __init__(self, rank: int, timeout: float):
        import threading

        self.rank, self.timeout = rank, timeout
        self.out = None  # the best line so far (rank 0)
        self.phase = ""
        self.deadline = None
        self._lock = threading.Lock()
        t = threading.Thread(target=self._run, daemon=True)
        t.start()

    def kick(self, phase: str, out=None, scale: float = 1.0):
        """A phase starts: (re)arm the deadline (`scale` x the timeout); `out` replaces the line a firing watchdog prints."""
        import faulthandler

        with self._lock:
            self.phase = phase
            if out is not None:
                self.out = json.loads(json.dumps(out))  # a private copy: the main thread keeps editing its own
            self.deadline = time.monotonic() + scale * self.timeout
        faulthandler.dump_traceback_later(scale * self.timeout + 30.0, exit=True)

    def done(self):
        import faulthandler

        with self._lock:
            self.deadline = None
        faulthandler.cancel_dump_traceback_later()

    def _run(self):
        while True:
            time.sleep(0.25)
            with self._lock:
                fire = self.deadline is not None and time.monotonic() > self.deadline
                out, phase = self.out, self.phase
            if fire:
                if self.rank == 0 and out is not None:
                    out.setdefault("config", {})["launch"] = (f"WATCHDOG: phase '{phase}' did not finish within {self.timeout:g} s (hung collective / "
                                                              "capture?); this is the line measured before it (eager step)")
                    out["watchdog_fired_in"] = phase
                    os.write(1, (json.dumps(out) + "\n").encode())
                sys.stderr.write(f"[bench.py rank {self.rank}] watchdog: phase '{phase}' timed out; leaving with "
                                 + ("the line measured so far\n" if out is not None else "NO line: nothing had been measured yet\n"))
                sys.stderr.flush()
                os._exit(0 if out is not None else 3)


def resolve_shard(args, world):
    """-> (mode, (V, E)) of the primary measurement."""
    if world == 1:
        return ("exposure" if args.shard in ("auto", "mesh") else args.shard), (1, 1)
    if args.shard == "exposure":
        return "exposure", (1, world)
    if args.shard == "views":
        return "views", (world, 1)
    if args.shard == "mesh":
        if not args.mesh:
            raise SystemExit("--shard mesh needs --mesh VxE")
        V, E = (int(x) for x in args.mesh.lower().split("x"))
        if V * E != world:
            raise SystemExit(f"--mesh {args.mesh}: V * E must equal --gpus ({world})")
        return "mesh", (V, E)
    # auto = BASELINE config 4 in the strict sense at every N (ADVICE r5 / VERDICT r5 #9: the primary number must mean the same
    # thing at N = 2 and N = 8 and be the one `configs[3]` describes); the views x exposure mesh and the views-only numbers ride
    # on the same line as secondary objects.  More ranks than sub-samples: the widest exposure sharding that divides the world.
    S = CONFIGS["tiny" if args.dry_run else args.config][3]
    if world <= S:
        return "exposure", (1, world)
    E = max(e for e in range(1, S + 1) if world % e == 0)
    return ("mesh", (world // E, E)) if E > 1 else ("views", (world, 1))


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    dry = args.dry_run
    if not dry:
        torch.cuda.set_device(local)
    dev = torch.device("cpu") if dry else torch.device("cuda", local)
    use_dist = world > 1 or args.force_dist
    # N > 1: the sharded step leaves each rank a fraction of a millisecond of device work behind ~1 ms of host launch work,
    # so the captured step (one hipGraphLaunch, RCCL collectives inside) is the default there; --no-graph times the eager one.
    # The EAGER step is measured first and its line kept; the capture and everything after it run under a watchdog (class Watchdog).
    graph_default = world > 1 and not args.graph and not args.no_graph
    want_graph = args.graph or graph_default
    eager_first = world > 1 and want_graph
    primary_mode, primary_mesh = resolve_shard(args, world)
    if use_dist:
        import torch.distributed as dist

        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the pool's driver only supports dmabuf IPC
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if dry:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)

    from deblur4dgs_amd import _lib as L
    from deblur4dgs_amd import engine
    from deblur4dgs_amd.exposure import render_exposure
    from deblur4dgs_amd.parallel import ShardedExposure

    name = "tiny" if dry else args.config
    if os.environ.get("D4GS_BENCH_S"):  # experiment hook (scripts/_g*.sh): the same scene with another number of exposure sub-samples
        CONFIGS[name] = CONFIGS[name][:3] + (int(os.environ["D4GS_BENCH_S"]),) + CONFIGS[name][4:]
    N, G, K, S, W, H = CONFIGS[name]
    channels = args.channels or (16 if name.startswith("refdefault") else 3)
    if use_dist and primary_mesh[1] > S:
        raise SystemExit(f"exposure sharding needs its width <= S ({primary_mesh[1]} > {S})")
    bg = torch.ones(channels, device=dev)
    lib = None if dry else L.lib()
    prof_ok = not args.no_profile and not dry
    _groups = {}

    def exposure_group(V, E):
        """The exposure sub-group of this rank's view in a V x E mesh (dist.new_group is collective: made once per mesh, by every rank)."""
        if E == 1 or V == 1:
            return None  # E == 1: no blend collective; V == 1: the world group
        if (V, E) not in _groups:
            from deblur4dgs_amd.parallel import mesh_exposure_group

            _groups[(V, E)] = mesh_exposure_group(world, rank, V, E)
        return _groups[(V, E)]

    def sync():
        if use_dist:
            import torch.distributed as dist

            dist.barrier()
        if not dry:
            torch.cuda.synchronize()

    def dry_render(means, quats, scales, opacities, colors, n_sig, motion_coefs, rots, transls, times, RTs, viewmat, Kmat, Wd, Hd,
                   background=None, blend=True, **_kw):
        """--dry-run: a differentiable stand-in for render_exposure on CPU tensors - [S_loc, H, W, 4] images that depend on every
        leaf, so the blend collectives and the gradient all-reduce carry real (if meaningless) data."""
        import types

        g = torch.Generator().manual_seed(11)
        base = torch.rand(Hd, Wd, 4, generator=g)
        per_g = sum(t.mean() for t in (means, quats, scales, opacities, colors, motion_coefs))
        shared = rots.mean() + transls.mean() + viewmat.mean()
        gain = torch.tanh(times + RTs.reshape(RTs.shape[0], -1).sum(-1))  # [S_loc]
        renders = base[None] * (1.0 + 0.1 * gain.view(-1, 1, 1, 1)) + 0.01 * (per_g + shared)
        alphas = torch.sigmoid(renders[..., :1].detach() * 0 + gain.view(-1, 1, 1, 1) + per_g)
        st = types.SimpleNamespace(n_isect=0, cfg=types.SimpleNamespace(S=times.shape[0]))
        return dict(renders=renders, alphas=alphas, state=st, blended=renders.mean(0) if blend else None,
                    acc=alphas.mean(0)[..., 0] if blend else None)

    def collect():
        lib.d4gs_profile_enable(0)
        buf = C.create_string_buffer(1 << 16)
        lib.d4gs_profile_collect(buf, C.c_size_t(len(buf)))
        got = {}
        for line in buf.value.decode().splitlines():
            nm, cnt, ms = line.split()
            got[nm] = (int(cnt), float(ms))
        return got

    def measured_peaks():
        """SURVEY 8d / BASELINE.md section 4: the ceilings are MEASURED on this box, in the same process as the timed region (after it) - a ~1 GiB
        device-to-device stream copy and an FMA issue loop (d4gs_measure_peaks, csrc/peaks.hip; ~30 ms of device time)."""
        try:
            scratch = torch.empty(1 << 31, dtype=torch.uint8, device=dev)  # 2 x 1 GiB: far beyond the 256 MB MALL
            scratch.zero_()
            out = (C.c_double * 4)()
            rc = lib.d4gs_measure_peaks(C.c_void_p(scratch.data_ptr()), C.c_size_t(scratch.numel()), out,
                                        C.c_void_p(torch.cuda.current_stream().cuda_stream))
            torch.cuda.synchronize()
            del scratch
            torch.cuda.empty_cache()
            if rc != 0:
                return None
            return {"hbm_stream_copy_gbs": out[0], "fp32_pk_fma_tflops": out[1], "fp32_fma_tflops": out[2],
                    "copy_bytes_per_launch": out[3],
                    "how": "d4gs_measure_peaks on this GPU right after the timed region: device-to-device float4 stream copy of 1 GiB "
                           "(non-temporal, 8 loads in flight per lane, 256 workgroups per CU; read + write bytes / best of 8 launches), v_pk_fma_f32 and v_fma_f32 issue loops at 8 waves per SIMD "
                           "(16 independent chains per lane, best of 5)"}
        except Exception as e:  # the ceilings are context, not the measurement: never take the bench line down
            sys.stderr.write(f"d4gs_measure_peaks failed: {e!r}\n")
            return None


    def measure(mode, mesh, steps, warmup, profile, use_graph):
        """One measurement of `mode` (mesh = (V, E)): warm-up, optional HIP-graph capture, pre-roll, the timed region.
        -> dict(dt = seconds for `steps` steps (max over ranks), kern / kern_all / n_break = live kernel timings, st = last state,
                graph_note, step_stats)"""
        graph_note, step_stats = {}, {}
        if use_graph:
            profile = False  # HIP events cannot bracket kernels inside a replayed graph
        V, E = mesh
        view = (rank // E) if use_dist else 0  # ranks of one view render the same scene
        sc, d, leaves, wimg, wacc = make_inputs(name, dev, seed_offset=view if V > 1 else 0, channels=channels,
                                                scale_mul=args.scale_mul)
        if args.spatial_order:  # the product's control.spatial_order_step, applied to the synthetic scene (all Gaussians dynamic or all
            from deblur4dgs_amd.control import morton_permutation  # static in the BASELINE configs; refdefault: per set)

            with torch.no_grad():
                m = leaves["means"].detach()
                vm = None if args.spatial_order == "3d" else leaves["viewmat"].detach()
                perm = torch.cat([morton_permutation(m[:G], vm), G + morton_permutation(m[G:], vm)]) if 0 < G < N else morton_permutation(m, vm)
                for k in ("means", "quats", "scales", "opacities", "colors", "motion_coefs"):
                    if k in leaves:
                        pk = perm if leaves[k].shape[0] == N else morton_permutation(m[:G], vm)
                        leaves[k] = leaves[k].detach()[pk].clone().requires_grad_()
        if args.share > 1 and not use_dist:
            for k in ("times", "RTs"):
                if k in leaves:
                    leaves[k] = leaves[k][::args.share].detach().clone().requires_grad_()
        sharder = None
        if use_dist:
            sharder = (ShardedExposure(world, rank, mode="mesh", mesh=(V, E), exposure_group=exposure_group(V, E)) if mode == "mesh"
                       else ShardedExposure(world, rank, mode=mode))
        if sharder is not None:
            sharder.fused = not args.staged
            if dry:
                sharder.render = dry_render
        deferred = not args.sync_size_check
        if sharder is not None:
            sharder.deferred_size_check = deferred
        last = {}
        mode_flag = {"deferred": deferred, "fused": not args.staged}

        def step():
            for v in leaves.values():
                v.grad = None
            if sharder is None:
                res = render_exposure(leaves["means"], leaves["quats"], leaves["scales"], leaves["opacities"],
                                      leaves["colors"], 3, leaves.get("motion_coefs"), leaves.get("rots"),
                                      leaves.get("transls"), leaves.get("times"), leaves["RTs"], leaves["viewmat"], d["K"],
                                      W, H, background=bg, return_depth=True, deferred_size_check=mode_flag["deferred"], fused=mode_flag["fused"],
                                      lazy_sort=True if args.lazy_sort else None)
                loss = torch.dot(res["blended"].reshape(-1), wimg.reshape(-1)) + torch.dot(res["acc"].reshape(-1), wacc.reshape(-1))
                loss.backward()
                last["st"] = res["state"]
            else:
                last["st"] = sharder.step(leaves, d["K"], W, H, bg, wimg, wacc)
            if mode_flag["deferred"] and not dry:
                # every step's list sizes are verified (an overflow raises) - at most CHECK_LAG steps late, and all of them before the
                # clock stops (round 6; rounds 4-5 waited for THIS step's forward here, which kept the host less than one step ahead of
                # the device: two of three bench runs of one session caught a 4-ms host stall as +8 ... +12 % of a 20-step mean,
                # profiles/r06_bench_host_stall.txt, at unchanged kernel times and an unchanged sustained rate)
                engine.check_deferred(keep_last=CHECK_LAG)

        eager_step = step
        for _ in range(warmup):
            step()
        sync()
        if use_graph:  # same kernels, same arithmetic; one hipGraphLaunch per step.  The sharded step is captured with its
            # RCCL collectives (verified at world size 1 only: tests/test_gpu_parallel.py - the pool has 1-GPU boxes).
            mode_flag["deferred"] = False  # one host-checked eager step: its state carries the exact list sizes
            if sharder is not None:
                sharder.deferred_size_check = False
            step()
            sync()
            real_state = last["st"]

            def grad_signature():  # one number per leaf: what a replay of the captured step must reproduce
                return torch.stack([v.grad.detach().double().abs().sum() if v.grad is not None else torch.zeros((), dtype=torch.float64, device=dev)
                                    for v in leaves.values()])

            ref_sig = grad_signature()
            mode_flag["deferred"] = deferred
            if not dry:
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())

            def gstep():
                if sharder is not None:
                    sharder.deferred_size_check = True
                    return sharder.step(leaves, d["K"], W, H, bg, wimg, wacc)
                for v in leaves.values():
                    v.grad = None
                res = render_exposure(leaves["means"], leaves["quats"], leaves["scales"], leaves["opacities"],
                                      leaves["colors"], 3, leaves.get("motion_coefs"), leaves.get("rots"),
                                      leaves.get("transls"), leaves.get("times"), leaves["RTs"], leaves["viewmat"], d["K"],
                                      W, H, background=bg, return_depth=True, deferred_size_check=True, fused=mode_flag["fused"],
                                      lazy_sort=True if args.lazy_sort else None)
                loss = torch.dot(res["blended"].reshape(-1), wimg.reshape(-1)) + torch.dot(res["acc"].reshape(-1), wacc.reshape(-1))
                loss.backward()
                return res["state"]

            if not dry:
                with torch.cuda.stream(side):
                    for _ in range(2):
                        gstep()
                torch.cuda.current_stream().wait_stream(side)
            if sharder is None:
                for v in leaves.values():
                    v.grad = None
            try:
                if dry:  # (rank 0 of a dry run "cannot capture": the agreement below must send EVERY rank to the eager step)
                    if os.environ.get("D4GS_BENCH_INJECT_HANG") == "graph" and rank == world - 1:
                        time.sleep(1e6)  # tests/test_parallel_gloo.py: a rank that never comes back from its capture
                    if rank == 0:
                        raise RuntimeError("no HIP graphs in a CPU dry run")
                    captured = True
                else:
                    graph = torch.cuda.CUDAGraph()
                    watch = engine.GraphWatch()  # the replays keep reporting their list sizes (an outgrown capture raises)
                    with watch.capturing(), torch.cuda.graph(graph):
                        gstep()
                    captured = True
            except Exception as e:  # (never seen at world size 1; an N > 1 capture has not run on this pool's 1-GPU boxes)
                captured = False
                graph_note["fallback"] = f"graph capture failed ({type(e).__name__}: {e}); timed the eager step"
                sys.stderr.write(graph_note["fallback"] + "\n")
                if not dry:
                    torch.cuda.synchronize()
            if use_dist:  # every rank must take the same path: one that could not capture sends them all to the eager step
                import torch.distributed as dist

                ok = torch.tensor([1 if captured else 0], device=dev, dtype=torch.int32)
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                if captured and int(ok.item()) == 0:
                    captured = False
                    graph_note["fallback"] = "another rank could not capture its step; timed the eager step"
            if captured:
                assert not dry, "a dry run never replays a graph: the cross-rank agreement failed"

                def step():  # noqa: F811
                    graph.replay()
                    last["st"] = real_state  # exact list sizes of the same scene (the captured state holds capacities)

                for _ in range(3):
                    step()
                watch.replayed()
                sync()
                watch.check()
                # the captured step - RCCL collectives included at N > 1 - must reproduce the eager step's gradients before
                # its time may stand for the frame; otherwise every rank goes back to the eager step
                got_sig = grad_signature()
                same = bool(((got_sig - ref_sig).abs() <= 1e-4 * ref_sig.abs().clamp(min=1e-30)).all().item())
                if use_dist:
                    ok = torch.tensor([1 if same else 0], device=dev, dtype=torch.int32)
                    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                    same = bool(int(ok.item()))
                if not same:
                    captured = False
                    graph_note["fallback"] = ("the captured step did not reproduce the eager step's gradients (leaf |grad| sums off by "
                                              f"up to {float(((got_sig - ref_sig).abs() / ref_sig.abs().clamp(min=1e-30)).max()):.2e}); timed the eager step")
                    sys.stderr.write(graph_note["fallback"] + "\n")
                    step = eager_step
            if not captured and sharder is not None:
                sharder.deferred_size_check = deferred
        # pre-roll: untimed steps of the FINAL step function (eager, or the replay of the graph captured above - capturing leaves the
        # device idle for a while) right in front of the timed region, so that it starts from the sustained state (--pre-roll)
        n_pre = 0 if dry else args.pre_roll
        for i in range(n_pre):
            if profile and i == max(n_pre - 2, 0):
                lib.d4gs_profile_enable(3)  # the event machinery's first use (a one-off of up to 1.4 ms) happens here, not in the timed region
            step()
        if n_pre:
            sync()
            if profile:
                collect()  # (discarded: the timed region starts from empty records)
        if profile:
            lib.d4gs_profile_enable(3)  # HIP events around the roofline's kernel - the composite backward - only: timing EVERY
        t0 = time.perf_counter()      # kernel costs two stream events per launch, ~0.12 ms of a 1.6 ms frame
        marks = []
        for _ in range(steps):
            step()
            marks.append(time.perf_counter())  # (host time a step's calls returned: says WHERE a slow run lost its time)
        sync()
        if not dry and mode_flag["deferred"]:
            engine.check_deferred()  # the last CHECK_LAG steps' list sizes: everything has landed, nothing to wait for
        dt = time.perf_counter() - t0
        host_gaps = [b - a for a, b in zip([t0] + marks[:-1], marks)]
        step_stats["host_ms_median"] = 1e3 * sorted(host_gaps)[len(host_gaps) // 2]
        step_stats["host_ms_max"] = 1e3 * max(host_gaps)
        step_stats["host_ms_max_at"] = host_gaps.index(max(host_gaps))
        step_stats["slowest"] = [(i, round(1e3 * g, 4)) for g, i in sorted(((g, i) for i, g in enumerate(host_gaps)), reverse=True)[:5]]
        step_stats["last_sync_ms"] = 1e3 * (t0 + dt - marks[-1])
        kern, kern_all, n_break = {}, {}, 0
        if profile:
            kern = collect()  # the dominant kernels, measured live over the timed region
        # sustained cross-check (N = 1, untimed for `value`): the SAME step function for --sustain seconds.  K = 20 steps are 25 ms of
        # device time - too short for an outside monitor to see the GPU busy at all (the driver's 5-s samples read 0.0) and
        # short enough to be a clock burst; the rate over several seconds is reported beside it.
        if args.sustain > 0 and not dry and world == 1 and not args.share > 1:
            n_s, t_s = 0, time.perf_counter()
            while time.perf_counter() - t_s < args.sustain:
                for _ in range(100):
                    step()
                sync()
                n_s += 100
            dt_s = time.perf_counter() - t_s
            step_stats["sustained"] = {"steps": n_s, "seconds": dt_s, "ms_per_step": 1e3 * dt_s / n_s,
                                       "note": f"the same step function run for >= {args.sustain:g} s right after the timed region (a host sync every 100 steps); not `value`"}
        if profile or (use_graph and prof_ok and not eager_first):
            n_break = min(steps, 10)  # untimed extra pass of EAGER steps with every kernel timed: the full per-kernel breakdown
            lib.d4gs_profile_enable(1)
            for _ in range(n_break):
                eager_step()
            sync()
            kern_all = collect()
            if not profile:  # a replayed graph: HIP events cannot bracket its kernels - the roofline's duration is this pass's
                kern = {k: v for k, v in kern_all.items() if k.startswith("k_raster_bwd")}
                graph_note["roofline_timing"] = (f"the timed region replays a HIP graph, whose kernels HIP events cannot bracket: the roofline "
                                                 f"kernel's duration is its average over {n_break} untimed eager steps run right after it")
        if not dry:  # one untimed STAGED step with host-checked sizes: its state carries the exact counts and the
            mode_flag["deferred"] = False  # per-stage buffers (tile offsets, last ids) of the byte / pair accounting below
            mode_flag["fused"] = False
            if sharder is not None:
                sharder.fused = False
                sharder.deferred_size_check = False
            eager_step()
            sync()
        if use_dist:
            import torch.distributed as dist

            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dict(dt=dt, kern=kern, kern_all=kern_all, n_break=n_break, st=last["st"], graph_note=graph_note, step_stats=step_stats,
                    graph_used=bool(use_graph and "fallback" not in graph_note))

    def parallelism_of(mode, mesh):
        V, E = mesh
        if world == 1:
            return "1 GPU"
        if mode == "exposure":
            return f"BASELINE cfg4: exposure sub-samples of ONE frame sharded x{world}, RCCL blend + gradient all-reduce"
        if mode == "views":
            return f"views sharded x{world} (data parallel: one full frame per rank), RCCL gradient all-reduce"
        return (f"mesh {V}x{E}: {V} camera views (data parallel) x {E}-way exposure sharding of each view's {S} sub-samples; RCCL blend "
                f"all-reduce inside each view's {E}-rank group, gradient all-reduce over all {world} ranks")

    def brief(res, mode, mesh):
        """A secondary measurement as a small object."""
        V = mesh[0]
        t = res["dt"] / args.steps
        return {"value": V * N / t, "unit": "Gaussians/s", "scaling": "strong" if V == 1 else "weak", "ms_per_step": 1e3 * t,
                "frames_per_step": V, "parallelism": parallelism_of(mode, mesh),
                "launch": ("one HIP graph per step" if res["graph_used"] else "eager") + (f" ({res['graph_note']['fallback']})" if res["graph_note"].get("fallback") else "")}

    def line_of(res, mode, mesh):
        """The JSON line of one measurement: the contract fields on every rank; rank 0 of a device run adds the roofline objects
        (live kernel timings of `res` + the committed counter files)."""
        dt, kern, kern_all, n_break, st = res["dt"], res["kern"], res["kern_all"], res["n_break"], res["st"]
        graph_note, step_stats = res["graph_note"], res["step_stats"]
        V = mesh[0] if use_dist else 1
        ms_step = 1e3 * dt / args.steps
        value = V * N / (dt / args.steps)  # whole job: V frames of N Gaussians per step
        metric = "Gaussians/s fwd+bwd, 288x512, N_exposure=8" if name == "cfg2" else f"Gaussians/s fwd+bwd ({name})"
        out = {
            "metric": metric, "value": value, "unit": "Gaussians/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "strong" if V == 1 else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "dry-run (CPU / gloo control-flow check, no kernels)" if dry else "synthetic",
            "config": {"workload": f"{name}: {N} Gaussians ({G} dynamic), {K} motion bases, {W}x{H}, N_exposure={S}, "
                                    f"{channels}+depth channels, fwd+bwd to all leaves"
                                    + (f", extents x{args.scale_mul:g}" if args.scale_mul != 1.0 else ""),
                       "gaussians": N, "exposure_subsamples": S, "frames_per_step": V,
                       "parallelism": parallelism_of(mode, mesh)},
            "instances_per_s": value * S,
        }
        if V > 1:
            out["config"]["scaling_note"] = (f"{V} frames per step (one per camera view), each view's exposure sub-samples split {mesh[1]} ways: the "
                                             "per-GPU work is fixed from N = 4 on (half a frame), so the contract's word is \"weak\"; the strict BASELINE "
                                             "cfg4 number (ONE frame over all ranks, strong scaling) is the `exposure_strong_scaling` object of this line")
        lazy_on = bool(getattr(getattr(st, "cfg", None), "lazy_sort", False))
        if args.lazy_sort:
            out["config"]["workload"] += "; D4GS_LAZY_SORT"
        # the library's default (D4GS_LAZY_SORT=auto) decides by the previous render's live-row fraction and list length; say what it did
        out["config"]["lazy_sort"] = "forced on (--lazy-sort)" if args.lazy_sort else ("on (auto)" if lazy_on else "off (auto)") \
            if os.environ.get("D4GS_LAZY_SORT", "auto") == "auto" else ("on" if lazy_on else "off")
        if args.spatial_order:
            out["config"]["workload"] += f"; Gaussians in Morton order ({args.spatial_order}; control.spatial_order_step)"
        if args.share > 1:
            out["metric"] = f"DIAGNOSTIC rank-0 share of {name} at world size {args.share} (no collectives), Gaussians / t"
            out["config"]["workload"] += f"; ONLY sub-samples s % {args.share} == 0 rendered (--share)"
        out["config"]["size_check"] = ("host waits for every render's intersection counts" if args.sync_size_check else
                                       f"intersection counts of every step verified behind its launches (deferred): at most {CHECK_LAG} steps late, all of them before the clock stops")
        if res["graph_used"] or graph_note.get("fallback"):
            out["config"]["launch"] = graph_note.get("fallback") or (
                "one HIP graph per step (render forward + backward" + (" + RCCL collectives" if use_dist else "")
                + " captured, deferred size check)")
        elif world > 1:
            out["config"]["launch"] = "eager step (--no-graph)" if args.no_graph else "eager step"
        if rank != 0 or dry:
            return out
        n_isect = st.n_isect
        S_loc = st.cfg.S
        out["n_isect_per_step"] = n_isect if world == 1 else None
        out["config"]["pre_roll_steps"] = args.pre_roll
        if "sustained" in step_stats:
            sus = step_stats.pop("sustained")
            sus["value"] = N / (sus["ms_per_step"] * 1e-3)
            out["sustained"] = sus
        out["host_step_times"] = dict(step_stats, note="per-step host time between the returns of consecutive steps inside the timed region (diagnostic)")
        # (after every timed region: ~30 ms of full-rate FMA issue right in front of one cost it 0.2 % - 1.4052 against 1.4024 ms, four runs each)
        peaks = measured_peaks() if not args.no_peaks else None
        out["peaks_measured"] = peaks
        if kern:
            per = {k: v[1] / n_break for k, v in sorted(kern_all.items(), key=lambda kv: -kv[1][1])}
            out["kernels_ms_per_step"] = per
            out["kernels_note"] = (f"per-kernel breakdown from {n_break} extra untimed steps with every kernel bracketed by "
                                   "HIP events; the roofline kernel's duration comes from the timed region itself")
            dom = max(kern.items(), key=lambda kv: kv[1][1])[0]
            cnt, tot = kern[dom]
            t_k = tot / cnt * 1e-3  # seconds per launch
            # pairs actually replayed by the backward: per tile (last contributor - start + 1) * 256
            to = st.proj_out["tile_offsets"].long()
            li = st.raster["last_ids"].long()  # [S,H,W]
            tw, th = st.cfg.tiles
            pad = torch.nn.functional.pad(li, (0, tw * 16 - W, 0, th * 16 - H), value=-1)
            tmax = pad.view(S_loc, th, 16, tw, 16).amax(dim=(2, 4)).reshape(-1)
            cnts = (to[1:] - to[:-1])
            proc_bwd = torch.where(cnts > 0, (tmax - to[:-1] + 1).clamp(min=0), torch.zeros_like(cnts))
            isect_replayed = float(proc_bwd.sum().item())
            pairs_bwd = isect_replayed * 256.0
            R = 6 + channels + 1
            # algorithmic bytes of k_raster_bwd (DESIGN.md "roofline"): per intersection replayed: id 4 + emit 4 +
            # geom 32 + colours 4*DP read, gradient row R*4 written; per pixel: v_out + v_alpha + alpha + last_id + T + out.
            DP = (channels + 3) // 4 * 4
            bytes_bwd = isect_replayed * (4 + 4 + 32 + 4 * DP + R * 4) + S_loc * H * W * (8.0 * (channels + 1) + 16.0)
            default_workload = (channels == (16 if name.startswith("refdefault") else 3) and args.scale_mul == 1.0 and not args.lazy_sort
                                and not args.spatial_order and args.share <= 1 and world == 1)  # the scene the committed counter passes ran on
            if dom.startswith("k_raster"):
                fpp = flops_per_pair("bwd" in dom, channels + 1)
                flops = pairs_bwd * fpp
                tr = _traffic(name, dom) if default_workload else None
                # key order on purpose: what actually binds the kernel and the hardware fraction come BEFORE the contract's nominal fields
                roof = {"kernel": dom, "bound_actual": "valu", "frac_hardware": None,
                        "bound": "mfma", "bound_note": ("the kernel is bound by fp32 VALU issue; no MFMA is issued - the contract's "
                        "field only admits hbm|mfma, and 157.3 TFLOP/s is both the fp32 vector and the fp32-input MFMA peak") if channels + 1 <= 5 else
                        ("the 17-channel instance: fp32 VALU issue at 4 waves per SIMD (registers + 40 KB LDS); only the 16 colour channels' "
                         "gradient rows go through the matrix pipe (v_mfma_f32_*: ~3 % of the issue slots) - the contract's field only admits hbm|mfma"),
                        "achieved": flops / t_k / 1e12, "peak": F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": flops / t_k / 1e12 / F32_PEAK_TFLOPS,
                        "peak_measured": max(peaks["fp32_pk_fma_tflops"], peaks["fp32_fma_tflops"]) if peaks else None,
                        "frac_of_measured": (flops / t_k / 1e12 / max(peaks["fp32_pk_fma_tflops"], peaks["fp32_fma_tflops"])) if peaks else None,
                        "peak_measured_detail": {"v_pk_fma_f32": peaks["fp32_pk_fma_tflops"], "v_fma_f32": peaks["fp32_fma_tflops"]} if peaks else None,
                        "traffic": tr["total_1x"] if tr else None,
                        "traffic_detail": tr, "avg_launch_ms": t_k * 1e3, "pairs_per_launch": pairs_bwd,
                        "flops_per_pair": fpp,
                        "note": f"`frac` is the NOMINAL work-equivalent fraction SURVEY 8d defines: {fpp:.0f} flop x 256 pixels for every (tile, splat) "
                                "pair the kernel replays, although the kernel skips most of those pixels by design (it can exceed the "
                                "measured FMA ceiling).  `frac_hardware` (= hardware.frac_necessary) is the honest utilisation: flops the "
                                "alpha-passing lanes need / peak; `hardware.valu_busy_frac` says how busy the pipe is doing it"}
                # what the hardware does, from the committed PMC pass + the offline lane statistics of the same scene
                cur = _load_json("profiles", "pmc_current.json") or {}
                sq_doc = _load_json("profiles", f"{cur.get('round', 'r03')}_pmc_sq_{name}.json")
                sq_ok = _counters_current(sq_doc) and default_workload
                sq = sq_doc.get(dom) if sq_ok else None
                lanes = _load_json("profiles", f"{cur.get('round', 'r03')}_lane_stats_{name}.json") if sq_ok else None
                if not _counters_current(lanes):  # scripts/lane_stats.py of the same profiling round, same library
                    lanes = None
                if not sq_ok and default_workload:
                    roof["counters_note"] = ("profiles/ holds PMC counters of another build of libd4gs.so (sha256 mismatch): "
                                             "`traffic` / `hardware` / `frac_hardware` omitted rather than quoted stale; "
                                             "scripts/profile_round.sh re-measures them")
                if sq:
                    clk_cycles = sq["GRBM_GUI_ACTIVE"] / N_XCD  # the counter is summed over the 8 XCDs
                    insts = sq["SQ_INSTS_VALU"]
                    hw = {"source": f"profiles/{cur.get('round', 'r03')}_pmc_sq_{name}.json (rocprofv3 --pmc, own pass, same libd4gs.so by "
                                    f"sha256), profiles/{cur.get('round', 'r03')}_lane_stats_{name}.json (scripts/lane_stats.py on the benched scene, same library)",
                          "valu_wave_insts_per_launch": insts, "kernel_cycles": clk_cycles,
                          "cycles_per_valu_inst_per_simd": clk_cycles * N_SIMD / insts,
                          # SQ_*_CYCLES / SQ_ACTIVE_* / SQ_WAIT_* are summed over waves in units of 4 clocks (SQ_WAVE_CYCLES x 4 / (clocks x
                          # 1024 SIMDs) reproduces the resident waves per SIMD): this is the average number of waves per SIMD that have a
                          # VALU instruction in flight - ~1 = the vector pipe never idles; it exceeds 1 where issue and execution of
                          # different waves overlap (cfg3 / cfg5: 1.17 - 1.19)
                          "valu_busy_frac": sq["SQ_ACTIVE_INST_VALU"] / (clk_cycles * N_CU),
                          "waves_resident_per_simd": sq["SQ_WAVE_CYCLES"] * 4.0 / (clk_cycles * N_SIMD) if "SQ_WAVE_CYCLES" in sq else None,
                          "wave_time_valu_active_frac": sq["SQ_ACTIVE_INST_VALU"] / sq["SQ_WAVE_CYCLES"] if "SQ_WAVE_CYCLES" in sq else None,
                          "wave_time_issue_stalled_frac": sq["SQ_WAIT_INST_ANY"] / sq["SQ_WAVE_CYCLES"] if "SQ_WAVE_CYCLES" in sq and "SQ_WAIT_INST_ANY" in sq else None,
                          "lds_wave_insts_per_launch": sq.get("SQ_INSTS_LDS"), "mfma_wave_insts_per_launch": sq.get("SQ_INSTS_MFMA"),
                          "salu_wave_insts_per_launch": sq.get("SQ_INSTS_SALU"),
                          "fp32_rate_if_every_inst_were_a_full_fma_tflops": insts * 64 * 2 / t_k / 1e12}
                    if lanes:
                        af = lanes["bwd_active_lane_fraction"]
                        vp = lanes["bwd_valid_pairs"] * (isect_replayed / max(lanes["n_isect"], 1))
                        replayed_lanes = lanes["bwd_quadrant_replays"] * 64.0 * (isect_replayed / max(lanes["n_isect"], 1))
                        nec = vp * fpp + (replayed_lanes - vp) * FLOPS_INVALID_PAIR
                        hw.update(active_lane_fraction=af, replays_with_no_valid_lane=lanes["bwd_replays_with_no_valid_lane"],
                                  alpha_passing_pairs=lanes["bwd_valid_pairs"], nominal_pairs=lanes.get("bwd_nominal_pairs"),
                                  quadrant_replays=lanes["bwd_quadrant_replays"],
                                  necessary_flops_per_launch=nec, necessary_tflops=nec / t_k / 1e12,
                                  frac_necessary=nec / t_k / 1e12 / F32_PEAK_TFLOPS,
                                  necessary_note=f"{fpp:.0f} flop only for lanes that pass the alpha test, 12 for the other lanes "
                                                 "of a replayed (quadrant, splat) pair")
                        roof["frac_hardware"] = hw["frac_necessary"]
                    roof["hardware"] = hw
                if graph_note.get("roofline_timing"):
                    roof["timing_note"] = graph_note["roofline_timing"]
                out["roofline"] = roof
                out["roofline_hbm"] = {"kernel": dom, "bound": "hbm", "achieved": bytes_bwd / t_k / 1e9,
                                       "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": bytes_bwd / t_k / 1e9 / HBM_PEAK_GBS,
                                       "peak_measured": peaks["hbm_stream_copy_gbs"] if peaks else None,
                                       "frac_of_measured": (bytes_bwd / t_k / 1e9 / peaks["hbm_stream_copy_gbs"]) if peaks else None,
                                       "traffic": tr["total_1x"] if tr else None, "algorithmic_bytes": bytes_bwd}
            else:
                out["roofline"] = {"kernel": dom, "bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS,
                                   "unit": "GB/s", "frac": None, "traffic": None, "avg_launch_ms": t_k * 1e3}
            # the streaming kernels against HBM: algorithmic bytes / live duration
            SN, n_i = float(S_loc) * N, float(n_isect)
            alg = {
                "k_project_fwd": N * (12 + 16 + 12 + 4 + 4 * channels) + G * 4 * K + SN * (8 + 4 + 12 + 4 + 32 + 8 + 4)
                                 + N * (4 + 4 * DP),
                "k_emit": SN * (4 + 8 + 4 + 4) + n_i * (8 + 4),
                "k_gather": n_i * R * 4 + SN * (8 + 8 + 12 + 4) + N * (4 + 4 * DP),
                "k_project_bwd": N * (12 + 16 + 12 + 4 + 4 * channels) * 2 + G * 8 * K + SN * (4 + 12 + 8 + 12 + 4)
                                 + N * (8 + 8 * DP),
                "k_tile_sort": n_i * (8 + 4 + 4 + 4),
            }
            stream = []
            if "k_tile_sort_w" in per:  # (lists of <= 512 keys sort in the one-wave register kernel: one pass over the same bytes)
                per = dict(per, k_tile_sort=per.get("k_tile_sort", 0.0) + per["k_tile_sort_w"])
            for kname, b in alg.items():
                if kname in per and per[kname] > 0:
                    tk = per[kname] * 1e-3
                    tr = _traffic(name, kname) if channels == 3 and args.scale_mul == 1.0 else None
                    stream.append({"kernel": kname, "bound": "hbm", "algorithmic_bytes": b, "avg_launch_ms": per[kname],
                                   "achieved": b / tk / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": b / tk / 1e9 / HBM_PEAK_GBS,
                                   "peak_measured": peaks["hbm_stream_copy_gbs"] if peaks else None,
                                   "frac_of_measured": (b / tk / 1e9 / peaks["hbm_stream_copy_gbs"]) if peaks else None,
                                   "traffic": tr["total_2x"] if tr else None, "traffic_detail": tr,
                                   "traffic_frac_of_peak": (tr["total_2x"] / tk / 1e9 / HBM_PEAK_GBS) if tr else None})
            out["roofline_streaming"] = stream
        return out

    # ---- the measurements.  N = 1: one (eager by default).  N > 1: the EAGER primary step first - its line is kept - then, under the
    # watchdog, the HIP-graph version of it and the secondary shardings; one JSON line whatever happens after the first measurement.
    first_graph = want_graph and not eager_first
    wd = Watchdog(rank, args.graph_timeout) if world > 1 else None
    if wd:  # nothing to print if THIS hangs (communicator bring-up, the first collective): leave with exit code 3 instead of waiting
        wd.kick("eager primary step (communicator bring-up, first collectives)", scale=3.0)  # for RCCL's 600 s watchdog / the driver's clock
        if dry and os.environ.get("D4GS_BENCH_INJECT_HANG") == "eager" and rank == world - 1:
            time.sleep(1e6)  # tests/test_parallel_gloo.py: a rank that never joins the first collective
    res = measure(primary_mode, primary_mesh, args.steps, args.warmup, prof_ok and not first_graph, first_graph)
    out = line_of(res, primary_mode, primary_mesh)
    if world > 1:
        graph_ok = False
        if eager_first:
            out["config"]["launch"] = "eager step (the HIP-graph attempt had not finished)"
            wd.kick("HIP-graph capture + replays of the primary step (RCCL collectives inside)", out)
            resg = measure(primary_mode, primary_mesh, args.steps, args.warmup, False, True)
            graph_ok = resg["graph_used"]
            if graph_ok:
                t = resg["dt"] / args.steps
                out["eager"] = {"value": out["value"], "ms_per_step": out["ms_per_step"], "host_step_times": out.pop("host_step_times", None),
                                "note": "the same step issued eagerly (~40 launches + the collectives per step from Python), measured first"}
                out["value"], out["ms_per_step"] = primary_mesh[0] * N / t, 1e3 * t
                out["instances_per_s"] = out["value"] * S
                out["config"]["launch"] = ("one HIP graph per step (render forward + backward + RCCL collectives captured, deferred size check; "
                                           "replays reproduce the eager step's gradients); default for N > 1, --no-graph times the eager step")
                if "roofline" in out:
                    out["roofline"]["timing_note"] = ("the kernel's duration was taken with HIP events inside the EAGER timed region of this run "
                                                      "(events cannot bracket kernels of a replayed graph); same kernels, same launch geometry")
            else:
                out["config"]["launch"] = "eager step: " + (resg["graph_note"].get("fallback") or "the captured step was not used")
        cands = [("exposure", (1, world), "exposure_strong_scaling")]
        if world >= 4 and world % 2 == 0:
            cands.append(("mesh", (world // 2, 2), "views_x_exposure_mesh"))
        cands.append(("views", (world, 1), "views_weak_scaling"))
        secondaries = [(m, ms, key) for m, ms, key in cands if (m, ms) != (primary_mode, primary_mesh) and ms[1] <= S]
        for m, ms, key in secondaries:
            wd.kick(f"secondary measurement {key}", out)
            out[key] = brief(measure(m, ms, args.steps, args.warmup, False, graph_ok), m, ms)
            if key == "exposure_strong_scaling":
                out[key]["note"] = "BASELINE cfg4 in the strict sense: ONE frame, its sub-samples over all ranks; value = N / t"
            elif key == "views_x_exposure_mesh":
                out[key]["note"] = (f"{ms[0]} camera views (data parallel) x {ms[1]}-way exposure sharding of each view's frame - the decomposition of the "
                                    "reference's own step (several renders behind one backward); NOT BASELINE cfg4: value = views * N / t, weak scaling")
            else:
                out[key]["note"] = "every rank renders its own full frame (one camera view per GPU), flat gradient all-reduce; value = world * N / t"
        wd.done()
    if rank == 0:
        if not dry and not args.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(name, channels, full=args.cpu_baseline_full)
            except Exception as e:  # the checker must never take the bench line down
                out["cpu_baseline"] = {"value": None, "unit": "Gaussians/s", "cores": 1, "kind": "port",
                                       "sample": f"failed: {e!r}"}
        print(json.dumps(out))
    sys.stdout.flush()
    if use_dist:
        import torch.distributed as dist

        # RCCL prints a version banner (RCCL / HIP / ROCm version, hostname, library path) to the C-level stdout of every
        # rank, flushed at exit: the contract is ONE JSON line on stdout, so whatever the library still holds goes to
        # /dev/null from here on
        os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
