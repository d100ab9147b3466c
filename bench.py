#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on its config: "Gaussians/s fwd+bwd, 288x512, N_exposure=8".

A step = ONE blurry frame, forward + backward: raw leaf params -> activations -> motion-basis deformation of all
S exposure sub-samples -> camera delta -> projection -> tile binning + per-tile depth sort -> composite ->
exposure blend -> loss = <blended, Wimg> + <acc, Wacc> -> gradients to every leaf (means, quats, scales,
opacities, colours, motion coefficients, bases, times, camera deltas, viewmat).  Inputs are resident in HBM
before the timed region.  value = Gaussians / t_frame (whole job); `instances_per_s` = N*S / t_frame.

N GPUs (torchrun, one rank per GPU, RCCL):
  --shard views (default): every rank renders its own full S-sub-sample frame (data parallel over camera views, the
      unit a training step batches: flow3d/trainer.py:211-222 renders 3 independent groups per step), leaf gradients
      all-reduced in one flat buffer.  Per-GPU work fixed -> "scaling": "weak"; value = world * N / t.
  --shard exposure: BASELINE config 4 - the S sub-samples of the SAME frame are split over the ranks, the blended
      image is an all-reduce (SUM, + MAX/MIN channels), leaf gradients are all-reduced.  Total work is fixed ->
      "scaling": "strong" (latency-bound: one sub-sample is ~0.5 ms of GPU work against a 24 MB gradient all-reduce).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (N, G, K, S, W, H)
    "cfg2": (300_000, 300_000, 6, 8, 512, 288),
    "cfg3": (300_000, 300_000, 6, 8, 1280, 720),
    "cfg5": (1_000_000, 1_000_000, 12, 16, 1280, 720),
    "tiny": (20_000, 12_000, 4, 4, 256, 144),
}
SEEDS = {"cfg2": 1001, "cfg3": 1002, "cfg5": 1004, "tiny": 1099}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
F32_PEAK_TFLOPS = 157.3  # fp32 vector peak == fp32-input dense MFMA peak (MI355X_MICROARCH.md)
# ALGORITHMIC FP32 ops per (splat, pixel) of the reference composite (gsplat rasterize_to_pixels fwd / bwd at 4
# channels), SURVEY.md section 8(d): ~30 forward, ~90 backward.  The HIP kernels execute fewer.
FLOPS_PER_PAIR_BWD = 90.0
FLOPS_PER_PAIR_FWD = 30.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="cfg2", choices=list(CONFIGS))
    ap.add_argument("--shard", default="views", choices=["exposure", "views"])
    ap.add_argument("--channels", type=int, default=3, choices=[3, 16],
                    help="colour channels before depth: 3 = RGB+ED (headline), 16 = the reference's dynamic-training "
                         "shape (rgb + mask + 4x3 track channels + depth = 17, scene_model.py:233-296)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--force-dist", action="store_true", help="exercise the RCCL code path with world_size 1")
    return ap.parse_args()


def to_dev(sc, dev):
    return {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in sc.items()}


def make_inputs(name, dev, seed_offset=0, channels=3):
    from deblur4dgs_amd.synth import make_scene

    N, G, K, S, W, H = CONFIGS[name]
    sc = make_scene(N, G, K, S, W, H, seed=SEEDS[name] + seed_offset, D=channels)
    d = to_dev(sc, dev)
    leaves = {k: d[k].clone().requires_grad_() for k in
              ("means", "quats", "scales", "opacities", "colors", "motion_coefs", "rots", "transls", "times", "RTs",
               "viewmat")}
    g = torch.Generator().manual_seed(7)
    wimg = torch.randn(H, W, channels + 1, generator=g).to(dev)
    wacc = torch.randn(H, W, generator=g).to(dev)
    return sc, d, leaves, wimg, wacc


def cpu_baseline(name):
    """The build's CPU restatement (oracle; the reference has NO CPU path - flow3d/scene_model.py:36,360) timed
    on this box's host cores on a bounded sample: ONE exposure sub-sample (the middle one) of the same seeded scene,
    forward + backward through oracle/raster_ref.c (scalar C, 1 core), deformation applied by oracle/deform.py."""
    import numpy as np

    from deblur4dgs_amd.synth import make_scene
    from oracle import cref, deform

    N, G, K, S, W, H = CONFIGS[name]
    sc = make_scene(N, G, K, S, W, H, seed=SEEDS[name])
    s = S // 2
    t0 = time.perf_counter()
    with torch.no_grad():
        fg = {k: sc[k] for k in ("means", "quats", "motion_coefs")}
        m, q = deform.compute_poses_fg(sc["times"][s:s + 1], fg["means"], fg["quats"], fg["motion_coefs"], sc["rots"],
                                       sc["transls"])
        m = deform.camera_delta(m[:, 0], sc["RTs"][s])
        q = q[:, 0]
        scales, opac, cols = torch.exp(sc["scales"]), torch.sigmoid(sc["opacities"]), torch.sigmoid(sc["colors"])
    out, al, ctx = cref.rasterization(m.numpy(), q.numpy(), scales.numpy(), opac.numpy(), cols.numpy(),
                                      sc["viewmat"].numpy(), sc["K"].numpy(), W, H, background=np.ones(3, np.float32),
                                      render_mode="RGB+ED", dtype=np.float32)
    rng = np.random.default_rng(0)
    cref.backward(ctx, rng.standard_normal(out.shape).astype(np.float32),
                  rng.standard_normal(al.shape).astype(np.float32))
    t_sub = time.perf_counter() - t0
    return {
        "value": N / (t_sub * S), "unit": "Gaussians/s", "cores": 1, "kind": "port",
        "sample": f"1 of {S} exposure sub-samples of {name} ({N} Gaussians, {W}x{H}), fwd+bwd through the oracle "
                  f"(torch deform + scalar C rasterizer, 1 thread) in {t_sub:.2f} s; value = N / (S * t_sub)",
        "n_isect_sample": ctx["n_isect"],
        "cpu_model": _cpu_model(), "host_cores": os.cpu_count(),
    }


def _cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        import torch.distributed as dist

        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)

    from deblur4dgs_amd import _lib as L
    from deblur4dgs_amd.exposure import render_exposure
    from deblur4dgs_amd.parallel import ShardedExposure

    name = args.config
    N, G, K, S, W, H = CONFIGS[name]
    views = world > 1 and args.shard == "views"
    sc, d, leaves, wimg, wacc = make_inputs(name, dev, seed_offset=rank if views else 0, channels=args.channels)
    bg = torch.ones(args.channels, device=dev)
    sharder = None
    if use_dist:
        sharder = ShardedExposure(world, rank, mode=args.shard)

    last = {}

    def step():
        for v in leaves.values():
            v.grad = None
        if sharder is None:
            res = render_exposure(leaves["means"], leaves["quats"], leaves["scales"], leaves["opacities"],
                                  leaves["colors"], 3, leaves["motion_coefs"], leaves["rots"], leaves["transls"],
                                  leaves["times"], leaves["RTs"], leaves["viewmat"], d["K"], W, H, background=bg,
                                  return_depth=True)
            loss = torch.dot(res["blended"].reshape(-1), wimg.reshape(-1)) + torch.dot(res["acc"].reshape(-1), wacc.reshape(-1))
            loss.backward()
        else:
            res = sharder.step(leaves, d["K"], W, H, bg, wimg, wacc)
        last["res"] = res

    def sync():
        if use_dist:
            import torch.distributed as dist

            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    lib = L.lib()
    prof = not args.no_profile
    sync()
    if prof:
        lib.d4gs_profile_enable(2)  # HIP events around the rasterization kernels only: timing EVERY kernel costs two
    t0 = time.perf_counter()      # stream events per launch, ~0.12 ms of a 1.6 ms frame
    for _ in range(args.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    def collect():
        lib.d4gs_profile_enable(0)
        buf = C.create_string_buffer(1 << 16)
        lib.d4gs_profile_collect(buf, C.c_size_t(len(buf)))
        got = {}
        for line in buf.value.decode().splitlines():
            nm, cnt, ms = line.split()
            got[nm] = (int(cnt), float(ms))
        return got

    kern, kern_all, n_break = {}, {}, 0
    if prof:
        kern = collect()  # the dominant kernels, measured live over the timed region
        n_break = min(args.steps, 10)  # untimed extra pass with every kernel timed: the full per-kernel breakdown
        lib.d4gs_profile_enable(1)
        for _ in range(n_break):
            step()
        sync()
        kern_all = collect()
    if use_dist:
        import torch.distributed as dist

        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_step = 1e3 * dt / args.steps
    frames_per_step = world if views else 1
    value = frames_per_step * N / (dt / args.steps)

    out = {
        "metric": "Gaussians/s fwd+bwd, 288x512, N_exposure=8" if name == "cfg2" else f"Gaussians/s fwd+bwd ({name})",
        "value": value, "unit": "Gaussians/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak" if args.shard == "views" else "strong",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{name}: {N} Gaussians ({G} dynamic), {K} motion bases, {W}x{H}, N_exposure={S}, "
                                f"{args.channels}+depth channels, fwd+bwd to all leaves", "gaussians": N, "exposure_subsamples": S,
                   "parallelism": "1 GPU" if world == 1 else f"{args.shard}-sharded x{world} (RCCL)"},
        "instances_per_s": value * S,
    }
    if rank == 0:
        st = last["res"]["state"] if isinstance(last["res"], dict) else last["res"]
        n_isect = st.n_isect
        S_loc = st.cfg.S
        out["n_isect_per_step"] = n_isect if sharder is None else None
        if kern:
            out["kernels_ms_per_step"] = {k: v[1] / n_break for k, v in sorted(kern_all.items(), key=lambda kv: -kv[1][1])}
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
            pairs_bwd = float(proc_bwd.sum().item()) * 256.0
            R = 6 + args.channels + 1
            # algorithmic bytes of k_raster_bwd (DESIGN.md "roofline"): per intersection replayed: id 4 + emit 4 +
            # geom 32 + colours 16 read, gradient row R*4 written; per pixel: v_out 16 + v_alpha 4 + alpha 4 +
            # last_id 4 + out 16 read.
            bytes_bwd = float(proc_bwd.sum().item()) * (4 + 4 + 32 + 16 + R * 4) + S_loc * H * W * 44.0
            traffic = None
            try:  # HBM bytes per launch from the committed rocprofv3 PMC passes (MI355X_MICROARCH.md, HBM section)
                pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))[name]["kernels"][dom]
                if args.channels == 3:
                    traffic = (pm["FETCH_SIZE"] + pm["WRITE_SIZE"]) * 1024.0
            except Exception:
                traffic = None
            if dom.startswith("k_raster"):
                flops = pairs_bwd * (FLOPS_PER_PAIR_BWD if "bwd" in dom else FLOPS_PER_PAIR_FWD)
                out["roofline"] = {"kernel": dom, "bound": "mfma", "achieved": flops / t_k / 1e12,
                                   "peak": F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": flops / t_k / 1e12 / F32_PEAK_TFLOPS,
                                   "traffic": traffic, "avg_launch_ms": t_k * 1e3, "pairs_per_launch": pairs_bwd,
                                   "traffic_note": "FETCH_SIZE + WRITE_SIZE (KB) x 1024 from separate rocprofv3 --pmc passes "
                                                   "(profiles/pmc_traffic.json); gfx950's FETCH_SIZE can under-count wide "
                                                   "coalesced reads 2x, so true read traffic lies between 1x and 2x the "
                                                   "fetch term",
                                   "note": "fp32 VALU-bound composite; algorithmic ops of the reference per (splat, pixel) "
                                           "of each 16x16 tile x pairs replayed; fp32 vector peak == fp32-input dense "
                                           "MFMA peak (157.3 TFLOP/s, MI355X_MICROARCH.md); no MFMA is issued"}
                out["roofline_hbm"] = {"kernel": dom, "bound": "hbm", "achieved": bytes_bwd / t_k / 1e9,
                                       "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": bytes_bwd / t_k / 1e9 / HBM_PEAK_GBS,
                                       "traffic": traffic, "algorithmic_bytes": bytes_bwd}
            else:
                out["roofline"] = {"kernel": dom, "bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS,
                                   "unit": "GB/s", "frac": None, "traffic": None, "avg_launch_ms": t_k * 1e3}
        if not args.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(name)
            except Exception as e:  # the checker must never take the bench line down
                out["cpu_baseline"] = {"value": None, "unit": "Gaussians/s", "cores": 1, "kind": "port",
                                       "sample": f"failed: {e!r}"}
        print(json.dumps(out))
    if use_dist:
        import torch.distributed as dist

        dist.destroy_process_group()


if __name__ == "__main__":
    main()
