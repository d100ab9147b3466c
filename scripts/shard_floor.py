"""What exposure sharding can reach at most: the single-GPU time of ONE rank's share of cfg2 at world size P (S / P sub-samples of the
same 300 k Gaussians, no collectives) against the full frame - the per-Gaussian kernels do not shrink with P."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from deblur4dgs_amd import engine
from deblur4dgs_amd.exposure import render_exposure

dev = torch.device("cuda:0")
N, G, K, S, W, H = bench.CONFIGS["cfg2"]
sc, d, leaves, wimg, wacc = bench.make_inputs("cfg2", dev, channels=3)
bg = torch.ones(3, device=dev)
for P in (1, 2, 4, 8):
    sl = slice(0, S, P)  # rank 0's sub-samples {s : s % P == 0}
    L = dict(leaves)
    times, RTs = leaves["times"][sl].detach().clone().requires_grad_(), leaves["RTs"][sl].detach().clone().requires_grad_()
    def step():
        for v in leaves.values():
            v.grad = None
        res = render_exposure(L["means"], L["quats"], L["scales"], L["opacities"], L["colors"], 3, L["motion_coefs"], L["rots"], L["transls"],
                              times, RTs, L["viewmat"], d["K"], W, H, background=bg, return_depth=True, deferred_size_check=True, fused=True)
        (torch.dot(res["blended"].reshape(-1), wimg.reshape(-1)) + torch.dot(res["acc"].reshape(-1), wacc.reshape(-1))).backward()
        engine.check_deferred()
    for _ in range(5):
        step()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 50
    for v in leaves.values():
        v.grad = None
    times.grad = RTs.grad = None
    import contextlib
    watch = engine.GraphWatch() if os.environ.get("SHARD_WATCH") else None  # (SHARD_WATCH=1: what the size-reporting node costs a replay)
    with (watch.capturing() if watch else contextlib.nullcontext()), torch.cuda.graph(g):
        step_g = step  # (deferred checks are skipped under capture)
        step_g()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        g.replay()
    torch.cuda.synchronize()
    dg = (time.perf_counter() - t0) / 50
    lib = __import__("deblur4dgs_amd._lib", fromlist=["lib"]).lib()
    import ctypes as C
    lib.d4gs_profile_enable(1)
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    lib.d4gs_profile_enable(0)
    buf = C.create_string_buffer(1 << 16)
    lib.d4gs_profile_collect(buf, C.c_size_t(len(buf)))
    rows = sorted(((ln.split()[0], float(ln.split()[2]) / 10) for ln in buf.value.decode().splitlines()), key=lambda r: -r[1])
    kern = ", ".join(f"{n} {1e3 * ms:.0f}" for n, ms in rows)
    print(f"P={P}: kernels (us per step, HIP events, eager): {kern}; sum {1e3 * sum(ms for _, ms in rows):.0f}")
    print(f"P={P}: rank 0 renders {len(range(0, S, P))} of {S} sub-samples: {1e3 * dt:.3f} ms per step eager, {1e3 * dg:.3f} ms from a HIP graph "
          f"(no collectives) -> at most {1.403e-3 / dg if P > 1 else 1.0:.2f}x")
