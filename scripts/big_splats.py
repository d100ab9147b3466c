"""Sanity: how the pipeline scales when splats cover many tiles (real scenes have wide splats)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deblur4dgs_amd.synth import make_scene
from deblur4dgs_amd.exposure import render_exposure
dev = "cuda:0"
for N, mul in ((300_000, 1.0), (300_000, 3.0), (100_000, 8.0), (30_000, 20.0)):
    sc = make_scene(N, N, 6, 8, 512, 288, seed=3)
    L = {k: (v.to(dev).clone().requires_grad_() if torch.is_tensor(v) and v.is_floating_point() and k != "K" else v) for k, v in sc.items()}
    with torch.no_grad():
        L["scales"] += torch.log(torch.tensor(mul))
    def step():
        for k in ("means", "quats", "scales", "opacities", "colors", "motion_coefs"):
            L[k].grad = None
        res = render_exposure(L["means"], L["quats"], L["scales"], L["opacities"], L["colors"], 3, L["motion_coefs"], L["rots"],
                              L["transls"], L["times"], L["RTs"], L["viewmat"], sc["K"].to(dev), 512, 288, return_depth=True)
        (res["blended"].sum() + res["acc"].sum()).backward()
        return res["state"]
    for _ in range(3): st = step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): st = step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"N={N:7d} scale x{mul:4.1f}: n_isect {st.n_isect:10d} ({st.n_isect / (8 * N):6.1f} tiles/instance), longest list {st.max_tile:7d}, "
          f"{1e3 * dt:8.2f} ms / frame, {1e9 * dt / max(st.n_isect, 1):6.2f} ns / intersection")
