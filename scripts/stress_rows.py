"""One-off stress run: random shapes / footprints / channel counts - the composite backward's SPARSE gradient rows (cooperative
gather) against its DENSE rows, bitwise, on every gradient; plus the forward image against the torch oracle on the small cases."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.util import static_inputs, frac_bad
from oracle import raster
from deblur4dgs_amd import engine
from deblur4dgs_amd.rasterization import rasterization

rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dev = "cuda:0"
t0 = time.time()
nbad = 0
for i in range(n_cases):
    D = int(rng.choice([1, 2, 3, 4, 5, 8, 16]))
    mode = str(rng.choice(["RGB", "RGB+ED"]))
    W, H = int(rng.randint(16, 400)), int(rng.randint(16, 260))
    N = int(rng.choice([3, 60, 900, 4000, 20000]))
    sm = float(rng.choice([0.5, 2.0, 6.0, 20.0, 60.0]))
    opaque = bool(rng.randint(2))
    inp = static_inputs(N, W, H, seed=9000 + i, dtype=torch.float32, D=D, scale_mul=sm)
    if opaque:
        inp["opac"] = torch.full_like(inp["opac"], 0.999)
    bg = torch.linspace(0.1, 0.9, D)
    g = torch.Generator().manual_seed(i)
    grads = {}
    for rows in ("dense", "sparse"):
        engine.BWD_ROWS = rows
        t = {k: v.float().to(dev).requires_grad_(k in ("means", "quats", "scales", "opac", "colors")) for k, v in inp.items()}
        rc, ra, info = rasterization(t["means"], t["quats"], t["scales"], t["opac"], t["colors"], t["V"][None], t["K"][None], W, H,
                                     backgrounds=bg.to(dev)[None], render_mode=mode)
        info["means2d"].retain_grad()
        if rows == "dense":
            w = torch.randn(rc.shape, generator=g).to(dev)
        ((rc * w).sum() + 0.3 * ra.sum()).backward()
        torch.cuda.synchronize()
        grads[rows] = [t[k].grad.clone() for k in ("means", "quats", "scales", "opac", "colors")] + [info["means2d"].grad.clone()]
    same = all(torch.equal(a, b) for a, b in zip(grads["dense"], grads["sparse"]))
    finite = all(bool(torch.isfinite(a).all()) for a in grads["dense"])
    fb = 0.0
    if N <= 900 and W * H <= 40000:
        d64 = {k: v.double() for k, v in inp.items()}
        ref_c, ref_a, _ = raster.rasterization(d64["means"], d64["quats"], d64["scales"], d64["opac"], d64["colors"], d64["V"], d64["K"],
                                               W, H, background=bg.double(), render_mode=mode)
        fb = max(frac_bad(rc[0].detach().cpu(), ref_c, 1e-4), frac_bad(ra[0].detach().cpu(), ref_a, 1e-4))
    ok = same and finite and fb < 5e-3
    nbad += not ok
    if not ok or i % 20 == 0:
        print(f"{'OK ' if ok else 'BAD'} case {i}: D={D} {mode} {W}x{H} N={N} scale={sm} opaque={opaque} n_isect={info['n_isect']} "
              f"sparse==dense {same} finite {finite} fwd frac_bad {fb:.1e}", flush=True)
engine.BWD_ROWS = "auto"
print(f"{n_cases} cases, {nbad} bad, {time.time() - t0:.0f} s")
