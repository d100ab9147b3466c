"""One-off stress run: many random shapes / footprints against the torch oracle (forward image + tile-list order)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.util import static_inputs, frac_bad, rel_err
from oracle import raster
from deblur4dgs_amd.rasterization import rasterization

rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 60
dev = "cuda:0"
worst = 0.0
t0 = time.time()
for i in range(n_cases):
    D = int(rng.choice([1, 3, 4, 5, 16]))
    mode = str(rng.choice(["RGB", "RGB+ED", "RGB+D"]))
    W, H = int(rng.randint(16, 260)), int(rng.randint(16, 180))
    N = int(rng.choice([2, 50, 700, 2500, 6000]))
    sm = float(rng.choice([0.3, 1.0, 3.0, 10.0, 40.0]))
    inp = static_inputs(N, W, H, seed=5000 + i, dtype=torch.float64, D=D, scale_mul=sm)
    bg = torch.linspace(0.1, 0.9, D, dtype=torch.float64) if rng.randint(2) else None
    ref_c, ref_a, ref_info = raster.rasterization(inp["means"], inp["quats"], inp["scales"], inp["opac"], inp["colors"],
                                                  inp["V"], inp["K"], W, H, background=bg, render_mode=mode)
    t = {k: v.float().to(dev) for k, v in inp.items()}
    ec = bool(rng.randint(2))
    rc, ra, info = rasterization(t["means"], t["quats"], t["scales"], t["opac"], t["colors"], t["V"][None], t["K"][None], W, H,
                                 backgrounds=None if bg is None else bg.float().to(dev)[None], render_mode=mode, exact_cull=ec)
    fb = max(frac_bad(rc[0].cpu(), ref_c, 1e-4), frac_bad(ra[0].cpu(), ref_a, 1e-4))
    worst = max(worst, fb)
    # every tile list sorted by depth
    offs = torch.cat([info["isect_offsets"].flatten().cpu().long(), torch.tensor([info["n_isect"]])])
    depth = info["depths"][0].cpu(); ids = info["flatten_ids"].cpu().long()
    d = depth[ids]
    bad = 0
    if len(d) > 1:
        same_tile = torch.ones(len(d) - 1, dtype=torch.bool)
        cut = offs[1:-1]
        cut = cut[(cut > 0) & (cut < len(d))] - 1   # pair (cut-1, cut) straddles a tile boundary
        same_tile[cut] = False
        bad = int(((d[1:] < d[:-1]) & same_tile).sum())
    status = "OK " if fb < 5e-3 and bad == 0 else "BAD"
    if status == "BAD" or i % 10 == 0:
        print(f"{status} case {i}: D={D} {mode} {W}x{H} N={N} scale={sm} exact_cull={ec} n_isect={info['n_isect']} frac_bad={fb:.2e} unsorted={bad}")
print(f"{n_cases} cases, worst frac_bad {worst:.2e}, {time.time() - t0:.0f} s")
