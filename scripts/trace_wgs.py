"""Per-workgroup timeline of the composite backward (A/B build with -DD4GS_TRACE): when each workgroup started / ended (wall
clock, 100 MHz), on which XCD / SE / CU, and how many list entries it owned.  usage:
  python -m deblur4dgs_amd.build --ab trace raster_bwd.hip -DD4GS_TRACE
  D4GS_LIB_PATH=scripts/ablate/libd4gs_trace.so python scripts/trace_wgs.py [--share 8] [--seg 0|1]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--share", type=int, default=8)
ap.add_argument("--seg", default=None)
ap.add_argument("--fwd", action="store_true", help="trace the forward composite (8 words per workgroup incl. phase times)")
a = ap.parse_args()
if a.seg is not None:
    os.environ["D4GS_SEG"] = a.seg
import torch, bench
from deblur4dgs_amd import engine
from deblur4dgs_amd.exposure import render_exposure

dev = torch.device("cuda:0")
N, G, K, S, W, H = bench.CONFIGS["cfg2"]
sc, d, leaves, wimg, wacc = bench.make_inputs("cfg2", dev, channels=3)
L = dict(leaves)
times = leaves["times"][::a.share].detach().clone().requires_grad_()
RTs = leaves["RTs"][::a.share].detach().clone().requires_grad_()
bg = torch.ones(3, device=dev)
NB = 1 << 16
RW = 10 if a.fwd else 4
trace = torch.zeros(NB * RW, dtype=torch.int64, device=dev)


def step():
    res = render_exposure(L["means"], L["quats"], L["scales"], L["opacities"], L["colors"], 3, L["motion_coefs"], L["rots"], L["transls"],
                          times, RTs, L["viewmat"], d["K"], W, H, background=bg, return_depth=True, fused=True)
    (torch.dot(res["blended"].reshape(-1), wimg.reshape(-1)) + torch.dot(res["acc"].reshape(-1), wacc.reshape(-1))).backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
os.environ["D4GS_TRACE_FWD_PTR" if a.fwd else "D4GS_TRACE_PTR"] = hex(trace.data_ptr())
trace.zero_()
step()
torch.cuda.synchronize()
t = trace.view(NB, RW).cpu()
used = t[:, 1] > 0
t = t[used]
t0 = int(t[:, 0].min())
st, en = (t[:, 0] - t0).double() / 100.0, (t[:, 1] - t0).double() / 100.0  # us
life = en - st
n = t[:, 3]
act = n > 0
hw = t[:, 2] & 0xffffffff
xcc = (t[:, 2] >> 32) & 0xf
cu = (hw >> 8) & 0xf
se = (hw >> 13) & 0x7
sh = (hw >> 12) & 1
print(f"share {a.share} seg {os.environ.get('D4GS_SEG')}: {int(used.sum())} workgroups traced, {int(act.sum())} with work; kernel span {float(en.max()):.1f} us")
print(f"  start times of working WGs (us): p0 {float(st[act].min()):.1f} p50 {float(st[act].median()):.1f} p90 {float(st[act].quantile(0.9)):.1f} max {float(st[act].max()):.1f}")
print(f"  lifetimes of working WGs (us):   p10 {float(life[act].quantile(0.1)):.1f} p50 {float(life[act].median()):.1f} p90 {float(life[act].quantile(0.9)):.1f} max {float(life[act].max()):.1f}")
print(f"  end times (us): p50 {float(en[act].median()):.1f} p90 {float(en[act].quantile(0.9)):.1f} p99 {float(en[act].quantile(0.99)):.1f} max {float(en[act].max()):.1f}")
print(f"  list entries per working WG: p10 {int(n[act].double().quantile(0.1))} p50 {int(n[act].median())} max {int(n[act].max())}")
key = (xcc * 8 + se) * 32 + sh * 16 + cu
per_cu = torch.zeros(int(key.max()) + 1)
per_cu.index_add_(0, key[act], life[act].float())
busy = per_cu[per_cu > 0]
print(f"  {len(busy)} distinct (xcc, se, sh, cu) ids; summed WG lifetime per id (us): min {float(busy.min()):.0f} p50 {float(busy.median()):.0f} max {float(busy.max()):.0f}")
if a.fwd:
    for nm, col in (("staging (global loads -> LDS -> barrier)", 4), ("row-list building", 5), ("compositing", 6)):
        v = t[:, col].double()[act] / 100.0
        print(f"  per-WG time in {nm}: p10 {float(v.quantile(0.1)):.1f} p50 {float(v.median()):.1f} p90 {float(v.quantile(0.9)):.1f} us")
if a.fwd:
    mhz = t[:, 8].double()[act] / life[act].clamp(min=1e-3)
    print(f"  shader clock over the WG lifetimes (s_memtime ticks per us of wall clock): p10 {float(mhz.quantile(0.1)):.0f} p50 {float(mhz.median()):.0f} p90 {float(mhz.quantile(0.9)):.0f} MHz")
    it = t[:, 7].double()[act]
    comp = t[:, 6].double()[act] / 100.0
    print(f"  walk iterations of wave 0 per WG: p10 {int(it.quantile(0.1))} p50 {int(it.median())} p90 {int(it.quantile(0.9))}; compositing ns per iteration: p50 {float((1e3 * comp / it.clamp(min=1)).median()):.0f}")
for lo in range(0, int(en.max()) + 20, 20):
    alive = int(((st <= lo) & (en > lo) & act).sum())
    print(f"    t = {lo:4d} us: {alive} working WGs alive")
