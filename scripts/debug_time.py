import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from deblur4dgs_amd.exposure import render_exposure
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
ch = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
N, G, K, S, W, H = bench.CONFIGS[name]
sc, d, leaves, wimg, wacc = bench.make_inputs(name, dev, channels=ch)
bg = torch.ones(ch, device=dev)
def sync(): torch.cuda.synchronize(); return time.perf_counter()
for it in range(4):
    for v in leaves.values(): v.grad = None
    t0 = sync()
    res = render_exposure(leaves["means"], leaves["quats"], leaves["scales"], leaves["opacities"], leaves["colors"], 3, leaves["motion_coefs"], leaves["rots"], leaves["transls"], leaves["times"], leaves["RTs"], leaves["viewmat"], d["K"], W, H, background=bg, return_depth=True)
    t1 = sync()
    loss = (res["blended"] * wimg).sum() + (res["acc"] * wacc).sum()
    t2 = sync()
    loss.backward()
    t3 = sync()
    print(f"it{it}: fwd {1e3*(t1-t0):.2f} ms  loss {1e3*(t2-t1):.2f}  bwd {1e3*(t3-t2):.2f}  mem {torch.cuda.memory_allocated()/1e9:.2f} GB reserved {torch.cuda.memory_reserved()/1e9:.2f}", flush=True)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for v in leaves.values(): v.grad = None
    res = render_exposure(leaves["means"], leaves["quats"], leaves["scales"], leaves["opacities"], leaves["colors"], 3, leaves["motion_coefs"], leaves["rots"], leaves["transls"], leaves["times"], leaves["RTs"], leaves["viewmat"], d["K"], W, H, background=bg, return_depth=True)
    loss = (res["blended"] * wimg).sum() + (res["acc"] * wacc).sum()
    loss.backward()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=50))
print(prof.key_averages().table(sort_by="cpu_time_total", row_limit=10, max_name_column_width=50))
