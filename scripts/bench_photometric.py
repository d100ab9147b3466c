"""Fused photometric loss vs the eager formulation (oracle code run in f32 on the GPU): wall and device time."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deblur4dgs_amd.losses import photometric_loss
from oracle import photometric as ph   # measurement script only (the eager baseline); not a product path

dev = "cuda:0"
B, H, W = 1, 288, 512
gt = torch.rand(B, H, W, 3, device=dev)
mask = (torch.rand(B, H, W, 1, device=dev) > 0.3).float()
base = (gt + 0.1 * torch.randn_like(gt)).clamp(0, 1)


def run(fn, n=200):
    for _ in range(20):
        x = base.clone().requires_grad_(); fn(x).backward()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        x = base.clone().requires_grad_(); fn(x).backward()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n, e0.elapsed_time(e1) / n


w_e, d_e = run(lambda x: ph.photometric_loss(x, gt, mask)[0])
w_f, d_f = run(lambda x: photometric_loss(x, gt, mask))
print(f"eager  (torch conv2d, ~85 launches): {w_e:.3f} ms wall / evaluation fwd+bwd")
print(f"fused  (3 launches)                : {w_f:.3f} ms wall / evaluation fwd+bwd")
from deblur4dgs_amd import _lib as L
import ctypes as C
lib = L.lib(); lib.d4gs_profile_enable(1)
for _ in range(50):
    x = base.clone().requires_grad_(); photometric_loss(x, gt, mask).backward()
torch.cuda.synchronize()
buf = C.create_string_buffer(1 << 16)
lib.d4gs_profile_collect(buf, C.c_size_t(len(buf)))
print(buf.value.decode()[:600])
