"""Can the exposure-sharded step (RCCL all-gathers + gradient all-reduce included) be captured in a HIP graph?
World size 1 only (the pool's boxes have one GPU).  Run under `timeout`."""
import os, socket, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.distributed as dist
import bench
from deblur4dgs_amd import engine
from deblur4dgs_amd.parallel import ShardedExposure

s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
name = sys.argv[1] if len(sys.argv) > 1 else "tiny"
mode = sys.argv[2] if len(sys.argv) > 2 else "exposure"
sc, d, leaves, wimg, wacc = bench.make_inputs(name, dev)
bg = torch.ones(3, device=dev)
N, G, K, S, W, H = bench.CONFIGS[name]
sh = ShardedExposure(1, 0, mode=mode)
sh.deferred_size_check = True
for _ in range(5):
    sh.step(leaves, d["K"], W, H, bg, wimg, wacc)
    engine.check_deferred()
torch.cuda.synchronize()
ref = {k: v.grad.clone() for k, v in leaves.items() if v.grad is not None}
t0 = time.perf_counter()
for _ in range(100):
    sh.step(leaves, d["K"], W, H, bg, wimg, wacc)
    engine.check_deferred()
torch.cuda.synchronize()
print(name, mode, "eager %.3f ms / step" % ((time.perf_counter() - t0) * 10), flush=True)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        sh.step(leaves, d["K"], W, H, bg, wimg, wacc)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
print("capturing", flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    sh.step(leaves, d["K"], W, H, bg, wimg, wacc)
print("captured", flush=True)
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
print("replayed", flush=True)
got = {k: v.grad.clone() for k, v in leaves.items() if v.grad is not None}
for k in ref:
    print(k, "bitwise equal to eager:", torch.equal(ref[k], got[k]))
t0 = time.perf_counter()
for _ in range(100):
    g.replay()
torch.cuda.synchronize()
print(name, mode, "graph %.3f ms / step" % ((time.perf_counter() - t0) * 10), flush=True)
dist.destroy_process_group()
