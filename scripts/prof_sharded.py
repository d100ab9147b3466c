"""Where does the exposure-sharded step spend its time at world size 1 (RCCL)?  torch.profiler kernel + CPU-op summary."""
import os, socket, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.distributed as dist
import bench
from deblur4dgs_amd.parallel import ShardedExposure

s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
sc, d, leaves, wimg, wacc = bench.make_inputs("cfg2", dev)
bg = torch.ones(3, device=dev)
sh = ShardedExposure(1, 0, mode=sys.argv[1] if len(sys.argv) > 1 else "exposure")
N, G, K, S, W, H = bench.CONFIGS["cfg2"]
for _ in range(5):
    sh.step(leaves, d["K"], W, H, bg, wimg, wacc)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(10):
        sh.step(leaves, d["K"], W, H, bg, wimg, wacc)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=60))
dist.destroy_process_group()
