"""Host-side cost of the exposure-sharded step (world size 1, RCCL) on bench's `tiny` scene: cProfile of the main thread
plus wall-clock of loss.backward() and of FlatGradAllReduce.reduce()."""
import cProfile, io, os, pstats, socket, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.distributed as dist
import bench
from deblur4dgs_amd import engine
from deblur4dgs_amd.parallel import ShardedExposure

s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
name = "tiny"
sc, d, leaves, wimg, wacc = bench.make_inputs(name, dev)
bg = torch.ones(3, device=dev)
mode = sys.argv[1] if len(sys.argv) > 1 else "exposure"
sh = ShardedExposure(1, 0, mode=mode)
sh.deferred_size_check = True
N, G, K, S, W, H = bench.CONFIGS[name]
def step():
    sh.step(leaves, d["K"], W, H, bg, wimg, wacc)
    engine.check_deferred()
for _ in range(20):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    step()
torch.cuda.synchronize()
print(mode, "%.3f ms / step" % ((time.perf_counter() - t0) / 0.2))
pr = cProfile.Profile(); pr.enable()
for _ in range(200):
    step()
torch.cuda.synchronize()
pr.disable()
st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(22); print(st.getvalue()[:5000])
dist.destroy_process_group()
