"""Wall-clock per phase of the example training step (host-bound vs device-bound)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib.util
spec = importlib.util.spec_from_file_location("ex", os.path.join(os.path.dirname(__file__), "..", "examples", "train_dynamic_step.py"))
ex = importlib.util.module_from_spec(spec); spec.loader.exec_module(ex)
dev = "cuda:0"; W, H = 512, 288
model, sc = ex.build(dev=dev)
w2c, K = sc["viewmat"][None].to(dev), sc["K"][None].to(dev)
tt = torch.tensor([1.0, 2.0, 4.0, 5.0], device=dev); tw = w2c.expand(4, 4, 4).contiguous()
def sync(): torch.cuda.synchronize(); return time.perf_counter()
acc = {}
def add(k, dt): acc[k] = acc.get(k, 0) + dt
for it in range(13):
    if it == 3: acc.clear()
    t0 = sync()
    RTs, times, dT = model.move_model.forward_start_end_mid({"R": w2c[0, :3, :3], "T": w2c[0, :3, 3:], "timestep": 3}, num_cameras=11)
    t1 = sync(); add("move_model fwd only", t1 - t0)
    (RTs.sum() + times.sum()).backward()
    t2 = sync(); add("move_model bwd only", t2 - t1)
    model.zero_grad(set_to_none=True)
    t0 = sync()
    o1 = model.render(3, w2c, K, (W, H), bg_only=True, return_depth=True, return_mask=True, mode="blury")
    t1 = sync(); add("render bg blury fwd", t1 - t0)
    o2 = model.render(3, w2c, K, (W, H), target_ts=tt, target_w2cs=tw, return_depth=True, return_mask=True, mode="blury")
    t2 = sync(); add("render dyn blury fwd (17ch)", t2 - t1)
    o3 = model.render(3, w2c, K, (W, H), bg_only=True, return_depth=True, mode="mid")
    t3 = sync(); add("render bg mid fwd", t3 - t2)
    loss = o1["img"].mean() + o2["img"].mean() + o3["img"].mean() + o2["tracks_3d"].mean()
    loss.backward()
    t4 = sync(); add("backward (all)", t4 - t3)
for k, v in acc.items():
    print(f"{k:32s} {1e3 * v / 10:8.2f} ms")
