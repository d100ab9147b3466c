"""Which part of a SceneModel training step survives HIP-graph capture + replay?  Each variant runs in its own process."""
import os, subprocess, sys
VARIANTS = ["big_plain", "big_tracks", "big_three", "big_three_adam", "big_three_adam_nofuse"]
if len(sys.argv) == 1:
    for v in VARIANTS:
        r = subprocess.run([sys.executable, __file__, v], capture_output=True, text=True, timeout=300)
        tail = (r.stdout + r.stderr).strip().splitlines()[-1:] if (r.stdout + r.stderr).strip() else [""]
        print(f"{v:12s} rc={r.returncode} {tail[0][:150]}")
    sys.exit(0)
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib.util
spec = importlib.util.spec_from_file_location("ex", os.path.join(os.path.dirname(__file__), "..", "examples", "train_dynamic_step.py"))
ex = importlib.util.module_from_spec(spec); spec.loader.exec_module(ex)
from deblur4dgs_amd.losses import photometric_loss
v = sys.argv[1]
dev = "cuda:0"; W, H = (512, 288) if v.startswith("big") else (128, 96)
model, sc = ex.build(W=W, H=H, dev=dev) if v.startswith("big") else ex.build(n_fg=3000, n_bg=5000, K=6, W=W, H=H, dev=dev)
model.deferred_size_check = True
w2c, K = sc["viewmat"][None].to(dev), sc["K"][None].to(dev)
tt = torch.tensor([1.0, 2.0, 4.0, 5.0], device=dev); tw = w2c.expand(4, 4, 4).contiguous()
tgt = torch.rand(1, H, W, 3, device=dev)
NN = model.num_gaussians
stats = {"xys_grad_norm_acc": torch.zeros(NN, device=dev), "vis_count": torch.zeros(NN, dtype=torch.int64, device=dev),
         "max_radii": torch.zeros(NN, device=dev)}
opts = []
if "adam" in v:
    opts = [torch.optim.Adam([p], lr=1e-4, fused="nofuse" not in v) for p in model.parameters()]
params = list(model.parameters())


def step():
    if v == "plain":
        o = model.render(3, w2c, K, (W, H), mode="blury"); loss = o["img"].sum()
    elif v == "mask_depth":
        o = model.render(3, w2c, K, (W, H), bg_only=True, return_depth=True, return_mask=True, mode="blury"); loss = o["img"].sum() + o["depth"].sum()
    elif v == "tracks":
        o = model.render(3, w2c, K, (W, H), target_ts=tt, target_w2cs=tw, return_depth=True, return_mask=True, mode="blury")
        loss = o["img"].sum() + o["tracks_3d"].square().mean()
    elif v == "photo":
        o = model.render(3, w2c, K, (W, H), mode="blury"); loss = photometric_loss(o["img"], tgt)
    elif v == "stats":
        model.attach_control_stats(stats, batch_size=1)
        o = model.render(3, w2c, K, (W, H), mode="blury"); loss = o["img"].sum()
        model.detach_control_stats()
    elif v == "mid":
        o = model.render(3, w2c, K, (W, H), bg_only=True, return_depth=True, mode="mid"); loss = o["img"].sum()
    elif v == "two_renders":
        o1 = model.render(3, w2c, K, (W, H), mode="blury"); o2 = model.render(3, w2c, K, (W, H), bg_only=True, mode="blury")
        loss = o1["img"].sum() + o2["img"].sum()
    elif v == "big_plain":
        o = model.render(3, w2c, K, (W, H), mode="blury"); loss = o["img"].sum()
    elif v == "big_tracks":
        o = model.render(3, w2c, K, (W, H), target_ts=tt, target_w2cs=tw, return_depth=True, return_mask=True, mode="blury")
        loss = o["img"].sum() + o["tracks_3d"].square().mean()
    elif v.startswith("big_three"):
        o1 = model.render(3, w2c, K, (W, H), bg_only=True, return_depth=True, return_mask=True, mode="blury")
        model.attach_control_stats(stats, batch_size=1)
        o2 = model.render(3, w2c, K, (W, H), target_ts=tt, target_w2cs=tw, return_depth=True, return_mask=True, mode="blury")
        model.detach_control_stats()
        o3 = model.render(3, w2c, K, (W, H), bg_only=True, return_depth=True, mode="mid")
        tg = tgt
        loss = photometric_loss(o1["img"], tg) + photometric_loss(o2["img"], tg) + 0.1 * photometric_loss(o3["img"], tg) + 1e-3 * o2["tracks_3d"].square().mean()
    loss.backward()
    return loss.detach()


for _ in range(3):
    for p in params: p.grad = None
    step()
    for o in opts: o.step()
for p in params: p.grad = None
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    ls = step()
for _ in range(5):
    g.replay()
    for o in opts: o.step()
torch.cuda.synchronize()
print("ok", float(ls))
