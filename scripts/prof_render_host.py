"""cProfile of one SceneModel.render + backward on a tiny scene (host-side cost only)."""
import cProfile, pstats, io, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib.util
spec = importlib.util.spec_from_file_location("ex", os.path.join(os.path.dirname(__file__), "..", "examples", "train_dynamic_step.py"))
ex = importlib.util.module_from_spec(spec); spec.loader.exec_module(ex)
dev = "cuda:0"; W, H = 64, 48
model, sc = ex.build(n_fg=1500, n_bg=1500, K=20, W=W, H=H, dev=dev)
w2c, K = sc["viewmat"][None].to(dev), sc["K"][None].to(dev)
tt = torch.tensor([1.0, 2.0, 4.0, 5.0], device=dev); tw = w2c.expand(4, 4, 4).contiguous()
def step():
    o = model.render(3, w2c, K, (W, H), target_ts=tt, target_w2cs=tw, return_depth=True, return_mask=True, mode="blury")
    (o["img"].sum() + o["tracks_3d"].sum()).backward()
for _ in range(10): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(50): step()
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:7000])
