"""Diagnostic for tests/test_gpu_refdefault_fullsize.py: which element of `_current_xys[s].grad` / bg.* is still off after the cotangents
were zeroed on the fragile pixels - which Gaussian, where on screen, and what the oracle's margins say there."""
import os, sys, json, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch.set_num_threads(16)
from deblur4dgs_amd.synth import make_scene
from deblur4dgs_amd.scene_model import GaussianParams, MotionBases, SceneModel
from oracle import camera as ocam, cref, margins, scene as oscene
N, G, K, W, H, SEED = 140_000, 40_000, 20, 512, 288, 1010
dev = torch.device("cuda:0")
sc = make_scene(N, G, K, 11, W, H, seed=SEED, dtype=torch.float32)
keys = ("means", "quats", "scales", "colors", "opacities")
fgp = GaussianParams(*[sc[k][:G].clone() for k in keys], motion_coefs=sc["motion_coefs"].clone())
bgp = GaussianParams(*[sc[k][G:].clone() for k in keys])
model = SceneModel(sc["K"][None].clone(), sc["viewmat"][None].clone(), fgp, MotionBases(sc["rots"].clone(), sc["transls"].clone()), bgp).to(dev)
torch.manual_seed(SEED)
with torch.no_grad():
    for head in (model.move_model.RT_head0, model.move_model.RT_head1):
        head[-1].bias.copy_(0.004 * torch.randn(6))
    model.move_model.time_params.copy_(torch.tensor([[0.5, 0.3, 0.45, 0.6, 0.2, 0.5, 0.7, 0.5]]))
t = 3.0
tt = torch.tensor([1.0, 2.5, 4.0, 6.0])
g = torch.Generator().manual_seed(3)
tw = torch.cat([ocam.se3_to_SE3(0.01 * torch.randn(4, 6, generator=g)), torch.tensor([0, 0, 0, 1.0]).expand(4, 1, 4)], 1)
oscene.raster.rasterization = cref.rasterization_torch
dd = lambda x: x.detach().double().cpu().clone().requires_grad_()
fg = {k: dd(v) for k, v in model.fg.params.items()}
bg = {k: dd(v) for k, v in model.bg.params.items()}
bases = {k: dd(v) for k, v in model.motion_bases.params.items()}
sd = {k: v.detach().cpu().double().requires_grad_() for k, v in model.move_model.state_dict().items()}
w2c = sc["viewmat"].double()
RTs, times, dT = ocam.forward_start_end_mid(sd, w2c[:3, :3], w2c[:3, 3:4], t, 11, "second")
ref = oscene.render_exposure(fg, bg, bases, times[0].double(), RTs.double(), w2c, sc["K"].double(), (W, H), bg_color=1.0,
                             return_depth=True, return_mask=True, target_ts=tt.double(), target_w2cs=tw.double())
ws = {k: torch.randn(ref[k].shape, generator=g) for k in ("img", "mask", "depth", "tracks_3d", "acc")}
raw = torch.stack(ref["raw_renders"], 0)[:, 0].detach()
stack = torch.cat([raw[:-1], raw.mean(0, keepdim=True)], 0)
ms, toggles = [], torch.zeros(H, W, dtype=torch.bool)
for s in range(11):
    inf = ref["info"][s]
    m_, q_, s_, o_ = inf["inputs"]
    ms.append(margins.pixel_margins(inf["means2d"], inf["conics"], o_, inf["depths"], inf["flatten_ids"], inf["isect_offsets"], W, H))
    toggles |= margins.gaussian_toggle_mask(m_, q_, s_, o_, w2c, sc["K"].double(), W, H, eps_px=32 * 6e-8 * 512)[0]
eps = float(os.environ.get("EPS", "1e-4"))
F = toggles | margins.blend_tie_mask(stack, eps=eps)
for mm in ms:
    F = F | margins.fragile_pixels(mm, eps)
keep = ~F
out = model.render(t, sc["viewmat"][None].to(dev), sc["K"][None].to(dev), (W, H), target_ts=tt.to(dev), target_w2cs=tw.to(dev),
                   return_depth=True, return_mask=True, mode="blury", stage="second")
shp = lambda x, k: k.view(1, H, W, *([1] * (x.dim() - 3)))
sum((out[k] * ws[k].to(dev) * shp(out[k], keep.to(dev).float())).sum() for k in ws).backward()
sum((ref[k] * ws[k].double() * shp(ref[k], keep.double())).sum() for k in ws).backward()
torch.cuda.synchronize()
rep = {}
exp_dev = out["exposure_imgs"][:, 0].detach().cpu().double()   # [S,H,W,17]; [-1] = blended
for s in (0, 10):
    got = model._current_xys[s].grad[0].cpu().double()
    want = ref["info"][s]["v_means2d"]
    err = (got - want).abs().max(-1)[0]
    top = torch.topk(err, 5)
    scale = float(want.abs().max())
    inf = ref["info"][s]
    rows = []
    for e, i in zip(top.values.tolist(), top.indices.tolist()):
        mx, my = inf["means2d"][i].tolist()
        px, py = int(mx), int(my)
        y0, y1, x0, x1 = max(py - 3, 0), min(py + 4, H), max(px - 3, 0), min(px + 4, W)
        img_err = None
        if s < 10:
            img_err = float((exp_dev[s, y0:y1, x0:x1] - raw[s, y0:y1, x0:x1]).abs().max() / raw[s].abs().max())
        rows.append(dict(gid=i, is_fg=i < G, rel_err=e / scale, got=got[i].tolist(), want=want[i].tolist(), means2d=[mx, my], radius=int(inf["radii"][i]),
                         depth=float(inf["depths"][i]), opacity=float(inf["inputs"][3][i]), conic=inf["conics"][i].tolist(),
                         fragile_near=bool(F[y0:y1, x0:x1].any()), n_fragile_near=int(F[y0:y1, x0:x1].sum()),
                         margins_near={k: float(ms[s][k][y0:y1, x0:x1].min()) for k in ("alpha", "T", "clamp", "order")},
                         subsample_image_err_near=img_err))
    rep[f"xys[{s}]"] = rows
for name, gotp, refp in (("bg.means", model.bg.params["means"], bg["means"]), ("bg.opacities", model.bg.params["opacities"], bg["opacities"]),
                         ("fg.means", model.fg.params["means"], fg["means"])):
    got, want = gotp.grad.cpu().double(), refp.grad
    err = (got - want).abs().reshape(got.shape[0], -1).max(-1)[0]
    top = torch.topk(err, 5)
    rep[name] = [dict(idx=i, rel_err=e / float(want.abs().max()), got=got[i].tolist(), want=want[i].tolist()) for e, i in zip(top.values.tolist(), top.indices.tolist())]
# per-sub-sample image errors outside F
for s in range(10):
    d = (exp_dev[s] - raw[s]).abs().max(-1)[0] / raw[s].abs().max()
    bad = d > 1e-4
    rep.setdefault("subsample_image", []).append(dict(s=s, bad_px=int(bad.sum()), bad_px_outside_F=int((bad & ~F).sum()), worst=float(d.max()),
                                                       worst_outside_F=float(d[~F].max()), where_outside=[int(x) for x in divmod(int(torch.where(~F, d, torch.zeros_like(d)).argmax()), W)]))
rep["fragile_fraction"] = float(F.float().mean())
os.makedirs("gpurun_out", exist_ok=True)
json.dump(rep, open("gpurun_out/diag_flip_refdefault.json", "w"), indent=1)
print(json.dumps(rep)[:3000])
