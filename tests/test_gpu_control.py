"""SURVEY 8f-1 on the device: the compaction kernels behind the control steps (d4gs_control_plan / d4gs_gather_rows)
against golden vectors produced by the reference's own `GaussianParams.densify_params / cull_params` (tests/golden F6,
bit-exact), and the whole densify / cull step - parameters, Adam moments, running statistics - against the same step
run on CPU tensors (the host-logic path pinned by tests/test_control_cpu.py)."""
import copy
import os

import numpy as np
import pytest
import torch

from deblur4dgs_amd import control
from deblur4dgs_amd.rows import RowPlan
from deblur4dgs_amd.scene_model import GaussianParams
from tests.test_control_cpu import NAMES, _model, _optimizers

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_device_row_surgery_matches_reference_golden(golden_dir):
    z = np.load(os.path.join(golden_dir, "f6_control_params.npz"))
    for c in range(int(z["n_cases"])):
        split, dup, cull = (torch.tensor(z[f"c{c}_{m}"]).to(DEV) for m in ("split", "dup", "cull"))
        for op, run in (("densify", lambda g: g.densify_params(split, dup)), ("cull", lambda g: g.cull_params(cull))):
            raw = {k: torch.tensor(z[f"c{c}_in_{k}"]).to(DEV) for k in NAMES if f"c{c}_in_{k}" in z.files}
            gp = GaussianParams(raw["means"], raw["quats"], raw["scales"], raw["colors"], raw["opacities"], raw.get("motion_coefs"))
            out = run(gp)
            want = {k[len(f"c{c}_{op}_"):]: z[k] for k in z.files if k.startswith(f"c{c}_{op}_")}
            assert set(out) == set(want)
            for k, v in out.items():
                assert v.is_cuda and isinstance(v, torch.nn.Parameter) and gp.params[k] is v
                np.testing.assert_array_equal(v.detach().cpu().numpy(), want[k])  # bit-exact, incl. scales - log 1.6


@pytest.mark.parametrize("n,p_split,p_dup", [(1, 1.0, 0.0), (1, 0.0, 0.0), (1023, 0.3, 0.3), (1024, 0.0, 1.0), (1025, 1.0, 0.0),
                                             (70_001, 0.2, 0.5), (300_000, 0.01, 0.02)])
def test_plans_of_any_size_match_torch_indexing(n, p_split, p_dup):
    g = torch.Generator().manual_seed(n)
    split = torch.rand(n, generator=g) < p_split
    dup = (torch.rand(n, generator=g) < p_dup) & ~split
    x = torch.randn(n, 3, generator=g)
    v = torch.randint(0, 1 << 40, (n,), generator=g)  # int64 rows (vis_count)
    for flags in ((split, dup), (split, None)):
        cpu = RowPlan(*flags)
        dev = RowPlan(*[None if f is None else f.to(DEV) for f in flags])
        assert (cpu.n_keep, cpu.n_dup, cpu.n_split, cpu.n_out) == (dev.n_keep, dev.n_dup, dev.n_split, dev.n_out)
        assert torch.equal(dev.src[: dev.n_out].cpu(), cpu.src)
        for kw in ({}, {"zero_new": True}, {"split_add": -0.47}):
            assert torch.equal(dev.gather(x.to(DEV), **kw).cpu(), cpu.gather(x, **kw))
        assert torch.equal(dev.gather(v.to(DEV)).cpu(), cpu.gather(v))
        assert torch.equal(dev.gather(x[:, 0].contiguous().to(DEV)).cpu(), cpu.gather(x[:, 0].contiguous()))


def test_densify_and_cull_steps_on_the_device_equal_the_host_logic():
    cfg = control.ControlCfg()
    res = {}
    for where in ("cpu", DEV):
        model = _model(n_fg=300, n_bg=200, seed=3)
        opts = _optimizers(model)  # two Adam steps on CPU, then everything moves
        if where != "cpu":
            state = {k: copy.deepcopy(o.state_dict()) for k, o in opts.items()}
            model = model.to(where)
            opts = {}
            for part in ("fg", "bg"):
                for name, p in getattr(model, part).params.items():
                    o = torch.optim.Adam([p], lr=1e-3)
                    o.load_state_dict(state[f"{part}.params.{name}"])
                    opts[f"{part}.params.{name}"] = o
        N = model.num_gaussians
        stats = control.new_running_stats(N, where)
        stats["vis_count"] += 4
        stats["xys_grad_norm_acc"] = torch.linspace(0, 0.004, N).to(where)
        stats["max_radii"] = torch.linspace(0, 0.2, N).to(where)
        with torch.no_grad():
            model.fg.params["scales"][::2] = 0.0
            model.bg.params["opacities"][::3] = -5.0
        n_sd = control.densify_step(model, stats, opts, cfg, global_step=500)
        n_c = control.cull_step(model, stats, opts, cfg, global_step=3500)
        snap = {f"{part}.{k}": v.detach().cpu() for part in ("fg", "bg") for k, v in getattr(model, part).params.items()}
        for k, o in opts.items():
            p = o.param_groups[0]["params"][0]
            snap[f"adam.{k}.m"], snap[f"adam.{k}.v"] = o.state[p]["exp_avg"].cpu(), o.state[p]["exp_avg_sq"].cpu()
            assert o.state[p]["exp_avg"].shape == p.shape
        snap.update({f"stats.{k}": v.cpu() for k, v in stats.items()})
        res[where] = (n_sd, n_c, snap)
    assert res["cpu"][:2] == res[DEV][:2] and res["cpu"][0][0] > 0 and res["cpu"][0][1] > 0 and res["cpu"][1] > 0
    for k, v in res["cpu"][2].items():
        assert torch.equal(v, res[DEV][2][k]), k


def test_spatial_order_step_leaves_the_render_unchanged_on_the_device():
    """control.spatial_order_step on device tensors (d4gs_gather_rows behind RowPlan.from_permutation): the same scene in another
    memory order renders the same sub-sample images bit for bit (the composite orders by depth; distinct depths), the optimizers
    keep stepping, and the per-Gaussian gradients are the permuted old ones."""
    from tests.test_gpu_scene_model import _build

    W, H = 128, 96
    model, sc = _build(5000, 3000, 4, W, H, 77, torch.device(DEV))
    dev = torch.device(DEV)
    w2c, Km = sc["viewmat"].to(dev)[None], sc["K"].to(dev)[None]

    def render():
        out = model.render(3.0, w2c, Km, (W, H), return_depth=True, mode="blury")  # all 11 sub-samples
        return out["img"], out["exposure_imgs"]

    img0, stack0 = render()
    (img0.sum()).backward()
    g_old = {p: {k: v.grad.clone() for k, v in getattr(model, p).params.items()} for p in ("fg", "bg") if getattr(model, p) is not None}
    for p in g_old:
        for v in getattr(model, p).params.values():
            v.grad = None
    opts = {}
    plans = control.spatial_order_step(model, None, opts)
    img1, stack1 = render()
    (img1.sum()).backward()
    torch.cuda.synchronize()
    # raw sub-sample renders (the last slot holds the blended frame): the composite orders by depth, so storage order cannot matter
    # - except between two splats of one tile with bit-equal fp32 depths, which are ordered by index (none expected at this size)
    diff = (stack0[:-1] - stack1[:-1]).abs()
    assert float((diff > 0).float().mean()) < 1e-4 and float(diff.max()) < 1e-3, (float((diff > 0).float().mean()), float(diff.max()))
    assert torch.allclose(img0, img1, rtol=0, atol=1e-5)
    for p, plan in plans.items():
        perm = plan.src.long()
        for k, v in getattr(model, p).params.items():
            assert torch.allclose(v.grad, g_old[p][k][perm], rtol=1e-4, atol=1e-5 * float(g_old[p][k].abs().max())), (p, k)
