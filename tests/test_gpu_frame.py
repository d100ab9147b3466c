"""The one-call entry points d4gs_forward / d4gs_backward behind ONE autograd node (engine.FrameFn) against the
staged chain (d4gs_project_fwd -> d4gs_bin_sort -> d4gs_raster_fwd -> d4gs_blend_fwd and back; ProjectFn / RasterFn /
BlendFn): the same kernels in the same order, so every output and every gradient must be BITWISE equal - which puts the
one-call path under all the oracle parity the staged chain has."""
import pytest
import torch

from deblur4dgs_amd.synth import make_scene

pytestmark = pytest.mark.gpu
NAMES = ("means", "quats", "scales", "opacities", "colors", "motion_coefs", "rots", "transls", "times", "RTs", "viewmat")


def _leaves(sc, dev):
    return {k: (sc[k].to(dev).clone().requires_grad_() if k in sc and sc[k] is not None else None) for k in NAMES}


def _render(L, K, W, H, fused, blend=True, D=3, **kw):
    from deblur4dgs_amd.exposure import render_exposure

    cols = L["colors"] if D == 3 else torch.cat([L["colors"], L["means"].repeat(1, 5)], -1)[:, :D]
    return render_exposure(L["means"], L["quats"], L["scales"], L["opacities"], cols, 3, L["motion_coefs"], L["rots"],
                           L["transls"], L["times"], L["RTs"], L["viewmat"], K, W, H, background=torch.linspace(0.2, 0.9, D).to(K.device),
                           return_depth=True, blend=blend, fused=fused, **kw)


@pytest.mark.parametrize("sub_losses", [True, False])
@pytest.mark.parametrize("N,G,K_,S,W,H,D", [(5000, 3000, 4, 3, 160, 96, 3), (800, 0, 1, 1, 64, 48, 3), (3000, 3000, 12, 5, 96, 64, 16)])
def test_one_call_path_equals_the_staged_chain_bitwise(N, G, K_, S, W, H, D, sub_losses):
    """sub_losses=False: gradients arrive on the blended frame and its accumulation only - d4gs_backward then folds the blend's adjoint
    into the composite backward's prologue (renders of <= 5 colour channels; the max / min channel's winner comes from k_blend_fwd's
    map) instead of launching k_blend_bwd, which the staged chain always does: the two must still agree bit for bit."""
    dev = torch.device("cuda:0")
    sc = make_scene(N, G, max(K_, 1), S, W, H, seed=31)
    K = sc["K"].to(dev)
    g = torch.Generator().manual_seed(7)
    wb, wa = torch.randn(H, W, D + 1, generator=g).to(dev), torch.randn(H, W, generator=g).to(dev)
    ws, wsa = torch.randn(S, H, W, D + 1, generator=g).to(dev), torch.randn(S, H, W, 1, generator=g).to(dev)
    out = {}
    for fused in (False, True):
        L = _leaves(sc, dev)
        if G == 0:
            L["motion_coefs"] = L["rots"] = L["transls"] = L["times"] = None  # a static scene (BASELINE cfg1's shape)
        r = _render(L, K, W, H, fused, D=D)
        assert bool(r["state"].frame_io) == fused
        # losses on the blurry frame, its accumulation AND the per-sub-sample images (flow3d/trainer.py:575-618)
        loss = (r["blended"] * wb).sum() + (r["acc"] * wa).sum()
        if sub_losses:
            loss = loss + (r["renders"] * ws).sum() + (r["alphas"] * wsa).sum()
        xys = None
        if fused:
            xys = [r["means2d"][s:s + 1].detach().requires_grad_() for s in range(S)]
            r["state"].xys_sink = xys
        else:
            r["means2d"].retain_grad()
        loss.backward()
        torch.cuda.synchronize()
        m2g = torch.cat([x.grad for x in xys], 0) if fused else r["means2d"].grad
        out[fused] = dict(blended=r["blended"], acc=r["acc"], renders=r["renders"], alphas=r["alphas"], means2d=r["means2d"],
                          radii=r["radii"], m2g=m2g, **{f"g_{k}": v.grad for k, v in L.items() if v is not None})
    for k, a in out[False].items():
        b = out[True][k]
        assert a is not None and b is not None and torch.equal(a.detach(), b.detach()), k
    assert float(out[True]["g_means"].abs().max()) > 0 and float(out[True]["m2g"].abs().max()) > 0


@pytest.mark.parametrize("fused", [True, False])
def test_exact_tiles_through_the_frame_paths(fused, monkeypatch):
    """D4GS_EXACT_TILES through d4gs_forward / d4gs_backward (the mask buffer lives in the frame workspace) and through the staged
    chain, on a dynamic S = 3 scene of splats a few tiles wide: blended frame, sub-sample images, radii and every leaf gradient are
    BITWISE those of whole rectangles; the lists shrink.  Then the `auto` rule: on from 3 intersections per instance (the previous
    render's count), with hysteresis, and the same bits again."""
    from deblur4dgs_amd import engine

    monkeypatch.setenv("D4GS_SEG", "0")
    dev = torch.device("cuda:0")
    N, G, K_, S, W, H = 9000, 6000, 4, 3, 1024, 576  # (3 x 2304 tiles: beyond the launches that run depth segments - `auto` never turns the test on there)
    sc = make_scene(N, G, K_, S, W, H, seed=35)
    sc["scales"] = sc["scales"] + 1.3  # exp(1.3) = 3.7 x at twice cfg2's focal length: rectangles of 2 x 2 ... 4 x 4 tiles
    K = sc["K"].to(dev)
    g = torch.Generator().manual_seed(7)
    wb, wa = torch.randn(H, W, 4, generator=g).to(dev), torch.randn(H, W, generator=g).to(dev)
    out = {}
    for xt in (False, True, None):
        if xt is None:  # auto: the previous renders of this shape left ~4 intersections per instance -> on
            monkeypatch.setattr(engine, "EXACT_TILES", "auto")
            monkeypatch.setattr(engine, "EXACT_TILES_FROM", 1.5)  # (the shape's measured count per instance is checked below)
            monkeypatch.setattr(engine, "EXACT_TILES_MIN_LIVE", 0.0)  # (and whatever share of its list entries is alive)
        L = _leaves(sc, dev)
        r = _render(L, K, W, H, fused, exact_tiles=xt)
        ((r["blended"] * wb).sum() + (r["acc"] * wa).sum()).backward()
        torch.cuda.synchronize()
        out[xt] = dict(blended=r["blended"], renders=r["renders"], alphas=r["alphas"], radii=r["radii"], n=r["state"].n_isect,
                       on=r["state"].cfg.exact_tiles, **{f"g_{k}": v.grad for k, v in L.items() if v is not None})
    assert out[False]["on"] is False and out[True]["on"] is True and out[None]["on"] is True
    assert out[False]["n"] > 2.0 * S * N and out[True]["n"] < 0.93 * out[False]["n"] and out[None]["n"] == out[True]["n"]
    for k, a in out[False].items():
        if k in ("n", "on"):
            continue
        for other in (True, None):
            assert torch.equal(a.detach(), out[other][k].detach()), (k, other)


@pytest.mark.parametrize("fused", [False, True])
def test_exact_tiles_auto_turning_off_under_deferred_size_check_counts_again(fused, monkeypatch):
    """`auto` had turned the exact-tiles test on, the shape's list capacity was measured with it (shorter lists);
    then the live fraction drops and `auto` turns it off.  A deferred render used to launch at that stale capacity: its kernels
    skipped the work (invalid frame) and a later poll raised.  Now the capacity is keyed on the flag and the flip leaves no
    guess, so the render counts first: same bits as the plain render, nothing to complain about afterwards."""
    from deblur4dgs_amd import engine

    monkeypatch.setenv("D4GS_SEG", "0")
    dev = torch.device("cuda:0")
    N, G, K_, S, W, H = 9000, 6000, 4, 3, 1024, 576  # (beyond the launches that run depth segments: `auto` is never on there)
    sc = make_scene(N, G, K_, S, W, H, seed=36)
    sc["scales"] = sc["scales"] + 1.3
    K = sc["K"].to(dev)
    g = torch.Generator().manual_seed(9)
    wb, wa = torch.randn(H, W, 4, generator=g).to(dev), torch.randn(H, W, generator=g).to(dev)

    def step(**kw):
        L = _leaves(sc, dev)
        r = _render(L, K, W, H, fused, **kw)
        ((r["blended"] * wb).sum() + (r["acc"] * wa).sum()).backward()
        return r, L

    engine.check_deferred()
    want, Lw = step(exact_tiles=False)
    torch.cuda.synchronize()
    n_off = want["state"].n_isect
    monkeypatch.setattr(engine, "EXACT_TILES", "auto")
    monkeypatch.setattr(engine, "EXACT_TILES_FROM", 1.5)
    monkeypatch.setattr(engine, "EXACT_TILES_MIN_LIVE", 0.0)
    monkeypatch.setattr(engine, "EXACT_TILES_LIVE_HYST", 0.0)
    base = engine._size_key(dev, S, N, W, H)
    engine._XT_ON.pop(base, None)
    r1, _ = step(deferred_size_check=True)   # auto: on (the rectangles' count per instance), counts its own lists
    r2, _ = step(deferred_size_check=True)   # launched at the capacity measured with the flag on
    engine.check_deferred()
    assert r1["state"].cfg.exact_tiles is True and r2["state"].cfg.exact_tiles is True
    assert engine._guess_get(base + (True,))[0] < 1.25 * n_off + 4096  # the on-capacity: sized for the shorter lists
    # whatever the off key holds dates from before the flag came on (here: a scene half the size) - it must not size the next render
    engine._guess_put(base + (False,), (n_off // 2, 512))
    monkeypatch.setattr(engine, "EXACT_TILES_MIN_LIVE", 2.0)  # "the live fraction dropped": auto turns the test off
    r3, L3 = step(deferred_size_check=True)
    engine.check_deferred()                   # used to raise: "... needed n intersections but its lists were sized for ..."
    torch.cuda.synchronize()
    assert r3["state"].cfg.exact_tiles is False and r3["state"].n_isect == n_off
    assert torch.equal(r3["blended"], want["blended"]) and torch.equal(r3["renders"], want["renders"])
    for k in NAMES:
        assert torch.equal(L3[k].grad, Lw[k].grad), k
    r4, L4 = step(deferred_size_check=True)   # and the next one runs deferred again, at the off-capacity
    engine.check_deferred()
    assert torch.equal(r4["blended"], want["blended"]) and torch.equal(L4["means"].grad, Lw["means"].grad)
    engine._XT_ON.pop(base, None)


def test_one_call_path_unblended_and_size_protocol():
    """blend=False (what exposure sharding renders), a cold size guess (the counting call), a warm one, and an overflowing
    deferred one - the protocol is the staged chain's."""
    from deblur4dgs_amd import engine

    dev = torch.device("cuda:0")
    N, G, K_, S, W, H = 4000, 2500, 3, 2, 128, 80
    sc = make_scene(N, G, K_, S, W, H, seed=33)
    K = sc["K"].to(dev)
    w = torch.randn(S, H, W, 4, generator=torch.Generator().manual_seed(1)).to(dev)
    ref = _leaves(sc, dev)
    r0 = _render(ref, K, W, H, False, blend=False)
    (r0["renders"] * w).sum().backward()
    engine.check_deferred()
    engine._SIZE_GUESS.clear()
    for it in range(2):  # cold (counting call + exact launch), then warm (one launch, checked afterwards)
        got = _leaves(sc, dev)
        r1 = _render(got, K, W, H, True, blend=False)
        assert r1["blended"] is None
        (r1["renders"] * w).sum().backward()
        torch.cuda.synchronize()
        assert torch.equal(r0["renders"], r1["renders"]) and all(torch.equal(ref[k].grad, got[k].grad) for k in NAMES), it
    big = _leaves(sc, dev)
    with torch.no_grad():
        big["scales"] += 2.0
    rb = _render(big, K, W, H, True, blend=False, deferred_size_check=True)  # the guess is far too small
    (rb["renders"] * w).sum().backward()
    with pytest.raises(RuntimeError, match="INVALID"):
        engine.check_deferred()
    torch.cuda.synchronize()
    assert float(big["means"].grad.abs().sum()) == 0.0  # an overflowed render's gradients are zeros


def test_scene_model_uses_the_one_call_path_and_keeps_the_side_channels():
    from tests.test_gpu_scene_model import _build

    dev = torch.device("cuda:0")
    N, G, K_, W, H = 900, 500, 4, 64, 48
    res = {}
    for fused in (True, False):
        torch.manual_seed(123)  # (MoveModel's random initialisation happens inside _build, before its own seeding)
        model, sc = _build(N, G, K_, W, H, 17, dev)
        model.fused = fused
        out = model.render(3.0, sc["viewmat"][None].to(dev), sc["K"][None].to(dev), (W, H), return_depth=True, return_mask=True,
                           mode="blury")
        loss = out["img"].square().sum() + out["depth"].sum() + out["exposure_imgs"][2].sum() + out["acc"].sum()
        loss.backward()
        torch.cuda.synchronize()
        res[fused] = dict(img=out["img"], depth=out["depth"], exp=out["exposure_imgs"],
                          xys=torch.cat([x.grad for x in model._current_xys], 0), radii=torch.cat(model._current_radii, 0),
                          **{n: p.grad for n, p in model.named_parameters() if p.grad is not None})
    assert set(res[True]) == set(res[False]) and "move_model.RT_head0.2.bias" in res[True]
    for k in res[True]:
        assert torch.equal(res[True][k].detach(), res[False][k].detach()), k


def test_no_grad_render_takes_the_forward_workspace_only_and_the_cold_call_only_counts():
    """(1) a render under torch.no_grad() (validation, the viewer's render_view) must not allocate - and pin through the
    returned state - the backward's scratch; (2) the cold call of a shape (no size guess yet) counts with a d4gs_forward that stops
    after the projection (capacity < 0) instead of running the whole frame at capacity 1 first.  Same image bit for bit."""
    import ctypes as C

    from deblur4dgs_amd import _lib as L
    from deblur4dgs_amd import engine

    dev = torch.device("cuda:0")
    N, G, K_, S, W, H = 6000, 4000, 3, 3, 160, 96
    sc = make_scene(N, G, K_, S, W, H, seed=35)
    K = sc["K"].to(dev)
    lv = _leaves(sc, dev)
    engine._SIZE_GUESS.clear()
    lib = L.lib()
    lib.d4gs_profile_enable(1)
    with torch.no_grad():
        r_ng = _render(lv, K, W, H, True)  # cold: counting call + exact launch
    torch.cuda.synchronize()
    lib.d4gs_profile_enable(0)
    buf = C.create_string_buffer(1 << 16)
    lib.d4gs_profile_collect(buf, C.c_size_t(len(buf)))
    launches = {ln.split()[0]: int(ln.split()[1]) for ln in buf.value.decode().splitlines()}
    n_fwd = 2 if engine.LAZY_SORT == "1" else 1  # (a suite run with D4GS_LAZY_SORT=1 emits and composites in two passes)
    assert launches["k_project_fwd"] == 2 and launches["k_raster_fwd_r"] == n_fwd and launches["k_emit"] == n_fwd, launches
    r_g = _render(lv, K, W, H, True)
    torch.cuda.synchronize()
    assert torch.equal(r_ng["blended"], r_g["blended"]) and torch.equal(r_ng["renders"], r_g["renders"])
    st_ng, st_g = r_ng["state"], r_g["state"]
    dims = st_g.cfg.dims()
    assert st_g.ws_bytes == lib.d4gs_frame_workspace_bytes(C.byref(dims), st_g.ws_cap[0])
    assert st_ng.ws_bytes == lib.d4gs_frame_workspace_bytes_fwd(C.byref(dims), st_ng.ws_cap[0])
    cap = st_ng.n_isect
    full, fwd = lib.d4gs_frame_workspace_bytes(C.byref(dims), cap), lib.d4gs_frame_workspace_bytes_fwd(C.byref(dims), cap)
    # what a no_grad render does not allocate: the gradient rows (40 B per intersection), the image-gradient stack, the
    # per-instance gradient buffers
    assert full - fwd >= 40 * cap + 4 * S * H * W * 5 + 4 * S * N * 4, (fwd, full)
    (r_g["blended"].sum()).backward()  # and the differentiable one still has its scratch
    torch.cuda.synchronize()
    assert float(lv["means"].grad.abs().sum()) > 0


def test_copy_counts_into_pinned_and_pageable_memory():
    """d4gs_copy_counts stores the four counts into PINNED host memory from a kernel (no copy node for the kernels behind it to wait
    for); a pageable buffer - which the device cannot address - still gets them, through an ordinary copy."""
    import ctypes as C

    import numpy as np

    from deblur4dgs_amd import _lib as L

    dev = torch.device("cuda:0")
    lib = L.lib()
    src = torch.tensor([123456789012, 4321, 77, 5], dtype=torch.int64, device=dev)
    pinned = torch.zeros(4, dtype=torch.int64).pin_memory()
    pageable = np.zeros(4, dtype=np.int64)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.d4gs_copy_counts(C.c_void_p(src.data_ptr()), C.c_void_p(pinned.data_ptr()), stream) == 0
    assert lib.d4gs_copy_counts(C.c_void_p(src.data_ptr()), C.c_void_p(pageable.ctypes.data), stream) == 0
    torch.cuda.synchronize()
    assert pinned.tolist() == [123456789012, 4321, 77, 5] and pageable.tolist() == [123456789012, 4321, 77, 5]
