"""the per-render host stall (waiting for the device-side intersection count) is optional.  With
`deferred_size_check` the lists are sized from the previous call, every kernel checks the count on the device, and the
host looks at it one call later - so a whole render, forward + backward, is capturable in a HIP graph."""
import pytest
import torch

from deblur4dgs_amd.synth import make_scene

pytestmark = pytest.mark.gpu
NAMES = ("means", "quats", "scales", "opacities", "colors", "motion_coefs", "rots", "transls", "times", "RTs", "viewmat")


def _leaves(sc, dev, scale_add=0.0):
    L = {k: sc[k].to(dev).clone() for k in NAMES}
    L["scales"] = L["scales"] + scale_add
    return {k: v.requires_grad_() for k, v in L.items()}


def _step(L, K, W, H, w, **kw):
    from deblur4dgs_amd.exposure import render_exposure

    res = render_exposure(L["means"], L["quats"], L["scales"], L["opacities"], L["colors"], 3, L["motion_coefs"], L["rots"],
                          L["transls"], L["times"], L["RTs"], L["viewmat"], K, W, H, background=torch.ones(3, device=K.device),
                          return_depth=True, **kw)
    (res["blended"] * w).sum().backward()
    return res


def test_deferred_size_check_equals_the_synchronous_path_and_reports_overflow_one_call_later():
    from deblur4dgs_amd import engine

    dev = torch.device("cuda:0")
    N, G, K_, S, W, H = 6000, 4000, 4, 3, 160, 96
    sc = make_scene(N, G, K_, S, W, H, seed=8)
    K = sc["K"].to(dev)
    w = torch.randn(H, W, 4, generator=torch.Generator().manual_seed(0)).to(dev)
    engine._SIZE_GUESS.clear()
    ref = _leaves(sc, dev)
    r0 = _step(ref, K, W, H, w)  # synchronous: exact sizes, leaves a guess
    before = dict(engine._SIZE_STATS)
    got = _leaves(sc, dev)
    r1 = _step(got, K, W, H, w, deferred_size_check=True)
    engine.check_deferred()  # the count arrives, fits: no error
    assert torch.equal(r0["blended"], r1["blended"]) and all(torch.equal(ref[k].grad, got[k].grad) for k in NAMES)
    assert engine._SIZE_STATS["relaunched"] == before["relaunched"]
    # same shape, 8x larger splats: the guess is far too small -> the kernels skip the work, the NEXT look at the count raises
    big = _leaves(sc, dev, scale_add=2.1)
    _step(big, K, W, H, w, deferred_size_check=True)
    with pytest.raises(RuntimeError, match="INVALID"):
        engine.check_deferred()
    again = _leaves(sc, dev, scale_add=2.1)  # the guess was updated from the real count: the re-run fits
    r2 = _step(again, K, W, H, w, deferred_size_check=True)
    engine.check_deferred()
    engine._SIZE_GUESS.clear()
    sync = _leaves(sc, dev, scale_add=2.1)
    r3 = _step(sync, K, W, H, w)
    assert torch.equal(r2["blended"], r3["blended"]) and torch.equal(again["means"].grad, sync["means"].grad)


def test_every_deferred_render_of_a_step_is_verified_not_only_the_last():
    """a step that issues several same-shape renders before any count has landed used to keep only
    the newest pending record, so an EARLIER render that overflowed went unnoticed (garbage images, zero gradients).
    Three renders of one shape, the first one overflowing, one check_deferred() at the end: it must raise; and the
    overflowed render's gather must not fault on the never-written offsets (k_gather)."""
    from deblur4dgs_amd import engine

    dev = torch.device("cuda:0")
    N, G, K_, S, W, H = 6000, 4000, 4, 3, 160, 96
    sc = make_scene(N, G, K_, S, W, H, seed=8)
    K = sc["K"].to(dev)
    w = torch.randn(H, W, 4, generator=torch.Generator().manual_seed(0)).to(dev)
    engine.check_deferred()
    engine._SIZE_GUESS.clear()
    _step(_leaves(sc, dev), K, W, H, w)  # synchronous warm-up: leaves the guess of the small scene
    big, small1, small2 = _leaves(sc, dev, scale_add=2.1), _leaves(sc, dev), _leaves(sc, dev)
    key = next(iter(engine._SIZE_GUESS))
    guess = engine._SIZE_GUESS[key]
    _step(big, K, W, H, w, deferred_size_check=True)     # overflows the guess
    engine._SIZE_GUESS[key] = guess                       # (a poll between the renders may have raised the guess already)
    try:
        _step(small1, K, W, H, w, deferred_size_check=True)
        _step(small2, K, W, H, w, deferred_size_check=True)
        raised = False
    except RuntimeError as e:  # the count of `big` may already have landed at one of the later calls: also fine
        raised = "INVALID" in str(e)
    if not raised:
        with pytest.raises(RuntimeError, match="INVALID"):
            engine.check_deferred()
    engine.check_deferred()  # drained: nothing left to complain about
    torch.cuda.synchronize()
    assert float(big["means"].grad.abs().sum()) == 0.0  # the overflowed render's gradients are zeros, not a fault


def test_whole_render_forward_and_backward_replays_from_a_hip_graph():
    dev = torch.device("cuda:0")
    N, G, K_, S, W, H = 20000, 12000, 4, 4, 256, 144
    sc = make_scene(N, G, K_, S, W, H, seed=9)
    K = sc["K"].to(dev)
    w = torch.randn(H, W, 4, generator=torch.Generator().manual_seed(1)).to(dev)
    eager = _leaves(sc, dev)
    r_e = _step(eager, K, W, H, w)  # also the warm-up that leaves the size guess
    static = _leaves(sc, dev)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):  # torch's capture protocol: a few eager iterations on a side stream first
        for _ in range(2):
            for v in static.values():
                v.grad = None
            _step(static, K, W, H, w, deferred_size_check=True)
    torch.cuda.current_stream().wait_stream(side)
    for v in static.values():
        v.grad = None
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        r_g = _step(static, K, W, H, w, deferred_size_check=True)
    for it in range(3):
        with torch.no_grad():  # new parameter values in the SAME storage: the graph re-reads them
            static["means"].copy_(sc["means"].to(dev) + 0.01 * it)
            eager["means"].copy_(sc["means"].to(dev) + 0.01 * it)
        graph.replay()
        for v in eager.values():
            v.grad = None
        r_e = _step(eager, K, W, H, w)
        torch.cuda.synchronize()
        assert torch.equal(r_g["blended"], r_e["blended"]), it
        for k in NAMES:
            assert torch.equal(static[k].grad, eager[k].grad), (it, k)


def test_a_replayed_graph_reports_lists_that_outgrew_their_capture():
    """a captured step keeps the list capacities of its capture and its kernels skip their work when the device-side
    count exceeds them.  engine.GraphWatch puts the copy of the counts INTO the graph; check() before the next replay raises
    when the scene has outgrown the capture (and updates the size guess), and stays silent while it fits."""
    from deblur4dgs_amd import engine

    dev = torch.device("cuda:0")
    N, G, K_, S, W, H = 6000, 3000, 3, 2, 128, 96
    sc = make_scene(N, G, K_, S, W, H, seed=21)
    K = sc["K"].to(dev)
    w = torch.randn(H, W, 4, generator=torch.Generator().manual_seed(2)).to(dev)
    static = _leaves(sc, dev)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            for v in static.values():
                v.grad = None
            _step(static, K, W, H, w, deferred_size_check=True)
    torch.cuda.current_stream().wait_stream(side)
    engine.check_deferred()
    for v in static.values():
        v.grad = None
    graph, watch = torch.cuda.CUDAGraph(), engine.GraphWatch()
    with watch.capturing(), torch.cuda.graph(graph):
        _step(static, K, W, H, w, deferred_size_check=True)
    assert len(watch.recs) == 1
    graph.replay()
    watch.replayed()
    watch.check()  # fits: silent
    n0 = int(watch.recs[0][1][0])
    assert 0 < n0 <= watch.recs[0][2]
    with torch.no_grad():  # the scene grows under the captured graph: 12x larger splats -> far more intersections
        static["scales"].add_(2.5)
    graph.replay()
    watch.replayed()
    torch.cuda.synchronize()
    assert int(watch.recs[0][1][0]) > watch.recs[0][2], (n0, watch.recs[0][1].tolist(), watch.recs[0][2])
    with pytest.raises(RuntimeError, match="replayed from a HIP graph needed"):
        watch.check()
    key = watch.recs[0][0]
    assert engine._guess_get(key)[0] > watch.recs[0][2]  # the next eager / captured step is sized for the new count
