"""The multi-GPU driver on one GPU: a world-size-1 RCCL ("nccl") group exercises exactly the code the N > 1 runs
take - collectives, flat gradient buffer, gradient arena - and must reproduce the plain single-process step.
(The N > 1 arithmetic itself is covered on CPU with gloo in tests/test_parallel_gloo.py.)"""
import os
import socket

import pytest
import torch

from deblur4dgs_amd.synth import make_scene

pytestmark = pytest.mark.gpu
NAMES = ("means", "quats", "scales", "opacities", "colors", "motion_coefs", "rots", "transls", "times", "RTs", "viewmat")


@pytest.fixture(scope="module")
def group():
    import torch.distributed as dist

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    yield dist
    dist.destroy_process_group()


@pytest.mark.parametrize("mode,need_stack", [("views", False), ("exposure", False), ("exposure", True)])
def test_sharded_step_equals_plain_step(group, mode, need_stack):
    from deblur4dgs_amd.exposure import render_exposure
    from deblur4dgs_amd.parallel import ShardedExposure

    dev = torch.device("cuda", 0)
    W, H, S = 96, 64, 4
    sc = make_scene(2500, 1500, 5, S, W, H, seed=21)
    K = sc["K"].to(dev)
    g = torch.Generator().manual_seed(2)
    wimg, wacc = torch.randn(H, W, 4, generator=g).to(dev), torch.randn(H, W, generator=g).to(dev)
    bg = torch.ones(3, device=dev)

    def leaves():
        return {k: sc[k].to(dev).clone().requires_grad_() for k in NAMES}

    ref = leaves()
    res = render_exposure(ref["means"], ref["quats"], ref["scales"], ref["opacities"], ref["colors"], 3, ref["motion_coefs"],
                          ref["rots"], ref["transls"], ref["times"], ref["RTs"], ref["viewmat"], K, W, H, background=bg,
                          return_depth=True)
    (torch.dot(res["blended"].reshape(-1), wimg.reshape(-1)) + torch.dot(res["acc"].reshape(-1), wacc.reshape(-1))).backward()

    got = leaves()
    sh = ShardedExposure(1, 0, mode=mode, need_stack=need_stack)  # exposure: reduce-based blend / gathered stack
    for _ in range(2):  # second step: the previous gradients alias the flat buffer and must not be accumulated into
        sh.step(got, K, W, H, bg, wimg, wacc)
    torch.cuda.synchronize()
    for k in NAMES:
        a, b = got[k].grad, ref[k].grad
        assert a is not None and a.data_ptr() == sh.reducer.views[k].data_ptr(), k
        tol = 1e-5 * max(1.0, float(b.abs().max()))
        assert float((a - b).abs().max()) <= tol, (k, float((a - b).abs().max()))


@pytest.mark.parametrize("mode,need_stack", [("exposure", False), ("exposure", True), ("views", False)])
def test_sharded_step_with_its_collectives_replays_from_a_hip_graph(group, mode, need_stack):
    """The whole sharded step - render, RCCL all-reduces / all-gathers, blend, backward, gradient all-reduce - captured once in a HIP
    graph (deferred size check: no host wait inside) and replayed: bitwise the eager step's gradients, also after the
    parameters have changed in place between replays."""
    from deblur4dgs_amd import engine
    from deblur4dgs_amd.parallel import ShardedExposure

    dev = torch.device("cuda", 0)
    W, H, S = 96, 64, 4
    sc = make_scene(2500, 1500, 5, S, W, H, seed=22)
    K = sc["K"].to(dev)
    g = torch.Generator().manual_seed(3)
    wimg, wacc = torch.randn(H, W, 4, generator=g).to(dev), torch.randn(H, W, generator=g).to(dev)
    bg = torch.ones(3, device=dev)
    lv = {k: sc[k].to(dev).clone().requires_grad_() for k in NAMES}
    sh = ShardedExposure(1, 0, mode=mode, need_stack=need_stack)
    sh.deferred_size_check = True

    def eager():
        sh.step(lv, K, W, H, bg, wimg, wacc)
        engine.check_deferred()
        torch.cuda.synchronize()
        return {k: lv[k].grad.clone() for k in NAMES}

    for _ in range(2):
        ref0 = eager()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        sh.step(lv, K, W, H, bg, wimg, wacc)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        sh.step(lv, K, W, H, bg, wimg, wacc)
    graph.replay()
    torch.cuda.synchronize()
    for k in NAMES:
        assert torch.equal(lv[k].grad, ref0[k]), k
    with torch.no_grad():  # an optimizer step between replays: same tensors, new values
        lv["means"].add_(0.01)
        lv["colors"].mul_(0.9)
    graph.replay()
    torch.cuda.synchronize()
    got = {k: lv[k].grad.clone() for k in NAMES}
    ref1 = eager()
    for k in NAMES:
        assert torch.equal(got[k], ref1[k]), k
    assert not torch.equal(ref1["means"], ref0["means"])


# ---- world size 2 over RCCL: runs only where two GPUs are visible (the 1-GPU test boxes skip it) -----------------
def _ws2_worker(rank, port, mode, S, need_stack, out_q):
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE="2",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda", rank)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=2, device_id=dev)
    from deblur4dgs_amd.parallel import ShardedExposure

    W, H = 96, 64
    sc = make_scene(2500, 1500, 5, S, W, H, seed=21 + (rank if mode == "views" else 0))
    g = torch.Generator().manual_seed(2)
    wimg, wacc = torch.randn(H, W, 4, generator=g).to(dev), torch.randn(H, W, generator=g).to(dev)
    leaves = {k: sc[k].to(dev).clone().requires_grad_() for k in NAMES}
    sh = ShardedExposure(2, rank, mode=mode, need_stack=need_stack)
    for _ in range(2):
        sh.step(leaves, sc["K"].to(dev), W, H, torch.ones(3, device=dev), wimg, wacc)
    torch.cuda.synchronize()
    out_q.put((rank, {k: leaves[k].grad.cpu().numpy() for k in NAMES}))  # by value: the sender exits before the receiver reads
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL over xGMI); 1-GPU boxes skip")
@pytest.mark.parametrize("mode,S,need_stack", [("exposure", 4, False), ("exposure", 4, True), ("exposure", 3, False), ("views", 4, False)])
def test_world_size_2_rccl_step_equals_the_single_process_step(mode, S, need_stack):
    """The whole ShardedExposure.step on two ranks: exposure sharding (the reduce-based blend, the gathered stack for
    need_stack, the ragged S = 3) must give every rank the single-process gradients of the same frame; view sharding the
    mean of the two views' gradients."""
    import torch.multiprocessing as mp

    from deblur4dgs_amd.exposure import render_exposure

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ws2_worker, args=(r, port, mode, S, need_stack, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    dev = torch.device("cuda", 0)
    W, H = 96, 64
    g = torch.Generator().manual_seed(2)
    wimg, wacc = torch.randn(H, W, 4, generator=g).to(dev), torch.randn(H, W, generator=g).to(dev)
    want = None
    for view in range(2 if mode == "views" else 1):
        sc = make_scene(2500, 1500, 5, S, W, H, seed=21 + view)
        L = {k: sc[k].to(dev).clone().requires_grad_() for k in NAMES}
        r = render_exposure(L["means"], L["quats"], L["scales"], L["opacities"], L["colors"], 3, L["motion_coefs"], L["rots"],
                            L["transls"], L["times"], L["RTs"], L["viewmat"], sc["K"].to(dev), W, H,
                            background=torch.ones(3, device=dev), return_depth=True)
        (torch.dot(r["blended"].reshape(-1), wimg.reshape(-1)) + torch.dot(r["acc"].reshape(-1), wacc.reshape(-1))).backward()
        gr = {k: L[k].grad.cpu() / (2.0 if mode == "views" else 1.0) for k in NAMES}
        want = gr if want is None else {k: want[k] + gr[k] for k in NAMES}
    for rank in (0, 1):
        for k in NAMES:
            tol = 1e-5 * max(1.0, float(want[k].abs().max()))
            assert float((torch.from_numpy(res[rank][k]) - want[k]).abs().max()) <= tol, (mode, rank, k)


@pytest.mark.parametrize("S,world,C", [(8, 8, 5), (8, 2, 17), (7, 3, 4), (1, 1, 5)])
def test_shard_blend_kernels_of_every_rank_combine_to_the_single_gpu_blend(S, world, C):
    """The HIP kernels of the reduce-based sharded blend (d4gs_blend_shard_*), all `world` ranks emulated on one GPU with the
    collectives done by hand (SUM of the parts, MAX of the candidates, MIN of the winners): the result must be the
    single-GPU blend (exposure.BlendFn) - max / min channels and every gradient exactly, mean channels up to fp32
    summation order (2e-6 * max|value|) - including the reference's max{raw_0..S-2, mean} quirk and exact ties."""
    import ctypes as Ct

    from deblur4dgs_amd import _lib as L
    from deblur4dgs_amd.engine import _stream
    from deblur4dgs_amd.exposure import BlendFn, reference_policy
    from deblur4dgs_amd.parallel import _shard_desc, owned_subsamples

    dev = torch.device("cuda", 0)
    H, W = 37, 21
    g = torch.Generator().manual_seed(5)
    renders = torch.rand(S, H, W, C, generator=g)
    renders[:, 0, 0, 3] = 0.0          # exact ties on the max channel
    renders[S - 1, 1, 1, 3] = 5.0      # the last sub-sample would win: the reference's max runs over {raw_0..S-2, mean}
    alphas = torch.rand(S, H, W, generator=g)
    v_out, v_acc = torch.randn(H, W, C, generator=g).to(dev), torch.randn(H, W, generator=g).to(dev)
    renders, alphas = renders.to(dev), alphas.to(dev)
    pol = reference_policy(C)
    npol = sum(1 for p in pol if p)
    r_ref = renders.clone().requires_grad_()
    a_ref = alphas.clone().requires_grad_()
    out_ref, acc_ref = BlendFn.apply(r_ref, a_ref, pol)
    torch.autograd.backward([out_ref, acc_ref], [v_out, v_acc])

    lib = L.lib()
    parts, cands, descs, locs = [], [], [], []
    for r in range(world):
        own = owned_subsamples(S, world, r)
        lr, la = renders[own].contiguous(), alphas[own].contiguous()
        desc, keep = _shard_desc(len(own), S, own, C, H * W, pol)
        part = torch.empty(H, W, C + 1, device=dev)
        cand = torch.empty(H, W, max(npol, 1), device=dev)
        L.check(lib.d4gs_blend_shard_partial_fwd(Ct.byref(desc), L.ptr(lr), L.ptr(la), L.ptr(part), L.ptr(cand), _stream()), "partial")
        parts.append(part), cands.append(cand), descs.append((desc, keep)), locs.append((own, lr, la))
    part = torch.stack(parts).sum(0)              # all-reduce SUM
    cand = torch.stack(cands).amax(0)             # all-reduce MAX
    out, acc = torch.empty(H, W, C, device=dev), torch.empty(H, W, device=dev)
    L.check(lib.d4gs_blend_shard_finish_fwd(Ct.byref(descs[0][0]), L.ptr(part), L.ptr(cand), L.ptr(out), L.ptr(acc), _stream()), "finish")
    tol = 2e-6 * float(out_ref.abs().max())
    assert float((out - out_ref).abs().max()) <= tol and float((acc - acc_ref).abs().max()) <= 2e-6
    pc = [c for c, p in enumerate(pol) if p]
    if S > 1 and pc:  # the policy channels are a max / min over exact values or the mean
        d = (out[..., pc] - out_ref[..., pc]).abs()
        assert float(d.max()) <= tol
    wins = []
    for r in range(world):
        own, lr, la = locs[r]
        win = torch.empty(H, W, max(npol, 1), dtype=torch.int32, device=dev)
        # (each rank compares with the REDUCED image, as after the forward collectives)
        L.check(lib.d4gs_blend_shard_winner(Ct.byref(descs[r][0]), L.ptr(lr), L.ptr(out), L.ptr(win), _stream()), "winner")
        wins.append(win)
    win = torch.stack(wins).amin(0) if npol and S > 1 else wins[0]  # all-reduce MIN
    for r in range(world):
        own, lr, la = locs[r]
        v_r, v_a = torch.empty_like(lr), torch.empty_like(la)
        L.check(lib.d4gs_blend_shard_bwd(Ct.byref(descs[r][0]), L.ptr(v_out), L.ptr(v_acc), L.ptr(win), L.ptr(v_r), L.ptr(v_a),
                                         _stream()), "bwd")
        torch.cuda.synchronize()
        if len(own):
            # where the sharded mean differs from the stacked mean by an ulp, a max / min that equals the MEAN may pick a
            # different winner only if a raw value ties with it to the last bit - not in this data: exact equality
            assert torch.equal(v_r, r_ref.grad[own]), r
            assert torch.equal(v_a, a_ref.grad[own]), r
