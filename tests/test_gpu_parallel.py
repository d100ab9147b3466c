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


@pytest.mark.parametrize("mode", ["views", "exposure"])
def test_sharded_step_equals_plain_step(group, mode):
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
    sh = ShardedExposure(1, 0, mode=mode)
    for _ in range(2):  # second step: the previous gradients alias the flat buffer and must not be accumulated into
        sh.step(got, K, W, H, bg, wimg, wacc)
    torch.cuda.synchronize()
    for k in NAMES:
        a, b = got[k].grad, ref[k].grad
        assert a is not None and a.data_ptr() == sh.reducer.views[k].data_ptr(), k
        tol = 1e-5 * max(1.0, float(b.abs().max()))
        assert float((a - b).abs().max()) <= tol, (k, float((a - b).abs().max()))
