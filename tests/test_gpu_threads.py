"""SURVEY 8b's threading contract on the device (missing #5): "re-entrant, thread-local error string; viewer thread +
trainer thread".  The reference renders from its trainer thread (flow3d/trainer.py:204-207 hands `Renderer.render_fn` to the viewer
server) and from the viewer's thread (flow3d/renderer.py:57-89) on ONE model.  Here: a trainer thread runs training renders
(SceneModel.render forward + backward, its own stream) while a viewer thread runs `render_view` (inference mode, its own stream, its
own resolutions - and the trainer's, so that both threads also meet on one list-size key) on the same SceneModel; every image and
every gradient must be BITWISE what the same calls give one after the other.  (The PyTorch-free version of the same check, raw C ABI
from two std::threads: examples/c_abi_demo.cpp / tests/test_c_abi_demo.py; the error string: tests/test_c_abi.py.)"""
import math
import threading

import pytest
import torch

from deblur4dgs_amd.synth import make_scene

pytestmark = pytest.mark.gpu


def _model(dev):
    from deblur4dgs_amd.scene_model import GaussianParams, MotionBases, SceneModel

    N, G, K, W, H = 30_000, 12_000, 6, 256, 144
    sc = make_scene(N, G, K, 1, W, H, seed=77, T=12)
    sc["scales"] = sc["scales"] + 0.7
    keys = ("means", "quats", "scales", "colors", "opacities")
    fg = GaussianParams(*[sc[k][:G].clone() for k in keys], motion_coefs=sc["motion_coefs"].clone())
    bg = GaussianParams(*[sc[k][G:].clone() for k in keys])
    model = SceneModel(sc["K"][None].clone(), sc["viewmat"][None].clone(), fg, MotionBases(sc["rots"].clone(), sc["transls"].clone()), bg).to(dev)
    torch.manual_seed(5)
    with torch.no_grad():
        for head in (model.move_model.RT_head0, model.move_model.RT_head1):
            head[-1].bias.copy_(0.003 * torch.randn(6))
    return model, sc, (W, H)


def test_viewer_thread_and_trainer_thread_render_the_same_bits_as_one_after_the_other():
    from deblur4dgs_amd.scene_model import render_view

    dev = torch.device("cuda:0")
    model, sc, (W, H) = _model(dev)
    w2c, Kmat = sc["viewmat"][None].to(dev), sc["K"][None].to(dev)
    g = torch.Generator().manual_seed(1)
    wimg, wdep = torch.randn(1, H, W, 3, generator=g).to(dev), torch.randn(1, H, W, 1, generator=g).to(dev)
    ITERS = 12
    watched = [model.fg.params["means"], model.fg.params["motion_coefs"], model.bg.params["scales"], model.motion_bases.params["rots"],
               model.move_model.RT_head0[-1].bias]

    def train_step(i):
        for p in model.parameters():
            p.grad = None
        out = model.render(1.0 + 0.5 * i, w2c, Kmat, (W, H), return_depth=True, return_mask=True, mode="blury", stage="second")
        ((out["img"] * wimg).sum() + (out["depth"] * wdep).sum()).backward()
        xys = torch.cat([x.grad for x in model._current_xys], 0)
        return [out["img"].detach().clone(), xys.clone()] + [p.grad.detach().clone() for p in watched]

    def view(i):
        c2w = torch.eye(4, device=dev)
        c2w[0, 3], c2w[2, 3] = 0.02 * i, -0.05 * (i % 3)
        wh = (W, H) if i % 2 == 0 else (320, 192)  # every other frame at the trainer's resolution: one shared list-size key
        return render_view(model, None if i % 5 == 4 else 0.5 + 0.7 * i, c2w, math.radians(50.0), wh).clone()

    def loop(fn, stream, sink, errs, bar=None):
        try:
            with torch.cuda.stream(stream):
                if bar is not None:
                    bar.wait()
                for i in range(ITERS):
                    sink.append(fn(i))
                stream.synchronize()
        except BaseException as e:  # noqa: BLE001 - reported by the main thread
            errs.append(e)

    # one after the other (each on its own stream, as below)
    sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    want_t, want_v, errs = [], [], []
    loop(train_step, sA, want_t, errs)
    loop(view, sB, want_v, errs)
    assert not errs, errs
    # ... and concurrently: two host threads, two streams, one model, one library
    for rep in range(2):
        got_t, got_v, errs = [], [], []
        bar = threading.Barrier(2)
        th = [threading.Thread(target=loop, args=(train_step, sA, got_t, errs, bar)),
              threading.Thread(target=loop, args=(view, sB, got_v, errs, bar))]
        for t in th:
            t.start()
        for t in th:
            t.join(timeout=600)
        assert not errs, errs
        assert not any(t.is_alive() for t in th)
        torch.cuda.synchronize()
        assert len(got_t) == ITERS and len(got_v) == ITERS
        for i in range(ITERS):
            assert torch.equal(got_v[i], want_v[i]), ("viewer frame", rep, i)
            for j, (a, b) in enumerate(zip(got_t[i], want_t[i])):
                assert torch.equal(a, b), ("trainer step", rep, i, j)
    assert float(want_t[0][2].abs().max()) > 0 and int(want_v[0].max()) > 0
