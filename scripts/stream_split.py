"""Do two half-renders (sub-samples 0..3 and 4..7 of cfg2) on two HIP streams overlap their kernel tails?
Compares: one S=8 render; two S=4 renders back to back on one stream; the same two on two streams."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from deblur4dgs_amd.exposure import render_exposure

dev = torch.device("cuda:0")
name = "cfg2"
N, G, K, S, W, H = bench.CONFIGS[name]
sc, d, leaves, wimg, wacc = bench.make_inputs(name, dev)
bg = torch.ones(3, device=dev)


def render(sel):
    return render_exposure(leaves["means"], leaves["quats"], leaves["scales"], leaves["opacities"], leaves["colors"], 3,
                           leaves["motion_coefs"], leaves["rots"], leaves["transls"], leaves["times"][sel], leaves["RTs"][sel],
                           leaves["viewmat"], d["K"], W, H, background=bg, return_depth=True, blend=False)


def loss_of(r):
    return (r["renders"] * wimg).sum() + (r["alphas"][..., 0] * wacc).sum()


def full():
    for v in leaves.values():
        v.grad = None
    loss_of(render(slice(0, S))).backward()


def halves(streams):
    for v in leaves.values():
        v.grad = None
    cur = torch.cuda.current_stream()
    losses = []
    for i, st in enumerate(streams):
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            losses.append(loss_of(render(slice(i * S // 2, (i + 1) * S // 2))))
    for i, st in enumerate(streams):
        with torch.cuda.stream(st):
            losses[i].backward()
    for st in streams:
        cur.wait_stream(st)


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


s0 = torch.cuda.current_stream()
a, b = torch.cuda.Stream(), torch.cuda.Stream()
print("one S=8 render            %.3f ms" % timeit(full))
print("two S=4, one stream       %.3f ms" % timeit(lambda: halves([s0, s0])))
print("two S=4, two streams      %.3f ms" % timeit(lambda: halves([a, b])))
