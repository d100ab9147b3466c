"""SURVEY 8c "K1": closed-form known answers asserted DIRECTLY on the HIP rasterizer (seam S1), without the oracle.

The rasterizer oracle is a restatement of gsplat 1.1.1 from its published algorithm (parity unpinned), so agreement with it
proves agreement with the builder's reading twice over.  These cases do not route through it: every expected value below
is a formula of SURVEY Appendix A.4 evaluated in numpy fp64 inside this file (isotropic Gaussians, identity view matrix -
the 3-D covariance is s^2 I whatever the quaternion, so the projection has a two-line closed form), and the kernels are
compared with it at fp32 rounding (1e-5 relative).  Nothing under `oracle/` is imported here.
Reference call site: flow3d/scene_model.py:360-373 (gsplat.rendering.rasterization, packed=False, C=1).
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

EPS2D = 0.3
ALPHA_MIN = 1.0 / 255.0


def _dev():
    assert torch.cuda.is_available(), "the known-answer tests need an MI355X"
    return torch.device("cuda:0")


def _K(f, W, H):
    return np.array([[f, 0, W / 2], [0, f, H / 2], [0, 0, 1]], np.float64)


def closed_form(mean, s, f, W, H):
    """SURVEY A.4 steps 2-5 for an isotropic Gaussian (Sigma = s^2 I) seen through the identity view matrix.
    -> dict(mu2 [2], cov2 [2,2] blurred, conic (a, b, c), radius, visible)"""
    x, y, z = (float(v) for v in mean)
    limx, limy = 1.3 * (0.5 * W / f), 1.3 * (0.5 * H / f)
    tx, ty = z * min(limx, max(-limx, x / z)), z * min(limy, max(-limy, y / z))
    J = np.array([[f / z, 0, -f * tx / z**2], [0, f / z, -f * ty / z**2]])
    cov2 = J @ (s * s * np.eye(3)) @ J.T + EPS2D * np.eye(2)
    det = np.linalg.det(cov2)
    inv = np.linalg.inv(cov2)
    mid = 0.5 * (cov2[0, 0] + cov2[1, 1])
    lam = mid + math.sqrt(max(0.01, mid * mid - det))
    radius = math.ceil(3.0 * math.sqrt(lam))
    mu2 = np.array([f * x / z + W / 2, f * y / z + H / 2])
    return dict(mu2=mu2, cov2=cov2, conic=(inv[0, 0], inv[0, 1], inv[1, 1]), radius=radius)


def alpha_image(cf, o, W, H):
    """alpha of ONE splat at every pixel centre (A.4 step 8): min(0.999, o exp(-sigma)), 0 below 1/255."""
    a, b, c = cf["conic"]
    px, py = np.meshgrid(np.arange(W) + 0.5, np.arange(H) + 0.5)
    dx, dy = cf["mu2"][0] - px, cf["mu2"][1] - py
    sig = 0.5 * (a * dx * dx + c * dy * dy) + b * dx * dy
    al = np.minimum(0.999, o * np.exp(-sig))
    al[(sig < 0) | (al < ALPHA_MIN)] = 0.0
    return al, sig, dx, dy


def render(means, scales, opac, colors, f, W, H, bg=None, mode="RGB+ED", need_grad=False, **kw):
    from deblur4dgs_amd.rasterization import rasterization

    dev = _dev()
    n = len(means)
    t = dict(means=torch.tensor(np.ascontiguousarray(means), dtype=torch.float32, device=dev),
             quats=torch.tensor([[0.6, 0.0, 0.8, 0.0]] * n, dtype=torch.float32, device=dev),  # any rotation: Sigma is s^2 I
             scales=torch.tensor(np.asarray(scales, np.float64).reshape(n, 1).repeat(3, 1), dtype=torch.float32, device=dev),
             opac=torch.tensor(np.ascontiguousarray(opac), dtype=torch.float32, device=dev),
             colors=torch.tensor(np.ascontiguousarray(colors), dtype=torch.float32, device=dev))
    if need_grad:
        for v in t.values():
            v.requires_grad_()
    V = torch.eye(4, device=dev)
    Km = torch.tensor(_K(f, W, H), dtype=torch.float32, device=dev)
    bgs = None if bg is None else torch.tensor(np.asarray(bg), dtype=torch.float32, device=dev)[None]
    rc, ra, info = rasterization(t["means"], t["quats"], t["scales"], t["opac"], t["colors"], V[None], Km[None], W, H,
                                 backgrounds=bgs, render_mode=mode, **kw)
    return rc[0], ra[0, ..., 0], info, t


def close(got, want, rtol=2e-5, atol=0.0, what=""):
    got = got.detach().double().cpu().numpy() if torch.is_tensor(got) else np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    tol = rtol * max(np.abs(want).max(), 1e-30) + atol
    err = np.abs(got - want).max()
    assert err <= tol, f"{what}: max |got - closed form| = {err:.3e} > {tol:.3e}"


# ---------------------------------------------------------------------------------------------------------------------
def test_single_isotropic_gaussian_image_and_analytic_gradients():
    """One isotropic Gaussian projected exactly onto the centre of pixel (10, 20): alpha = min(0.999, o e^{-d^T Q d / 2}),
    Q^-1 = s^2 f^2 / z^2 (+ the off-axis Jacobian terms) + 0.3 I; image, expected depth, radius, conic - and the gradients of
    a linear loss with respect to colour, opacity, the 2-D mean (closed form) and the 3-D mean / scale (central differences of
    the closed form in fp64)."""
    W, H, f, z, s, o = 48, 32, 40.0, 4.0, 0.3, 0.7
    col = np.array([0.2, 0.5, 0.9])
    mean = np.array([(10.5 - W / 2) * z / f, (20.5 - H / 2) * z / f, z])
    cf = closed_form(mean, s, f, W, H)
    al, sig, dx, dy = alpha_image(cf, o, W, H)
    rc, ra, info, t = render([mean], [s], [o], [col], f, W, H, bg=[0, 0, 0], need_grad=True)
    info["means2d"].retain_grad()
    assert abs(cf["mu2"][0] - 10.5) < 1e-12 and abs(cf["mu2"][1] - 20.5) < 1e-12
    close(info["means2d"][0, 0], cf["mu2"], what="means2d")
    close(info["conics"].view(-1, 3)[0], cf["conic"], what="conic")
    assert int(info["radii"][0, 0]) == cf["radius"]
    assert abs(ra[20, 10].item() - o) < 1e-6  # d = 0 at the pixel centre: alpha = opacity
    close(ra, al, what="alpha image")
    close(rc[..., :3], al[..., None] * col, what="colour image")
    close(rc[..., 3], np.where(al > 0, z, 0.0), what="expected depth (sum w z / alpha)")
    assert (al > 0).sum() > 20 and (al == 0).sum() > 100  # the footprint is a proper subset of the image

    # loss = <w_c, colours> + <w_a, alpha>; one splat, zero background: colour_p = alpha_p c, alpha unclamped (o < 0.999)
    g = np.random.default_rng(3)
    wc, wa = g.standard_normal((H, W, 3)), g.standard_normal((H, W))
    loss = (rc[..., :3] * torch.tensor(wc, dtype=torch.float32, device=rc.device)).sum() \
        + (ra * torch.tensor(wa, dtype=torch.float32, device=rc.device)).sum()
    loss.backward()
    torch.cuda.synchronize()
    v_al = (wc * col).sum(-1) + wa                    # dL / d alpha_p
    close(t["colors"].grad[0], (wc * al[..., None]).sum((0, 1)), what="dL/dcolour = sum_p w_p alpha_p")
    close(t["opac"].grad[0], (v_al * al).sum() / o, what="dL/dopacity = sum_p v_p alpha_p / o")
    a, b, c = cf["conic"]
    gx = (v_al * (-al) * (a * dx + b * dy)).sum()     # d alpha / d mu = -alpha Q d,  d = mu - p
    gy = (v_al * (-al) * (b * dx + c * dy)).sum()
    close(info["means2d"].grad[0, 0], [gx, gy], what="dL/dmeans2d = -sum_p v_p alpha_p Q d_p")

    def L(mean_, s_):  # the same loss through the closed form, fp64; the support (alpha >= 1/255) is part of it
        cf_ = closed_form(mean_, s_, f, W, H)
        al_ = alpha_image(cf_, o, W, H)[0]
        return (v_al * al_).sum()

    h = 1e-6
    gm = [(L(mean + h * e, s) - L(mean - h * e, s)) / (2 * h) for e in np.eye(3)]
    gs = (L(mean, s + h) - L(mean, s - h)) / (2 * h)
    # (pixels crossing the 1/255 support edge inside +-h would make the difference quotient jump: none do at h = 1e-6 -
    # the loss is smooth there to 1e-7; fp32 kernels against it: 1e-4 of the largest component)
    close(t["means"].grad[0], gm, rtol=1e-4, what="dL/dmean (3-D) vs central differences of the closed form")
    close(t["scales"].grad[0].sum(), gs, rtol=1e-4, what="dL/dscale (sum of the three axes) vs central differences")


def test_two_overlapping_splats_order_and_transmittance():
    """Nearer splat first whatever the input order; C = a_f c_f + (1 - a_f) a_b c_b, alpha = 1 - (1 - a_f)(1 - a_b)."""
    W = H = 16
    f = 16.0
    means = [[0.0, 0.0, 5.0], [0.0, 0.0, 3.0]]  # the second one is in front
    s, o = 0.5, [0.6, 0.5]
    col = np.array([[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]])
    rc, ra, info, _ = render(means, [s, s], o, col, f, W, H, bg=[0, 0, 0], mode="RGB")
    assert info["flatten_ids"][:2].tolist() == [1, 0]
    al_b = alpha_image(closed_form(means[0], s, f, W, H), o[0], W, H)[0]
    al_f = alpha_image(closed_form(means[1], s, f, W, H), o[1], W, H)[0]
    want = al_f[..., None] * col[1] + ((1 - al_f) * al_b)[..., None] * col[0]
    close(rc, want, what="two-splat colour")
    close(ra, 1 - (1 - al_f) * (1 - al_b), what="two-splat alpha")
    assert (al_f[8, 8] > 0.3) and (al_b[8, 8] > 0.3)  # both really overlap at the centre
    # swapping the input order changes the ids, not the picture
    rc2, ra2, info2, _ = render(means[::-1], [s, s], o[::-1], col[::-1], f, W, H, bg=[0, 0, 0], mode="RGB")
    assert info2["flatten_ids"][:2].tolist() == [0, 1]
    assert torch.equal(rc, rc2) and torch.equal(ra, ra2)


@pytest.mark.parametrize("above", [True, False])
def test_alpha_threshold_at_one_over_255(above):
    """A splat whose alpha peaks just above 1/255 paints exactly its centre pixel; just below it paints nothing."""
    W = H = 16
    f, z, s = 16.0, 4.0, 0.05  # sigma_px^2 = (s f / z)^2 + 0.3 = 0.34: the neighbours' alpha is e^{-1.47} of the peak
    o = ALPHA_MIN * (1.0 + (1e-3 if above else -1e-3))
    mean = [(5.5 - W / 2) * z / f, (9.5 - H / 2) * z / f, z]
    rc, ra, info, _ = render([mean], [s], [o], [[1.0, 1.0, 1.0]], f, W, H, bg=[0, 0, 0], mode="RGB")
    want = np.zeros((H, W))
    if above:
        want[9, 5] = o
    close(ra, want, rtol=1e-5, what="alpha at the 1/255 cut")
    close(rc[..., 0], want, rtol=1e-5, what="colour at the 1/255 cut")
    assert int(info["radii"][0, 0]) > 0  # projected and binned by radius either way; the cut is per pixel


def test_transmittance_stops_at_1e_minus_4():
    """40 identical splats stacked along the optical axis: a pixel stops at the first k with T_k (1 - alpha_k) <= 1e-4 and
    that splat is NOT composited (T is not advanced): last contributor, final alpha and colour by the rule replayed in fp64."""
    W = H = 16
    f, s, o, n = 16.0, 1.0, 0.5, 40
    zs = np.linspace(2.0, 6.0, n)
    means = np.stack([np.zeros(n), np.zeros(n), zs], 1)
    col = np.linspace(0.1, 1.0, n)[:, None] * np.ones((1, 3))
    rc, ra, info, _ = render(means, [s] * n, [o] * n, col, f, W, H, bg=[0, 0, 0], mode="RGB")
    assert info["flatten_ids"][:n].tolist() == list(range(n))  # one tile, already in depth order
    T, acc, last, margin = 1.0, 0.0, -1, np.inf
    for k in range(n):
        al = alpha_image(closed_form(means[k], s, f, W, H), o, W, H)[0][8, 8]
        assert al >= ALPHA_MIN
        nT = T * (1 - al)
        margin = min(margin, abs(nT - 1e-4) / 1e-4)
        if nT <= 1e-4:
            break
        acc += al * T * col[k, 0]
        T, last = nT, k
    assert margin > 1e-2, "the case must not sit on the threshold"
    assert 5 < last < n - 1  # terminated early
    assert int(info["last_ids"][0, 8, 8]) == last
    assert abs(ra[8, 8].item() - (1 - T)) < 1e-6 and T > 1e-4
    assert abs(rc[8, 8, 0].item() - acc) < 2e-6


def test_near_plane_and_tile_rectangles():
    """z just inside / outside near_plane = 0.01; a radius that reaches exactly to / just across a tile edge."""
    W, H, f = 64, 48, 64.0
    means = [[0.0, 0.0, 0.0099], [0.0, 0.0, 0.0101], [0.0, 0.0, -1.0], [100.0, 0.0, 1.0], [0.0, 0.0, 2.0]]
    rc, ra, info, _ = render(means, [1e-5] * 5, [0.5] * 5, [[1, 1, 1]] * 5, f, W, H, mode="RGB")
    radii = info["radii"][0].tolist()
    assert radii[0] == 0 and radii[2] == 0 and radii[3] == 0  # nearer than the plane, behind the camera, off-screen
    assert radii[1] > 0 and radii[4] > 0
    # gsplat's rectangle: tiles [floor((mu - r) / 16), ceil((mu + r) / 16)) - counted with exact culling off
    z, s = 4.0, 0.01  # sigma_px^2 = 0.3256 -> lambda = 0.3256 + 0.1 (the sqrt floor max(0.01, .)) -> radius = ceil(3 * 0.652) = 2
    cases = [((13.5, 8.5), 1), ((14.0, 8.5), 1),   # 14 + 2 = 16.0 exactly: ceil(1.0) = 1 -> still one tile
             ((14.5, 8.5), 2), ((14.5, 14.5), 4), ((8.5, 8.5), 1), ((15.5, 24.5), 2)]
    ms = [[(px - W / 2) * z / f, (py - H / 2) * z / f, z] for (px, py), _ in cases]
    rc, ra, info, _ = render(ms, [s] * len(ms), [0.9] * len(ms), [[1, 1, 1]] * len(ms), f, W, H, mode="RGB", exact_cull=False)
    assert closed_form(ms[0], s, f, W, H)["radius"] == 2
    assert info["tiles_per_gauss"][0].tolist() == [c for _, c in cases]
    # and the picture across the tile edge is the closed form (the splat at (15.5, 24.5) paints pixels of tiles (0, 1) and (1, 1))
    T = np.ones((H, W))
    for m in ms:  # same depth: the order is irrelevant for alpha
        T *= 1 - alpha_image(closed_form(m, s, f, W, H), 0.9, W, H)[0]
    close(ra, 1 - T, what="alpha across a tile edge")
    assert ra[8, 15].item() > 0.01 and ra[8, 16].item() == 0.0  # 1 px / 2 px from (14.5, 8.5), sigma_px = 0.57: in / below the cut
    assert ra[24, 14].item() > 0.01 and ra[24, 16].item() > 0.01  # one pixel either side of (15.5, 24.5): both tiles painted


def test_fov_clamp_of_the_projection_jacobian():
    """x / z beyond 1.3 tan(fov / 2): the Jacobian uses the CLAMPED x (A.4 step 3), the 2-D mean does not.  The conic, the
    radius and the pixels the splat reaches at the image border must follow the clamped closed form - and differ
    measurably from the unclamped one."""
    W, H, f, z, s, o = 64, 48, 64.0, 2.0, 0.5, 0.9
    lim = 1.3 * 0.5 * W / f
    mean = np.array([1.5 * (0.5 * W / f) * z, 0.0, z])  # x / z = 0.75 > lim = 0.65; projects to x = 80 (16 px off-screen)
    assert mean[0] / z > lim
    cf = closed_form(mean, s, f, W, H)
    rc, ra, info, _ = render([mean], [s], [o], [[1.0, 1.0, 1.0]], f, W, H, bg=[0, 0, 0], mode="RGB")
    close(info["means2d"][0, 0], [80.0, 24.0], what="unclamped 2-D mean")
    close(info["conics"].view(-1, 3)[0], cf["conic"], what="conic with the clamped Jacobian")
    assert int(info["radii"][0, 0]) == cf["radius"]
    al = alpha_image(cf, o, W, H)[0]
    assert al[24, 63] > 0.05  # it does reach the image
    close(ra, al, what="alpha of a clamped splat")
    J_un = np.array([[f / z, 0, -f * mean[0] / z**2], [0, f / z, 0.0]])
    inv_un = np.linalg.inv(J_un @ (s * s * np.eye(3)) @ J_un.T + EPS2D * np.eye(2))
    assert abs(inv_un[0, 0] - cf["conic"][0]) > 1e-2 * cf["conic"][0]  # the clamp matters in this case
