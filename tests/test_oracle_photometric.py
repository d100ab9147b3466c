"""Oracle self-checks for the photometric term.  pytorch-msssim is absent (parity unpinned, see
oracle/photometric.py); what can be checked here: closed forms, and an independent numpy/scipy evaluation of the same
published definition."""
import numpy as np
import torch
from scipy.ndimage import correlate1d

from oracle import photometric as ph


def _ssim_numpy(X, Y):
    w = ph.gaussian_window().numpy()

    def filt(a):  # [B,C,H,W] valid separable correlation
        a = correlate1d(a, w, axis=2, mode="constant")[:, :, 5:-5]
        return correlate1d(a, w, axis=3, mode="constant")[:, :, :, 5:-5]

    mu1, mu2 = filt(X), filt(Y)
    s1, s2, s12 = filt(X * X) - mu1 ** 2, filt(Y * Y) - mu2 ** 2, filt(X * Y) - mu1 * mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    m = (2 * mu1 * mu2 + C1) / (mu1 ** 2 + mu2 ** 2 + C1) * (2 * s12 + C2) / (s1 + s2 + C2)
    return m.reshape(*m.shape[:2], -1).mean(-1).mean()


def test_ssim_closed_forms_and_independent_numpy():
    g = torch.Generator().manual_seed(0)
    x = torch.rand(2, 3, 37, 52, generator=g, dtype=torch.float64)
    y = (x + 0.2 * torch.randn(2, 3, 37, 52, generator=g, dtype=torch.float64)).clamp(0, 1)
    assert abs(float(ph.ssim(x, x)) - 1.0) < 1e-12
    np.testing.assert_allclose(float(ph.ssim(x, y)), _ssim_numpy(x.numpy(), y.numpy()), rtol=1e-10)
    assert abs(float(ph.ssim(x, y)) - float(ph.ssim(y, x))) < 1e-14
    c1, c2 = 0.3, 0.7  # constant images: variances vanish, only the luminance term is left
    want = (2 * c1 * c2 + 1e-4) / (c1 * c1 + c2 * c2 + 1e-4)
    got = float(ph.ssim(torch.full((1, 3, 20, 20), c1, dtype=torch.float64), torch.full((1, 3, 20, 20), c2, dtype=torch.float64)))
    assert abs(got - want) < 1e-12
    assert abs(float(ph.gaussian_window().sum()) - 1.0) < 1e-15


def test_photometric_loss_composition():
    g = torch.Generator().manual_seed(1)
    p, q = torch.rand(1, 30, 33, 3, generator=g, dtype=torch.float64), torch.rand(1, 30, 33, 3, generator=g, dtype=torch.float64)
    m = (torch.rand(1, 30, 33, 1, generator=g) > 0.4).double()
    loss, l1, s = ph.photometric_loss(p, q, m)
    np.testing.assert_allclose(float(l1), float(((p - q) * m).abs().mean()), rtol=1e-14)
    np.testing.assert_allclose(float(loss), 0.8 * float(l1) + 0.2 * (1 - float(s)), rtol=1e-14)
