"""Oracle: the photometric term of the reference's training loss.

Test infrastructure (see oracle/__init__.py).  The reference computes, three to four times per step,
    0.8 * F.l1_loss(pred * m, gt * m) + 0.2 * (1 - SSIM(pred * m, gt * m))
(flow3d/trainer.py:388-392,575-586) with `SSIM = pytorch_msssim.SSIM(data_range=1.0, size_average=True, channel=3)`
(trainer.py:93).  pytorch-msssim==1.0.0 (requirements.txt:366) is a third-party wheel absent from /root/reference and
from this image: its published algorithm is restated below [RECALLED] - **parity unpinned**:
  * 11-tap Gaussian window, sigma 1.5, normalised to sum 1, applied separably per channel with NO padding
    ("valid": the maps are (H-10) x (W-10));
  * mu = filt(X), sigma^2 = filt(X*X) - mu^2, sigma12 = filt(X*Y) - mu1*mu2 (compensation 1.0);
  * C1 = (0.01 * data_range)^2, C2 = (0.03 * data_range)^2;
  * ssim_map = (2 mu1 mu2 + C1) / (mu1^2 + mu2^2 + C1) * (2 sigma12 + C2) / (sigma1^2 + sigma2^2 + C2);
  * mean over the valid pixels per (image, channel), then over images and channels (nonnegative_ssim=False).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

WIN, SIGMA, K1, K2 = 11, 1.5, 0.01, 0.03


def gaussian_window(dtype=torch.float64):
    c = torch.arange(WIN, dtype=dtype) - WIN // 2
    g = torch.exp(-(c ** 2) / (2 * SIGMA ** 2))
    return g / g.sum()


def _filt(x, win):  # x [B,C,H,W], separable valid correlation per channel
    C = x.shape[1]
    w = win.to(device=x.device, dtype=x.dtype)
    x = F.conv2d(x, w.view(1, 1, -1, 1).expand(C, 1, -1, 1), groups=C)
    return F.conv2d(x, w.view(1, 1, 1, -1).expand(C, 1, 1, -1), groups=C)


def ssim(X, Y, data_range=1.0):
    """X, Y [B,C,H,W] -> scalar (size_average=True)."""
    win = gaussian_window(X.dtype)
    C1, C2 = (K1 * data_range) ** 2, (K2 * data_range) ** 2
    mu1, mu2 = _filt(X, win), _filt(Y, win)
    s1 = _filt(X * X, win) - mu1 * mu1
    s2 = _filt(Y * Y, win) - mu2 * mu2
    s12 = _filt(X * Y, win) - mu1 * mu2
    cs = (2 * s12 + C2) / (s1 + s2 + C2)
    m = (2 * mu1 * mu2 + C1) / (mu1 * mu1 + mu2 * mu2 + C1) * cs
    return m.flatten(2).mean(-1).mean()


def photometric_loss(pred, gt, mask=None, w_l1=0.8, w_ssim=0.2):
    """pred, gt [B,H,W,C] (channel-last, as rendered), mask [B,H,W,1] or None -> (loss, l1, ssim)."""
    if mask is not None:
        pred, gt = pred * mask, gt * mask
    l1 = (pred - gt).abs().mean()
    s = ssim(pred.permute(0, 3, 1, 2), gt.permute(0, 3, 1, 2))
    return w_l1 * l1 + w_ssim * (1 - s), l1, s
