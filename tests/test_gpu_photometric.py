"""SURVEY 8f-2 on the GPU: fused L1 + SSIM loss (csrc/photometric.hip through the C ABI) against the fp64 oracle.
Tolerances: value 2e-6 relative (f32 sums of ~1e5 terms, partials added in double), gradient 1e-5 of its max norm."""
import numpy as np
import pytest
import torch

from deblur4dgs_amd.losses import photometric_loss
from oracle import photometric as ph

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("B,H,W,masked", [(1, 288, 512, False), (1, 288, 512, True), (2, 45, 77, True), (3, 16, 11, False),
                                          (1, 11, 11, False), (1, 100, 33, True)])
def test_value_and_gradient_match_oracle(B, H, W, masked):
    g = torch.Generator().manual_seed(H * 1000 + W)
    gt = torch.rand(B, H, W, 3, generator=g)
    pred = (gt + 0.15 * torch.randn(B, H, W, 3, generator=g)).clamp(0, 1)
    mask = (torch.rand(B, H, W, 1, generator=g) > 0.3).float() if masked else None
    p64 = pred.double().requires_grad_()
    lo, l1o, so = ph.photometric_loss(p64, gt.double(), None if mask is None else mask.double())
    (3.0 * lo).backward()
    pg = pred.to(DEV).requires_grad_()
    loss, l1, s = photometric_loss(pg, gt.to(DEV), None if mask is None else mask.to(DEV), return_terms=True)
    (3.0 * loss).backward()
    np.testing.assert_allclose(float(loss), float(lo), rtol=2e-6)
    np.testing.assert_allclose(float(l1), float(l1o), rtol=2e-6)
    np.testing.assert_allclose(float(s), float(so), rtol=2e-6, atol=2e-7)
    ref = p64.grad.numpy()
    np.testing.assert_allclose(pg.grad.cpu().numpy(), ref, rtol=0, atol=1e-5 * np.abs(ref).max())


def test_deterministic_and_rejects_bad_shapes():
    g = torch.Generator().manual_seed(0)
    a, b = torch.rand(1, 64, 64, 3, generator=g).to(DEV), torch.rand(1, 64, 64, 3, generator=g).to(DEV)
    x = a.clone().requires_grad_()
    l0 = photometric_loss(x, b)
    l0.backward()
    y = a.clone().requires_grad_()
    l1 = photometric_loss(y, b)
    l1.backward()
    assert torch.equal(l0, l1) and torch.equal(x.grad, y.grad)
    with pytest.raises(RuntimeError):
        photometric_loss(torch.rand(1, 8, 64, 3, device=DEV), torch.rand(1, 8, 64, 3, device=DEV))  # H < 11
    with pytest.raises(RuntimeError):
        photometric_loss(torch.rand(1, 32, 32, 4, device=DEV), torch.rand(1, 32, 32, 4, device=DEV))  # C != 3
    with pytest.raises(RuntimeError):
        photometric_loss(torch.rand(1, 32, 32, 3), torch.rand(1, 32, 32, 3))  # CPU tensors: no fallback
