"""The wave reduction every gradient of the composite backward goes through (csrc/common.h: wave_sum_store - permlane swaps, then
one DPP ladder shared by the folded registers), checked directly: R per-lane values summed over the 64 lanes of one wave, for every
folding pattern (R = 4 a + 2 b + c) including the ones no kernel instantiates today.  The hook lives in the A/B build only
(tests/libd4gs_variants.so)."""
import ctypes as C
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("R", [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 21, 22, 23])
def test_wave_sum_store_sums_every_value_into_its_own_slot(R):
    from deblur4dgs_amd import build

    assert os.path.exists(build.VARIANTS_LIB), "run __graft_entry__.build() (builds tests/libd4gs_variants.so)"
    lib = C.CDLL(build.VARIANTS_LIB)
    lib.d4gs_test_wave_sum.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.d4gs_test_wave_sum.restype = C.c_int
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(R)
    for trial in range(3):
        x = torch.randn(64, R, generator=g)
        if trial == 1:  # one lane, one value: lands in exactly one slot
            x = torch.zeros(64, R)
            x[(7 * R + 3) % 64, R // 2] = 1.0
        if trial == 2:  # distinct weights per (lane, value): a wrong slot or a lane counted twice cannot cancel
            x = (torch.arange(64)[:, None] + 1.0) * (10.0 ** (torch.arange(R)[None] % 4)) / 64.0
        xd = x.to(dev).contiguous()
        out = torch.full((R,), float("nan"), device=dev)
        rc = lib.d4gs_test_wave_sum(R, xd.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        torch.cuda.synchronize()
        ref = x.double().sum(0)
        tol = 1e-5 * x.double().abs().sum(0) + 1e-12
        assert ((out.cpu().double() - ref).abs() <= tol).all(), (R, trial, out.cpu(), ref)
