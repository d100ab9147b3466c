"""examples/c_abi_demo.cpp: the drop-in boundary used WITHOUT PyTorch - a host program that links libd4gs.so + the HIP runtime,
renders a frame forward + backward through d4gs_forward / d4gs_backward (hipMalloc'ed buffers, explicit stream, one workspace)
and through the CPU twins, and compares the two.  Without a GPU: it must compile and link against the header and the library
(every symbol it uses resolves); on the device: it must run and agree."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = "/tmp/d4gs_c_abi_demo"


def _build():
    from deblur4dgs_amd import _lib as L

    L.lib()  # (raises if libd4gs.so is missing)
    libdir = os.path.dirname(L.LIB_PATH)
    cmd = ["hipcc", "--offload-arch=gfx950", "-O2", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "c_abi_demo.cpp"),
           "-L" + libdir, "-ld4gs", "-Wl,-rpath," + libdir, "-o", EXE]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return EXE


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_c_abi_demo_compiles_and_links():
    exe = _build()
    assert os.path.getsize(exe) > 0


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_c_abi_demo_device_and_cpu_twin_agree_without_pytorch():
    exe = _build()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr
