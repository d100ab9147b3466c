"""The C-ABI library loads and exports every symbol include/d4gs.h declares; argument validation happens before
any HIP call, so the error paths can be exercised without a GPU."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from deblur4dgs_amd import build

    return C.CDLL(build.build())


def _declared():
    src = open(os.path.join(ROOT, "include", "d4gs.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(d4gs_[a-z_0-9]+)\s*\(", src)))


def test_exports_match_header(lib):
    names = _declared()
    assert len(names) >= 13
    for n in names:
        assert hasattr(lib, n), f"{n} declared in d4gs.h but not exported"


def test_version_and_error_paths(lib):
    from deblur4dgs_amd import _lib as L

    assert lib.d4gs_version() == 100
    lib.d4gs_last_error.restype = C.c_char_p
    assert lib.d4gs_project_fwd(None, None, None, None) == -1  # D4GS_EINVAL, no HIP call made
    assert b"NULL" in lib.d4gs_last_error()
    d = L.Dims(N=10, G=20, K=1, T=1, S=1, D=3, width=16, height=16)  # G > N
    assert lib.d4gs_project_fwd(C.byref(d), None, None, None) == -1
    assert b"motion dims" in lib.d4gs_last_error()
    d = L.Dims(N=10, G=0, K=0, T=0, S=1, D=3, width=16, height=16)
    assert lib.d4gs_project_fwd(C.byref(d), None, None, None) == -1  # NULL required inputs


def test_struct_layouts_match_header():
    """ctypes mirrors must have exactly the fields of the C structs, in order."""
    from deblur4dgs_amd import _lib as L

    src = open(os.path.join(ROOT, "include", "d4gs.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)

    def fields(name):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), src, re.S).group(1)
        out = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            decl = re.sub(r"^(const\s+)?(float|int32_t|int64_t|uint64_t|void)\s*", "", decl)
            out += [x.strip().lstrip("*").strip() for x in decl.split(",")]
        return out

    for cname, cls in (("D4gsDims", L.Dims), ("D4gsProjIn", L.ProjIn), ("D4gsProjOut", L.ProjOut),
                       ("D4gsIsect", L.Isect), ("D4gsRaster", L.Raster), ("D4gsRasterGrads", L.RasterGrads),
                       ("D4gsLeafGrads", L.LeafGrads)):
        assert fields(cname) == [f[0] for f in cls._fields_], cname


def test_product_never_imports_oracle():
    """The shipped package must not route through the oracle (or any CPU fallback)."""
    pkg = os.path.join(ROOT, "deblur4dgs_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            txt = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, re.M), fn
