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


def test_dynamic_symbol_table_is_exactly_the_header(lib):
    """`nm -D`: the library exports the C entry points of include/d4gs.h and nothing else - no mangled C++ internals, no libstdc++
    template instantiations (-fvisibility=hidden + the linker version script csrc/libd4gs.map).  Both the product library and the
    tests' A/B build."""
    import subprocess

    from deblur4dgs_amd import build

    for path in (build.LIB, build.build_variants()):
        out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
        syms = sorted(line.split()[-1] for line in out.splitlines() if line.strip())
        if path != build.LIB:  # the A/B build may carry test hooks (d4gs_test_*: tests/test_gpu_wave_sum.py); the product library may not
            syms = [x for x in syms if not x.startswith("d4gs_test_")]
        assert syms == _declared(), (path, sorted(set(syms) ^ set(_declared())))


def test_version_and_error_paths(lib):
    from deblur4dgs_amd import _lib as L

    assert lib.d4gs_version() == 305
    lib.d4gs_last_error.restype = C.c_char_p
    assert lib.d4gs_project_fwd(None, None, None, None) == -1  # D4GS_EINVAL, no HIP call made
    assert b"NULL" in lib.d4gs_last_error()
    d = L.Dims(N=10, G=20, K=1, T=1, S=1, D=3, width=16, height=16)  # G > N
    assert lib.d4gs_project_fwd(C.byref(d), None, None, None) == -1
    assert b"motion dims" in lib.d4gs_last_error()
    d = L.Dims(N=10, G=0, K=0, T=0, S=1, D=3, width=16, height=16)
    assert lib.d4gs_project_fwd(C.byref(d), None, None, None) == -1  # NULL required inputs


def test_raster_bwd_rejects_misaligned_scratch_before_any_hip_call(lib):
    """k_gather streams isect_grad as 16-byte words and isect_live as 4-byte words: the C ABI says so and checks it on
    the host (fake addresses - nothing is dereferenced before the validation)."""
    from deblur4dgs_amd import _lib as L

    lib.d4gs_last_error.restype = C.c_char_p
    d = L.Dims(N=10, G=0, K=0, T=0, S=1, D=3, width=16, height=16)
    fake = 0x10000
    pout = L.ProjOut(**{n: fake for n, _ in L.ProjOut._fields_})
    isect = L.Isect(n_isect=4, max_tile_count=0, keys=fake, gid_of_emit=fake, sorted_gid=fake, sorted_emit=fake)
    ras = L.Raster(**{n: fake for n, _ in L.Raster._fields_})
    ok = dict(v_render_colors=fake, isect_grad=fake, isect_live=fake, v_means2d=fake, v_conics=fake, v_depths=fake,
              v_opac_act=fake, v_ctab=fake)
    for bad, word in ((dict(isect_grad=fake + 8), b"16-byte"), (dict(isect_live=fake + 2), b"4-byte"), (dict(isect_live=0), b"NULL")):
        rg = L.RasterGrads(**{**ok, **bad})
        assert lib.d4gs_raster_bwd(C.byref(d), C.byref(pout), C.byref(isect), C.byref(ras), C.byref(rg), None) == -1
        assert word in lib.d4gs_last_error(), lib.d4gs_last_error()


def test_exact_tiles_and_alignment_contracts_are_checked_before_any_hip_call(lib):
    """(a) D4GS_EXACT_TILES without D4gsProjOut.tile_masks in d4gs_bin_sort / d4gs_raster_fwd / d4gs_raster_bwd - the counts
    and offsets were built from popcount(mask), rectangles walked against them would write out of bounds; (b) ctab / v_ctab are read
    and written as 16-byte words by d4gs_project_fwd / d4gs_project_bwd.  Both are host-side D4GS_EINVAL (fake addresses)."""
    from deblur4dgs_amd import _lib as L

    lib.d4gs_last_error.restype = C.c_char_p
    fake = 0x10000
    d = L.Dims(N=10, G=0, K=0, T=0, S=1, D=3, width=16, height=16, flags=L.EXACT_CULL | L.EXACT_TILES)
    pout = L.ProjOut(**{**{n: fake for n, _ in L.ProjOut._fields_}, "tile_masks": 0})
    isect = L.Isect(n_isect=4, max_tile_count=0, keys=fake, gid_of_emit=fake, sorted_gid=fake, sorted_emit=fake)
    ras = L.Raster(**{n: fake for n, _ in L.Raster._fields_})
    rg = L.RasterGrads(v_render_colors=fake, isect_grad=fake, isect_live=fake, v_means2d=fake, v_conics=fake, v_depths=fake,
                       v_opac_act=fake, v_ctab=fake)
    for call in (lambda: lib.d4gs_bin_sort(C.byref(d), C.byref(pout), C.byref(isect), None),
                 lambda: lib.d4gs_raster_fwd(C.byref(d), C.byref(pout), C.byref(isect), C.byref(ras), None),
                 lambda: lib.d4gs_raster_bwd(C.byref(d), C.byref(pout), C.byref(isect), C.byref(ras), C.byref(rg), None)):
        assert call() == -1 and b"tile_masks" in lib.d4gs_last_error(), lib.d4gs_last_error()
    d = L.Dims(N=10, G=0, K=0, T=0, S=1, D=3, width=16, height=16)
    pin = L.ProjIn(**{n: fake for n in ("means", "quats", "scales", "opacities", "colors", "viewmat", "Kmat")})
    pout = L.ProjOut(**{**{n: fake for n, _ in L.ProjOut._fields_}, "ctab": fake + 4})
    assert lib.d4gs_project_fwd(C.byref(d), C.byref(pin), C.byref(pout), None) == -1 and b"16-byte" in lib.d4gs_last_error()
    lg = L.LeafGrads(**{n: fake for n in ("v_means", "v_quats", "v_scales", "v_opacities", "v_colors", "partials")})
    vp = C.c_void_p
    for pout, v_ctab in ((pout, fake), (L.ProjOut(**{n: fake for n, _ in L.ProjOut._fields_}), fake + 8)):
        assert lib.d4gs_project_bwd(C.byref(d), C.byref(pin), C.byref(pout), vp(fake), vp(fake), vp(fake), vp(fake), vp(v_ctab),
                                    C.byref(lg), None) == -1
        assert b"16-byte" in lib.d4gs_last_error(), lib.d4gs_last_error()


def test_last_error_is_thread_local(lib):
    """SURVEY 8b's threading contract, host half (no GPU): d4gs_last_error() is per thread.  Two threads fail with DIFFERENT messages at
    the same moment, a third never fails: each reads its own string (ctypes releases the GIL around the calls)."""
    import threading

    from deblur4dgs_amd import _lib as L

    lib.d4gs_last_error.restype = C.c_char_p
    bar = threading.Barrier(3)
    seen = {}

    def worker(name):
        bar.wait()
        for _ in range(200):
            if name == "null_dims":
                assert lib.d4gs_project_fwd(None, None, None, None) == -1
            elif name == "motion_dims":
                d = L.Dims(N=10, G=20, K=1, T=1, S=1, D=3, width=16, height=16)
                assert lib.d4gs_project_fwd(C.byref(d), None, None, None) == -1
            msg = lib.d4gs_last_error()
            seen.setdefault(name, set()).add(msg)

    th = [threading.Thread(target=worker, args=(n,)) for n in ("null_dims", "motion_dims", "quiet")]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert seen["quiet"] == {b""}
    assert len(seen["null_dims"]) == 1 and b"NULL" in next(iter(seen["null_dims"]))
    assert len(seen["motion_dims"]) == 1 and b"motion dims" in next(iter(seen["motion_dims"]))


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
            decl = re.sub(r"^(const\s+)?(float|int32_t|int64_t|uint64_t|uint8_t|void)\s*", "", decl)
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


def test_query_sizes_is_a_pure_host_call(lib):
    """d4gs_query_sizes (the workspace query of SURVEY 8b) needs no GPU: element counts of every caller-owned buffer."""
    from deblur4dgs_amd import _lib as L

    lib.d4gs_query_sizes.argtypes = [C.POINTER(L.Dims), C.POINTER(L.Sizes)]
    d = L.Dims(N=1000, G=400, K=6, T=24, S=8, D=3, width=100, height=50, depth_mode=L.DEPTH_ED)
    z = L.Sizes()
    assert lib.d4gs_query_sizes(C.byref(d), C.byref(z)) == 0
    SN, tw, th = 8 * 1000, 7, 4
    assert (z.tiles_x, z.tiles_y, z.channels) == (tw, th, 4)
    assert (z.means2d, z.depths, z.conics, z.radii) == (SN * 2, SN, SN * 3, SN)
    assert (z.opac_act, z.ctab, z.geom) == (1000, 1000 * 4, SN * L.GEOM_STRIDE)
    assert (z.tile_rects, z.tiles_touched, z.isect_offsets) == (SN * 2, SN, SN)
    assert (z.tile_counts, z.tile_offsets, z.n_isect) == (2 * 8 * tw * th, 8 * tw * th + 1, 4)
    assert z.render_colors == 8 * 50 * 100 * 4 and z.render_alphas == z.last_ids == z.final_T == 8 * 50 * 100
    assert z.isect_grad_row == 6 + 4
    lib.d4gs_scan_ws_elems.restype = C.c_size_t
    lib.d4gs_scan_ws_elems.argtypes = [C.c_int64]
    lib.d4gs_bwd_partials_elems.restype = C.c_size_t
    assert z.scan_ws == lib.d4gs_scan_ws_elems(SN) and z.bwd_partials == lib.d4gs_bwd_partials_elems(C.byref(d))
    bad = L.Dims(N=10, G=20, K=1, T=1, S=1, D=3, width=16, height=16)
    assert lib.d4gs_query_sizes(C.byref(bad), C.byref(z)) == -1 and lib.d4gs_query_sizes(C.byref(d), None) == -1


def test_one_call_entry_points_validate_before_any_hip_call(lib):
    from deblur4dgs_amd import _lib as L

    lib.d4gs_last_error.restype = C.c_char_p
    lib.d4gs_frame_workspace_bytes.restype = C.c_size_t
    lib.d4gs_frame_workspace_bytes.argtypes = [C.POINTER(L.Dims), C.c_int64]
    d = L.Dims(N=1000, G=0, K=0, T=0, S=2, D=3, width=64, height=48, depth_mode=1)
    small, big = lib.d4gs_frame_workspace_bytes(C.byref(d), 0), lib.d4gs_frame_workspace_bytes(C.byref(d), 100000)
    assert 0 < small < big and small % 256 == 0 and big - small >= 100000 * (8 + 12 + 4 * 10)  # keys, ids, gradient rows
    assert lib.d4gs_frame_workspace_bytes(C.byref(L.Dims(N=-1, S=1, D=3, width=8, height=8)), 0) == 0
    lib.d4gs_forward.argtypes = [C.POINTER(L.Dims), C.POINTER(L.ProjIn), C.POINTER(L.FrameIO), C.c_void_p, C.c_size_t, C.c_int64,
                                 C.c_int64, C.c_void_p]
    fake = 0x10000
    pin = L.ProjIn(**{n: fake for n, _ in L.ProjIn._fields_})
    io = L.FrameIO(**{n: fake for n in ("renders", "alphas", "means2d", "radii", "n_isect")})
    # the forward alone needs only the prefix of the workspace in front of the backward's scratch (validation / viewer renders)
    lib.d4gs_frame_workspace_bytes_fwd.restype = C.c_size_t
    lib.d4gs_frame_workspace_bytes_fwd.argtypes = [C.POINTER(L.Dims), C.c_int64]
    fwd_small = lib.d4gs_frame_workspace_bytes_fwd(C.byref(d), 0)
    assert 0 < fwd_small < small and fwd_small % 256 == 0
    assert small - fwd_small >= 4 * (2 * 64 * 48 * 5 + 2 * 1000 * 4)  # at least the image-gradient stack and the per-instance rows
    assert lib.d4gs_forward(C.byref(d), C.byref(pin), C.byref(io), C.c_void_p(fake), fwd_small - 1, 0, 0, None) == -3  # D4GS_ECAPACITY
    assert b"workspace" in lib.d4gs_last_error()
    assert lib.d4gs_forward(C.byref(d), C.byref(pin), C.byref(io), C.c_void_p(fake + 8), small, 0, 0, None) == -1
    assert b"aligned" in lib.d4gs_last_error()
    io2 = L.FrameIO(**{n: fake for n in ("renders", "alphas", "means2d", "radii")})  # n_isect missing
    assert lib.d4gs_forward(C.byref(d), C.byref(pin), C.byref(io2), C.c_void_p(fake), small, 0, 0, None) == -1
