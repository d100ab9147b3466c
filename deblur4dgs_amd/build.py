"""Build libd4gs.so (HIP, gfx950 only) in-tree with hipcc.  `python -m deblur4dgs_amd.build [--force]` (also builds the
tests' A/B library tests/libd4gs_variants.so).

hipcc cross-compiles gfx950 code objects without a GPU, so this runs in the build container; the resulting
.so is git-ignored but travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "_obj")
LIB = os.path.join(HERE, "libd4gs.so")
# the linker keeps `d4gs_*` and drops the rest from the dynamic symbol table (libstdc++ template instantiations and hipcc's
# __hip_cuid_* markers are emitted with default visibility whatever -fvisibility says)
LINK = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + os.path.join(CSRC, "libd4gs.map")]
# A/B build for the tests only: the same sources with -DD4GS_VARIANTS, which adds the non-default composite kernels
# (csrc/variants/*.inc, selected by D4GS_{FWD,BWD}_* environment variables).  Never loaded by the package itself.
VARIANTS_LIB = os.path.join(HERE, "..", "tests", "libd4gs_variants.so")
VARIANT_SOURCES = ("raster_fwd.hip", "raster_bwd.hip")
# -fvisibility=hidden: only the D4GS_API entry points of include/d4gs.h are exported (tests/test_c_abi.py checks `nm -D`).  No
# -munsafe-fp-atomics: the library issues no floating-point atomics at all (the gradient reductions are deterministic gathers).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]
# per-file extras.  project_bwd: the SLP vectorizer turns the adjoint chain into v_pk_* math, which is not faster on gfx950 (4.5
# cycles for two operations against 2.5 for one) and parks ~40 duplicated operands in VGPR pairs: 256 + 30 registers instead of 128
FILE_FLAGS = {"project_bwd.hip": ["-fno-slp-vectorize"],
              # raster_fwd: the four unrolled composite steps per list read pair up into v_pk_* math behind a dozen v_mov shuffles
              "raster_fwd.hip": ["-fno-slp-vectorize"]}


def _cuid(src: str) -> str:
    """hipcc derives a compilation-unit id from the ABSOLUTE source path; a fixed one makes libd4gs.so byte-identical wherever the
    tree is checked out (profiles/pmc_current.json pins the counter files to the library's sha256)."""
    return "-cuid=d4gs_" + os.path.splitext(src)[0]


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(src: str, obj: str) -> bool:
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    deps = [os.path.join(CSRC, src), os.path.join(CSRC, "common.h"), os.path.join(HERE, "..", "include", "d4gs.h")]
    deps += [os.path.join(CSRC, "variants", f) for f in os.listdir(os.path.join(CSRC, "variants"))] if obj.endswith(".var.o") else []
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src: str, extra: list[str], suffix: str = ".o") -> str:
    obj = os.path.join(OBJ, src.replace(".hip", suffix))
    if _stale(src, obj) or (extra and suffix == ".o"):
        cmd = ["hipcc", *FLAGS, *FILE_FLAGS.get(src, []), _cuid(src), *extra, "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr}")
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
    return obj


def build(force: bool = False, extra: list[str] | None = None) -> str:
    os.makedirs(OBJ, exist_ok=True)
    extra = list(extra or [])
    srcs = _sources()
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    with cf.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, extra), srcs))
    if force or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [*LINK, "-o", LIB, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
    return LIB


def build_variants() -> str:
    """tests/libd4gs_variants.so: raster_fwd / raster_bwd recompiled with -DD4GS_VARIANTS, every other object shared with
    the product build (call build() first)."""
    build()
    objs = []
    for s in _sources():
        objs.append(_compile(s, ["-DD4GS_VARIANTS"], ".var.o") if s in VARIANT_SOURCES else os.path.join(OBJ, s.replace(".hip", ".o")))
    if not os.path.exists(VARIANTS_LIB) or any(os.path.getmtime(o) > os.path.getmtime(VARIANTS_LIB) for o in objs):
        r = subprocess.run([*LINK, "-o", VARIANTS_LIB, *objs], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
    return VARIANTS_LIB


def build_ab(name: str, src: str, defines: list[str]) -> str:
    """scripts/ablate/libd4gs_<name>.so: the product objects with `src` (one file, or several separated by commas) recompiled
    under extra -D flags (A/B timing of one kernel on the GPU box: `D4GS_LIB_PATH=scripts/ablate/libd4gs_<name>.so python bench.py ...`)."""
    build()
    out_dir = os.path.join(HERE, "..", "scripts", "ablate")
    os.makedirs(out_dir, exist_ok=True)
    ab = {}
    for one in src.split(","):
        obj = os.path.join(out_dir, f"ab_{name}_{os.path.splitext(one)[0]}.o")
        cmd = ["hipcc", *FLAGS, *FILE_FLAGS.get(one, []), _cuid(one), *defines, "-c", os.path.join(CSRC, one), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {one}:\n{r.stderr}")
        ab[one] = obj
    objs = [ab.get(s) or os.path.join(OBJ, s.replace(".hip", ".o")) for s in _sources()]
    lib = os.path.join(out_dir, f"libd4gs_{name}.so")
    r = subprocess.run([*LINK, "-o", lib, *objs], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr}")
    return lib


if __name__ == "__main__":
    if len(sys.argv) > 3 and sys.argv[1] == "--ab":  # python -m deblur4dgs_amd.build --ab <name> <src.hip> [-DX=1 ...]
        print(build_ab(sys.argv[2], sys.argv[3], sys.argv[4:]))
    else:
        print(build(force="--force" in sys.argv))
        print(build_variants())
