"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads, and exports every symbol
include/saber_hip.h declares. No compute (no GPU here)."""
import os
import re

import pytest

from anakin_amd import build as B
from anakin_amd import lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    if not os.path.exists(L.LIB_PATH) or os.path.exists("/opt/rocm/bin/hipcc"):
        B.build()
    return L.load()


def test_header_symbols_are_exported(built):
    hdr = open(os.path.join(ROOT, "include", "saber_hip.h")).read()
    declared = set(re.findall(r"\b(saber_hip_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(L.SYMBOLS), (declared ^ set(L.SYMBOLS))
    for name in sorted(declared):
        assert hasattr(built, name), name


def test_nothing_else_is_exported_with_c_linkage(built):
    """The ABI's translation units (api_*.hip) take their C linkage from the header: an entry point defined without a
    declaration there would silently become a mangled C++ symbol (and a helper declared in an extern "C" block an
    accidental export). The unmangled dynamic symbols of the library are exactly the header's."""
    import shutil
    import subprocess
    nm = shutil.which("nm") or "/opt/rocm/lib/llvm/bin/llvm-nm"
    out = subprocess.run([nm, "-D", "--defined-only", L.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.split()}
    plain = {n for n in exported if not n.startswith(("_Z", "__", "_init", "_fini", "_edata", "_end", "_bss"))}
    assert plain == set(L.SYMBOLS), (plain ^ set(L.SYMBOLS))


def test_create_validates_without_gpu(built):
    import ctypes as C
    d = L.ConvDesc()
    h = C.c_void_p()
    assert built.saber_hip_conv2d_create(C.byref(d), C.byref(h)) == -2  # SaberInvalidValue: empty geometry
    assert b"geometry" in built.saber_hip_last_error()
    d.n, d.h, d.w, d.c, d.k, d.kh, d.kw = 1, 8, 8, 6, 9, 3, 3
    d.stride_h = d.stride_w = d.dil_h = d.dil_w = 1
    d.group = 4
    assert built.saber_hip_conv2d_create(C.byref(d), C.byref(h)) == -2  # c % group != 0
    assert built.saber_hip_pool_out_dim(112, 0, 3, 2, 0) == 56      # ceil mode, pooling.h:109-115
    assert built.saber_hip_pool_out_dim(112, 0, 3, 2, 1) == 55
    assert built.saber_hip_pool_out_dim(224, 1, 3, 2, 0) == 113 - 0


def test_executor_level_objects_validate_their_arguments_without_gpu(built):
    """the fused-launch objects of round 4 (stem + pair, strided head + pair) refuse null / unset operators with a status, no device needed"""
    import ctypes as C
    h = C.c_void_p()
    assert built.saber_hip_conv2d_stem_pair_create(None, None, None, C.byref(h)) == -2
    assert built.saber_hip_conv2d_chain_create3_pair(None, None, None, None, C.byref(h)) == -2
    d = L.ConvDesc()
    d.n, d.h, d.w, d.c, d.k, d.kh, d.kw = 1, 16, 16, 64, 64, 1, 1
    d.stride_h = d.stride_w = d.dil_h = d.dil_w = d.group = 1
    d.in_dtype, d.out_dtype, d.in_layout, d.out_layout, d.int8_weights = L.U8, L.U8, L.NHWC, L.NHWC, 1
    c = C.c_void_p()
    assert built.saber_hip_conv2d_create(C.byref(d), C.byref(c)) == 0
    assert built.saber_hip_conv2d_stem_pair_create(c, c, c, C.byref(h)) == -2      # not a fused stem conv + pooling, no weights
    assert b"stem pair" in built.saber_hip_last_error()
    assert built.saber_hip_conv2d_stem_pair_run(None, None, None, None, None, None, None) == -2
    built.saber_hip_conv2d_destroy(c)


def test_algorithm_selection_is_host_side(built):
    """create() picks the kernel family from shapes alone (no device needed)."""
    import ctypes as C

    def algo(c, k, kh, int8, in_dt, out_dt, n=8, hw=14, group=1, in_layout=L.NHWC, out_layout=L.NHWC):
        d = L.ConvDesc()
        d.n, d.h, d.w, d.c, d.k, d.kh, d.kw = n, hw, hw, c, k, kh, kh
        d.pad_h = d.pad_w = kh // 2
        d.stride_h = d.stride_w = d.dil_h = d.dil_w = 1
        d.group = group
        d.in_dtype, d.out_dtype, d.in_layout, d.out_layout, d.int8_weights = in_dt, out_dt, in_layout, out_layout, int8
        h = C.c_void_p()
        assert built.saber_hip_conv2d_create(C.byref(d), C.byref(h)) == 0, built.saber_hip_last_error()
        name = built.saber_hip_conv2d_algo(h).decode()
        built.saber_hip_conv2d_destroy(h)
        return name

    assert algo(256, 256, 3, 1, L.U8, L.U8).startswith("igemm_i8_")
    assert algo(3, 64, 7, 1, L.F32, L.U8, hw=224, in_layout=L.NCHW).startswith("igemm_i8_c4_")
    assert algo(64, 64, 3, 0, L.F32, L.F32, in_layout=L.NCHW, out_layout=L.NCHW).startswith("igemm_f32_")
    assert algo(32, 32, 3, 1, L.U8, L.U8, group=4) == "direct_i8"
    assert algo(20, 24, 5, 1, L.S8, L.F32) == "direct_i8"


def test_descriptors_and_selection_codes_fuzzed_without_gpu(built):
    """tests/fuzz_desc.py in a subprocess (a crash would take pytest with it): 800 random convolution / fc descriptors (edge values in every field:
    negative, zero, 2^31 - 1, NaN / inf coefficients), pooling shape queries with zero strides, then 2 000 plausible convolutions with twelve random
    kernel-selection codes each through saber_hip_conv2d_set_tile - every call returns a status (the reference's contract: SaberInvalidValue /
    SaberUnImplError, saber_types.h:223-233), none crashes, hangs or aborts."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_desc.py"), "7", "800"], capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    assert "plausible convs: created" in r.stdout
    created = int(r.stdout.split("plausible convs: created")[1].split(",")[0])
    assert created > 500, r.stdout[-500:]
