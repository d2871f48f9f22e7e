"""Runs the C++ parity test (tests/cpp/test_saber_conv_int8.cpp: the Saber interface driven from C++,
checked bit for bit against the oracle library) on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "test_saber_conv_int8.bin")


@pytest.mark.gpu
def test_cpp_saber_conv_int8():
    assert os.path.exists(BIN), "tests/cpp/test_saber_conv_int8.bin is missing: run __graft_entry__.build()"
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.pathsep.join([os.path.join(ROOT, "anakin_amd"), os.path.join(ROOT, "oracle"),
                                              env.get("LD_LIBRARY_PATH", "")])
    p = subprocess.run([BIN], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    assert "cases bit-exact" in p.stdout


def test_cpp_test_binary_is_built():
    """CPU-side: the C++ test builds (hipcc host compile + link against both libraries)."""
    from anakin_amd import build as B
    B.build_cpp_tests()
    assert os.path.exists(BIN)


MI355X_BIN = os.path.join(ROOT, "integration", "_build", "test_saber_conv_mi355x.bin")


@pytest.mark.gpu
def test_mi355x_target_inside_the_reference_operator_stack(tmp_path):
    """The MI355X target executed by the reference's OWN Saber code (b, f-4 saber half):
    integration/test_saber_conv_mi355x.cpp is compiled against a patched copy of /root/reference/saber
    (integration/apply_mi355x_target.py: eMI355X, TargetWrapper<MI355X> on HIP, Device / Env / Context, SaberTimer,
    the Conv facade ladder) and drives Conv<MI355X,AK_INT8>::init -> BaseFunc::operator() (incl. the shape-change
    re-create path, base.h:151-161) -> SaberConv2D<MI355X> -> integration/saber_mi355x_adaptor.h -> the C ABI;
    every output byte equals the oracle's."""
    assert os.path.exists(MI355X_BIN), "integration/_build/test_saber_conv_mi355x.bin is missing: run __graft_entry__.build()"
    p = subprocess.run([MI355X_BIN], capture_output=True, text=True, timeout=600, cwd=str(tmp_path))   # (the reference's logger writes ./log/)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    assert "0 failed" in p.stdout and "bit-exact" in p.stdout and "SaberTimer<MI355X>" in p.stdout
    for kind in ("sigmoid", "swish", "prelu"):       # Activation<MI355X, AK_FLOAT> beyond relu, under the reference's BaseFunc
        assert "ok   Activation<MI355X,AK_FLOAT> %s" % kind in p.stdout, p.stdout[-2000:]


def test_mi355x_target_test_binary_is_built():
    """CPU-side: where the reference tree is available, the patched Saber tree + test build (g++ against the reference's
    headers, linked to the HIP library); elsewhere the prebuilt binary must be present."""
    script = os.path.join(ROOT, "integration", "build_mi355x_test.sh")
    if os.path.isdir("/root/reference/saber"):
        subprocess.check_call(["bash", script])
    assert os.path.exists(MI355X_BIN)
