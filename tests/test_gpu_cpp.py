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
