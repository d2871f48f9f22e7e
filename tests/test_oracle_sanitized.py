"""The parity checker itself under the sanitizers: oracle/saber_oracle.c (the CPU restatement every `np.array_equal` of the GPU tests rests on)
built with AddressSanitizer + UndefinedBehaviorSanitizer and run over the golden vectors the compiled reference produced (tests/golden/) and the
oracle-vs-reference comparisons - an out-of-bounds read, a signed overflow in the int32 accumulation or a misaligned access in the checker would
make "bit-exact" mean less than it says. CPU build only (the sanitizers are not available on the GPU pool)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _asan_runtime():
    p = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


@pytest.mark.skipif(_asan_runtime() is None, reason="gcc has no libasan here")
def test_oracle_under_asan_and_ubsan_reproduces_the_golden_vectors(tmp_path):
    so = str(tmp_path / "libsaber_oracle_san.so")
    subprocess.check_call(["gcc", "-O1", "-g", "-march=x86-64-v3", "-fopenmp", "-fPIC", "-std=c11", "-ffp-contract=off", "-fno-fast-math",
                           "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-shared", "-o", so, os.path.join(ROOT, "oracle", "saber_oracle.c"), "-lm"])
    env = dict(os.environ, SABER_ORACLE_LIB=so, LD_PRELOAD=_asan_runtime(), ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=97",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1:exitcode=98")
    tests = ["tests/test_oracle_golden.py", "tests/test_oracle_second_source.py"]
    if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libanakin_x86_ref.so")):
        tests.append("tests/test_oracle_vs_ref.py")      # (the compiled reference is not instrumented; the restatement beside it is)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + tests, env=env, cwd=ROOT, capture_output=True, text=True, timeout=1800)
    tail = r.stdout[-3000:] + r.stderr[-3000:]
    assert r.returncode == 0, tail
    assert "AddressSanitizer" not in tail and "runtime error" not in tail, tail
    assert " passed" in r.stdout
