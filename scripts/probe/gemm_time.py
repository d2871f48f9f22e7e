"""saber_hip_gemm_f32 timing probe: 2048^3 (the bench's shape) and VGG16's fc6 as a GEMM, plane path on / off
(SABER_HIP_GEMM_F32_PLANES=0|1 forces the path). Usage: python scripts/probe/gemm_time.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def one():
    import torch
    from anakin_amd import saber as S
    for (m, n, k, tb) in ((2048, 2048, 2048, False), (4096, 4096, 4096, False), (1024, 1024, 1024, False), (512, 4096, 4096, True),
                          (64, 4096, 25088, True), (8, 4096, 25088, True)):
        a = torch.randn(m, k, device="cuda")
        b = torch.randn((n, k) if tb else (k, n), device="cuda")
        c = torch.empty(m, n, device="cuda")
        for _ in range(3):
            S.gemm(False, tb, m, n, k, 1.0, a, b, 0.0, c)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            S.gemm(False, tb, m, n, k, 1.0, a, b, 0.0, c)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 10
        print("planes=%s m=%d n=%d k=%d tb=%d: %.1f us, %.1f TFLOP/s, B stream %.0f GB/s" % (
            os.environ.get("SABER_HIP_GEMM_F32_PLANES", "auto"), m, n, k, tb, us, 2.0 * m * n * k / us / 1e6, n * k * 4.0 / us / 1e3))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        one()
    else:
        for p in ("1", "0"):
            subprocess.run([sys.executable, os.path.abspath(__file__), "x"], env=dict(os.environ, SABER_HIP_GEMM_F32_PLANES=p))
