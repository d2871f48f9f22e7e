"""bench.py's N > 1 control flow on ONE GPU: two ranks launched by torch.distributed.run share cuda:0
(BENCH_SHARE_GPU=1) and exchange their logits through gloo (BENCH_DIST_BACKEND=gloo) - mode agreement across ranks,
the asynchronous double-buffered logits gather, barrier + max-over-ranks timing and the whole-job aggregate all run
exactly as they do with one GPU per rank over RCCL (which needs a multi-GPU node the driver owns)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_two_ranks_share_one_gpu():
    env = dict(os.environ)
    env.update(BENCH_DIST_BACKEND="gloo", BENCH_SHARE_GPU="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "3",
           "--no-cpu-baseline", "--no-b1"]
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]          # rank 0 prints ONE line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["config"]["global_batch"] == 16
    assert out["value"] > 0 and abs(out["value"] - 16 * 1000.0 / out["ms_per_step"]) < 0.02 * out["value"]
    assert out["config"]["parallelism"] == "batch-shard x2"
