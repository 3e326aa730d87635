"""The N-rank path of bench.py on the one GPU a test box has: two ranks under torch.distributed.run share GPU 0, render
their row strips of every frame with the library, exchange the packed rows (over gloo through host memory: RCCL wants
one GPU per rank), and rank 0 requires the completed frame to equal, bit for bit, the same frames rendered whole by
one context on the single kernel, one frame per call (the script's own epilogue).  What it covers beyond the other tests: the script's own orchestration with more than one rank (strip
ownership by rank, pack / gather / unpack order across the two streams, the ranks' reductions, rank 0's result line)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_with_two_ranks_completes_the_frame_bit_for_bit():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse", "--steps", "1", "--warmup", "0",
           "--no-cpu-baseline"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=100)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "bit-identical: True" in r.stderr, r.stderr[-2000:]  # rank 0's untimed check of the frames the ranks completed
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout  # rank 0 prints ONE line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 1 and out["scaling"] == "strong" and out["value"] > 0
    assert "row strips" in out["config"]["partition"] and out["verified"] is True
