"""Two ranks through FusedTrainStep on ONE GPU (both on cuda:0, gloo process group): the multi-rank control flow
(bucketed all-reduce on the side stream, global-count loss normalisation, a rank with an empty batch still joining
the exchange) must reproduce the single-rank step on the concatenated batch.  RCCL itself is exercised by the
driver's multi-GPU bench; here the point is the arithmetic and that no rank deadlocks."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def test_two_rank_fused_step_equals_one_rank_step(tmp_path):
    ok = tmp_path / "ok"
    port = 29500 + os.getpid() % 2000
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(HERE, "ddp_worker.py"), str(ok)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    print(r.stdout[-2000:], r.stderr[-3000:])
    assert r.returncode == 0
    assert ok.read_text() == "ok"
