"""The DDP training step on the GPU: 2 ranks (gloo, both on cuda:0), LiteMono channels-last with all HIP hooks + the fused loss."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_two_rank_training_steps_keep_weights_in_sync(tmp_path):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "result.txt")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29537", os.path.join(root, "tests", "ddp_worker_gpu.py"), out]
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-4000:]
    assert open(out).read().startswith("OK"), (open(out).read(), res.stdout[-2000:])
