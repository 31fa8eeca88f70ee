"""Multi-process data-parallel path on CPU: 2 ranks over gloo (the GPU runs use the same code over RCCL)."""
import os
import subprocess
import sys


import pytest


@pytest.mark.parametrize("mode", ["ddp", "flat"])
def test_two_rank_gloo_gradients_are_averaged(tmp_path, mode):
    """mode 'ddp': torch's reducer per phase (the CPU default); 'flat': Trainer.FlatGradients -- one gradient buffer, one all-reduce
    behind backward() -- what eager steps use on the GPU (round 5; the replayed steps have averaged that way since round 3)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "result.txt")
    env = dict(os.environ, OMP_NUM_THREADS="2", DD_EAGER_REDUCE=mode)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533" if mode == "ddp" else "29537", os.path.join(root, "tests", "ddp_worker.py"), out]
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:]
    assert open(out).read().startswith("OK"), res.stdout[-2000:]
