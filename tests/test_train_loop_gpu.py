"""train.py end to end on synthetic triplets: all four phases, log steps (materialised outputs), validation with
DepthMetrics, checkpoints in the reference's per-module layout.  GPU only."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_four_phase_schedule_runs(tmp_path):
    cmd = [sys.executable, "train.py", "-d", "kitti", "--synthetic", "--weights_init", "scratch", "-b", "2", "--height", "64", "--width", "96",
           "--epoch-size", "4", "--epoch_schedules", "1", "1", "1", "1", "--log_frequency", "2", "--num_workers", "0",
           "--log_dir", str(tmp_path), "-n", "smoke", "--depth_model", "monodepthv2"]
    env = dict(os.environ, MIOPEN_LOG_LEVEL="2")
    res = subprocess.run(cmd, cwd=os.path.join(ROOT, "dynamo-depth_amd"), env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-4000:]
    assert res.stdout.count("examples/s") >= 4, res.stdout[-2000:]
    models = tmp_path / "smoke" / "models"
    assert (models / "opt.json").exists()
    for phase in ("disp_init", "motion_init", "mask_init", "fine_tune"):
        folder = models / "{}_00".format(phase)
        assert sorted(p.name for p in folder.iterdir()) == sorted(
            ["adam.pth", "resume.json"] + [m + ".pth" for m in ("depth_enc", "depth_dec", "pose_enc", "pose_dec", "motion_enc", "motion_dec", "motion_mask")])
    assert json.load(open(models / "fine_tune_00" / "resume.json"))["phase"] == "fine_tune"
