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
            ["adam.pth", "resume.json", "rng.pth"] + [m + ".pth" for m in ("depth_enc", "depth_dec", "pose_enc", "pose_dec", "motion_enc", "motion_dec", "motion_mask")])
    assert json.load(open(models / "fine_tune_00" / "resume.json"))["phase"] == "fine_tune"


def _train(tmp, name, extra, schedules):
    cmd = [sys.executable, "train.py", "-d", "kitti", "--synthetic", "--weights_init", "scratch", "-b", "2", "--height", "64", "--width", "96",
           "--epoch-size", "3", "--epoch_schedules"] + [str(e) for e in schedules] + [
           "--log_frequency", "2", "--num_workers", "0", "--log_dir", str(tmp), "-n", name, "--depth_model", "litemono", "--channels_last"] + extra
    env = dict(os.environ, MIOPEN_LOG_LEVEL="2")
    res = subprocess.run(cmd, cwd=os.path.join(ROOT, "dynamo-depth_amd"), env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-4000:]
    return res.stdout


def _max_diff(tmp, run_a, run_b, folder):
    import torch
    worst = 0.0
    for name in ("depth_enc", "depth_dec", "pose_enc", "pose_dec", "motion_enc", "motion_dec", "adam"):
        a = torch.load(tmp / run_a / "models" / folder / (name + ".pth"), map_location="cpu")
        b = torch.load(tmp / run_b / "models" / folder / (name + ".pth"), map_location="cpu")
        if name == "adam":
            a = {(i, k): v for i, st in a["state"].items() for k, v in st.items()}
            b = {(i, k): v for i, st in b["state"].items() for k, v in st.items()}
        assert a.keys() == b.keys()
        for k in a:
            if torch.is_tensor(a[k]):
                worst = max(worst, float((a[k].double() - b[k].double()).abs().max()))
    return worst


def test_resume_continues_the_run(tmp_path):
    """a run resumed from its epoch-0 checkpoint (weights, Adam moments, StepLR, counters, random streams)
    continues the uninterrupted run -- across an epoch boundary AND a phase boundary (disp_init x2 -> motion_init).

    Yardstick: the same command run twice in two processes.  MIOpen's solver search is timing-based, so two processes may
    pick different (equally valid) convolution kernels and differ in the last bits; the resumed run has to stay within that
    band -- four orders of magnitude below what a lost Adam state or a restarted random stream costs (one Adam step moves
    a weight by ~1e-4)."""
    schedules = [2, 1, 0, 0]
    _train(tmp_path, "straight", [], schedules)
    _train(tmp_path, "again", [], schedules)
    ckpt = tmp_path / "straight" / "models" / "disp_init_00"
    record = json.load(open(ckpt / "resume.json"))
    assert record["phase"] == "disp_init" and record["epoch"] == 0 and record["step"] == 3 and "scheduler" in record
    assert (ckpt / "rng.pth").exists()
    out = _train(tmp_path, "resumed", ["--resume", str(ckpt)], schedules)
    assert "resumed disp_init after epoch 0" in out, out[-2000:]
    for folder in ("disp_init_01", "motion_init_00"):
        control = _max_diff(tmp_path, "straight", "again", folder)
        resumed = _max_diff(tmp_path, "straight", "resumed", folder)
        print("%s: two identical runs differ by %.3e, the resumed run by %.3e" % (folder, control, resumed))
        assert resumed <= max(4.0 * control, 5e-6), (folder, control, resumed)


def _train_ddp(tmp, name, extra, schedules, port):
    """Two ranks on cuda:0 over gloo (RCCL refuses two ranks on one device): the launch line of the reference's README with
    torchrun, the per-phase DDP wrapper, the multi-stream communication hook and -- by default -- the per-network hipGraphs
    with their own gradient all-reduce."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           "train.py", "-d", "kitti", "--synthetic", "--weights_init", "scratch", "-b", "2", "--height", "64", "--width", "96", "--epoch-size", "3",
           "--epoch_schedules"] + [str(e) for e in schedules] + ["--log_frequency", "2", "--num_workers", "0", "--log_dir", str(tmp), "-n", name,
           "--depth_model", "litemono", "--dist_backend", "gloo", "--cuda_ids", "0", "0"] + extra
    env = dict(os.environ, MIOPEN_LOG_LEVEL="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run(cmd, cwd=os.path.join(ROOT, "dynamo-depth_amd"), env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
    assert res.returncode == 0, res.stdout[-5000:]
    return res.stdout


@pytest.mark.parametrize("graph", [False, True])
def test_two_rank_run_resumes_across_a_phase_switch(tmp_path, graph):
    """a two-rank run is stopped after disp_init and resumed into motion_init: the per-phase wrapper
    (static graph, frozen out-of-phase parameters), the stream-joining communication hook and, with graph=True, the captured
    per-network graphs with their own all-reduce all come up again in the new phase; every rank restores ITS random streams."""
    extra = [] if graph else ["--no_hip_graph"]
    _train_ddp(tmp_path, "first", extra, [1, 0, 0, 0], 29561 + int(graph))
    ckpt = tmp_path / "first" / "models" / "disp_init_00"
    assert (ckpt / "rng.pth").exists() and (ckpt / "rng_rank1.pth").exists() and (ckpt / "resume.json").exists()
    out = _train_ddp(tmp_path, "second", extra + ["--resume", str(ckpt)], [1, 1, 0, 0], 29571 + int(graph))
    assert "MOTION_INIT - finished before the resumed checkpoint" not in out, out[-3000:]      # the phase behind the checkpoint runs
    assert "resumed disp_init after epoch 0" in out, out[-3000:]
    assert (tmp_path / "second" / "models" / "motion_init_00" / "motion_dec.pth").exists()
    assert out.count("examples/s") >= 2, out[-2000:]
