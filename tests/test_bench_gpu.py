"""bench.py's contract on a GPU box: the one-line JSON of the N=1 run, and the N>1 code path (auto probe decided on the slowest
rank, max-over-ranks timing, per-rank host time) with two ranks sharing cuda:0 over gloo -- RCCL refuses two ranks on one
device, so DD_BENCH_BACKEND=gloo stands in for it; a smoke test of the code path, not a measurement."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--steps", "3", "--warmup", "2", "--batch", "2", "--no_cpu_baseline", "--no_miopen_find"]


def last_json_line(text):
    lines = [ln for ln in text.splitlines() if ln.startswith("{")]
    assert lines, text[-3000:]
    return json.loads(lines[-1])


def test_single_gpu_line_carries_the_roofline_of_the_replayed_step():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "graph"] + SMALL, env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-4000:]
    line = last_json_line(res.stdout)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["steps"] == 3 and line["value"] > 0 and line["dtype"] == "f32" and line["config"]["mode"] == "graph"
    roof = line["roofline"]
    assert roof["bound"] == "hbm" and roof["peak"] == 8000.0 and 0 < roof["frac"] < 1 and 0 < roof["frac_loss_path"] <= roof["frac"]
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    # the timed step is train.py's (the loss is ONE graph); tile kernel and loss path are timed by host-issued evaluations on the
    # last step's buffers directly behind the timed region, the loss graph's stream time inside the timed steps themselves
    assert "host-issued evaluations" in roof["timed_in"], roof["timed_in"]
    assert roof["launches_timed"] >= 3
    # `frac` is quoted on SURVEY 8(d)'s synthetic network outputs; the step's own (random-init: identity warps) and the per-pixel
    # white-noise draw are reported beside it, and the reference's CPU timing is read from the committed record
    assert roof["workload"].startswith("SURVEY 8(d)"), roof["workload"]
    for tag in ("identity_warps", "white_noise_fields"):
        assert 0 < roof["frac_" + tag] < 1 and roof["avg_launch_us_" + tag] > 0 and 0 < roof["frac_loss_path_" + tag] <= roof["frac_" + tag], (tag, roof)
    assert roof["loss_path_replayed_us"] > roof["avg_launch_us"] and roof["loss_path_us"] > roof["avg_launch_us"]
    assert line["config"]["capture_fallback"] is None and line["config"]["rccl_ranks"] == 0


def test_single_gpu_line_with_the_instrumented_split_loss():
    """DD_BENCH_SPLIT_LOSS=1 (round 3's instrumented step): graph | tile kernel launched by the host | graph, timed inside the timed region."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", DD_BENCH_SPLIT_LOSS="1")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "graph"] + SMALL, env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-4000:]
    roof = last_json_line(res.stdout)["roofline"]
    assert roof["timed_in"].startswith("timed region"), roof["timed_in"]
    assert roof["launches_timed"] == 3


def test_bench_path_over_rccl_with_one_rank():
    """the N > 1 branch of bench.py over REAL RCCL -- a process group of one rank (RCCL accepts it): captures in
    thread_local mode beside the live NCCL watchdog thread, all_reduce(AVG) of the flat gradient buffers behind the backward
    graphs on their side streams, the auto probe's cross-rank decision, max-over-ranks timing."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", DD_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29549")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-4000:]
    line = last_json_line(res.stdout)
    cfg = line["config"]
    assert cfg["rccl_ranks"] == 1 and cfg["dist_backend"] == "nccl"
    # the capture works next to RCCL: the probe has a replayed-step time and no fall-back reason (which of the two the probe then keeps
    # is its business: since round 5 the eager step no longer loses 16 % to torch's reducer beside a process group, and at this small
    # shape either may win)
    assert cfg["capture_fallback"] is None and cfg["auto_probe"]["graph_ms"] is not None and cfg["mode"] in ("graph", "eager"), cfg
    assert line["value"] > 0 and cfg["final_loss"] == cfg["final_loss"]


def test_two_rank_bench_path_on_one_gpu():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", DD_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29547",
           os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1200)
    assert res.returncode == 0, res.stdout[-4000:]
    line = last_json_line(res.stdout)
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 4 and line["config"]["parallelism"] == "dp2" and line["scaling"] == "weak"
    per_rank = line["config"]["host_enqueue_ms_per_rank"]
    assert len(per_rank) == 2 and all(x > 0 for x in per_rank)
    assert line["config"]["mode"] in ("graph", "eager") and line["config"]["auto_probe"] is not None
    assert line["value"] > 0 and line["config"]["final_loss"] == line["config"]["final_loss"]          # finite
    assert sum(1 for ln in res.stdout.splitlines() if ln.startswith("{")) == 1                         # rank 0 alone prints the line
