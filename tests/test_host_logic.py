"""Host-side decisions that need no GPU: which find-db records apply to which device, the sweep of stale MIOpen
user-db copies, bench.py's bookkeeping (algorithmic bytes, the committed reference-CPU record, the all-reduce budget)."""
import os
import time

import Trainer as trainer_module
from Trainer import Trainer


def test_find_db_records_are_matched_against_device_and_miopen_build():
    db = os.path.join(os.path.dirname(os.path.abspath(trainer_module.__file__)), "miopen_db")
    assert Trainer._find_db_matches_device(db, "gfx950", 256, (3, 5, 0))
    assert Trainer._find_db_matches_device(db, "gfx950", 256, None)              # version unknown: arch + CU count decide
    assert not Trainer._find_db_matches_device(db, "gfx942", 304, (3, 5, 0))     # another GPU: Find stays on
    assert not Trainer._find_db_matches_device(db, "gfx950", 128, (3, 5, 0))     # a partitioned device is another db file
    assert not Trainer._find_db_matches_device(db, "gfx950", 256, (3, 6, 0))     # another MIOpen build
    assert not Trainer._find_db_matches_device("/nonexistent", "gfx950", 256, (3, 5, 0))


def test_sweep_leaves_recent_copies_of_other_checkouts_alone(tmp_path):
    import miopen_env
    root = str(tmp_path)
    for name, age in (("aaaaaaaaaaaa_rank0", 0), ("bbbbbbbbbbbb_rank0", 0), ("cccccccccccc_rank0", 5 * 24 * 3600)):
        d = os.path.join(root, name)
        os.makedirs(d)
        f = os.path.join(d, "x.ufdb.txt")
        open(f, "w").write("r")
        t = time.time() - age
        os.utime(f, (t, t))
        os.utime(d, (t, t))
    miopen_env._sweep(root, keep="aaaaaaaaaaaa")
    assert sorted(os.listdir(root)) == ["aaaaaaaaaaaa_rank0", "bbbbbbbbbbbb_rank0"]      # the live neighbour stays, the stale one goes


def test_bench_bookkeeping():
    """SURVEY 8(d)'s algorithmic bytes at the bench shape, the reference's CPU timing parsed from the committed record (VERDICT r5 weak
    #13: no literals in bench.py), and the all-reduce budget's ring arithmetic."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    total, single = bench.algorithmic_bytes(12, 192, 640, [0, 1, 2], True)
    assert abs(total / 12 - 48.75e6) < 0.01e6 and single < total                 # 48.75 MB per image with the motion terms
    assert abs(bench.algorithmic_bytes(12, 192, 640, [0, 1, 2], False)[0] / 12 - 29.40e6) < 0.01e6
    rec = bench.reference_cpu_record()
    assert rec["source"].endswith("_reference_cpu_build_container.txt") and rec["threads"] >= 1
    assert rec["loss_path_fwd_bwd_img_per_s"] > 0 and rec["full_step_img_per_s"] > 0
    assert abs(rec["loss_path_fwd_bwd_img_per_s"] * rec["loss_path_fwd_bwd_median_s"] - 12) < 0.1     # B = 12 images per evaluation

    class Buffer:
        @staticmethod
        def numel():
            return 45205504          # the headline step's flat gradient buffer (180.8 MB)

    class Step:
        flat_all = Buffer
    bud = bench.allreduce_budget(Step, 8)
    assert bud["gradient_bytes"] == 4 * 45205504 and bud["exposed_ms_at_8_gpus"][0] < bud["exposed_ms_at_8_gpus"][1]
    assert abs(bud["exposed_ms_at_8_gpus"][1] - 2 * 7 / 8 * 4 * 45205504 / 150e9 * 1e3) < 1e-3        # 2.11 ms at one link's worth
    assert abs(bud["exposed_ms_at_2_gpus"][0] - 4 * 45205504 / 300e9 * 1e3) < 1e-3
    assert bench.allreduce_budget(None, 1) is None
