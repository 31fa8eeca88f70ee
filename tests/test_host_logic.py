"""Host-side decisions that need no GPU: which find-db records apply to which device, the sweep of stale MIOpen
user-db copies."""
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
