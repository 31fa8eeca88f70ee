"""MIOpen environment of a training process; import BEFORE torch.  `miopen_db/` holds find-db records (text, written by MIOpen's
own Find on an MI355X for the KITTI / LiteMono workload): with them the first step skips most of the solver search.  Every
rank works on a private writable copy (MIOpen appends to its user db)."""
import os
import shutil

_HERE = os.path.dirname(os.path.abspath(__file__))


def setup(find_mode="FAST"):
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this platform
    os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")           # kernel arguments in device memory: ~7 % at ~3 000 launches per step
    os.environ.setdefault("MIOPEN_FIND_MODE", os.environ.get("DD_MIOPEN_FIND_MODE", find_mode))
    os.environ.setdefault("MIOPEN_LOG_LEVEL", "2")                # errors only: the fallback-solver warnings flood stderr
    src = os.path.join(_HERE, "miopen_db")
    if os.path.isdir(src) and "MIOPEN_USER_DB_PATH" not in os.environ:
        dst = "/tmp/dd_miopen_db_{}".format(os.environ.get("LOCAL_RANK", "0"))
        if not os.path.isdir(dst):
            shutil.copytree(src, dst)
        os.environ["MIOPEN_USER_DB_PATH"] = dst
