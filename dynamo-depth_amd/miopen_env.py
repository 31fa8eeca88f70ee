"""MIOpen environment of a training process; import BEFORE torch.  `miopen_db/` holds find-db records (text, written by MIOpen's
own Find on an MI355X for the KITTI / LiteMono workload): with them the first step skips most of the solver search.  Every
rank works on a private writable copy (MIOpen appends to its user db)."""
import os
import shutil

_HERE = os.path.dirname(os.path.abspath(__file__))


def setup(find_mode="FAST"):
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this platform
    os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")           # kernel arguments in device memory: ~7 % at ~3 000 launches per step
    # hipGraph launches through the runtime's "packet capture" path (AQL packets and kernel arguments recorded once per graph and
    # copied into the queue at launch, ROCm 7.2's default) are not safe when several launches are in flight: training runs of the
    # replayed step turned non-finite within 100-400 steps EVERY time -- first the depth network's gradients, out of finite inputs,
    # never with a host sync per step, never with the kernels serialised (AMD_SERIALIZE_KERNEL=3), never with this switch off
    # (scripts/nan_hunt.sh).  Off: graph launches build their packets per launch.
    os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
    # (GPU_MAX_HW_QUEUES stays at HIP's default of 4: with 8 every branch of the step got a hardware queue of its own and the step
    # took 62 ms instead of 47.7 -- more queues than the command processor serves at once are time-sliced)
    os.environ.setdefault("MIOPEN_FIND_MODE", os.environ.get("DD_MIOPEN_FIND_MODE", find_mode))
    os.environ.setdefault("MIOPEN_LOG_LEVEL", "2")                # errors only: the fallback-solver warnings flood stderr
    src = os.path.join(_HERE, "miopen_db")
    if os.path.isdir(src) and "MIOPEN_USER_DB_PATH" not in os.environ:
        os.environ["MIOPEN_USER_DB_PATH"] = _private_copy(src)


def _private_copy(src):
    """A writable copy of the shipped find-db for THIS rank (MIOpen appends what its Find learns to its user db): keyed by the
    content of the shipped records, so a refreshed `miopen_db/` is never shadowed by a stale copy, and by the local rank, so that
    the ranks of a job never append to one file -- and PERSISTENT: what Find learns about shapes the shipped records lack is
    there again in the next run (a pid-keyed copy, round 3, threw it away with the process).  Built under a temporary name and
    renamed into place; processes of the same rank that overlap (a test and the train.py it starts) share the directory, which
    MIOpen guards with its own lock files."""
    import hashlib
    import tempfile
    digest = hashlib.sha1()
    for name in sorted(os.listdir(src)):
        digest.update(name.encode())
        with open(os.path.join(src, name), "rb") as fh:
            digest.update(fh.read())
    root = os.path.join(tempfile.gettempdir(), "dd_miopen_db_{}".format(os.getuid() if hasattr(os, "getuid") else 0))
    os.makedirs(root, exist_ok=True)
    rank = os.environ.get("LOCAL_RANK", "0")
    # ... and by the devices the process may see: two independent one-GPU jobs on one host (both local rank 0, different
    # HIP_VISIBLE_DEVICES) then append to different copies instead of relying on MIOpen's lock files alone
    vis = ",".join(os.environ.get(k, "") for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES")).strip(",")
    dev = "_dev" + hashlib.sha1(vis.encode()).hexdigest()[:6] if vis else ""
    dst = os.path.join(root, "{}_rank{}{}".format(digest.hexdigest()[:12], rank if rank.isdigit() else "0", dev))
    if not os.path.isdir(dst):
        tmp = tempfile.mkdtemp(prefix=".incoming_", dir=root)
        for name in os.listdir(src):
            shutil.copy2(os.path.join(src, name), os.path.join(tmp, name))
        try:
            os.rename(tmp, dst)
        except OSError:                    # a process of the same rank got there first: theirs is complete
            shutil.rmtree(tmp, ignore_errors=True)
    _sweep(root, keep=digest.hexdigest()[:12])
    return dst


def _sweep(root, keep, max_age_s=2 * 24 * 3600.0):
    """Removes copies of OTHER shipped databases -- but only ones nothing has touched for two days: a job from another checkout (or
    from a refreshed miopen_db/) may be running on its copy right now, and MIOpen appends to it while it runs."""
    import time
    now = time.time()
    for name in os.listdir(root):
        if name.startswith(keep + "_rank"):
            continue
        path = os.path.join(root, name)
        try:
            newest = max([os.path.getmtime(path)] + [os.path.getmtime(os.path.join(path, f)) for f in os.listdir(path)])
        except OSError:
            continue
        if now - newest > max_age_s:
            shutil.rmtree(path, ignore_errors=True)
