"""Tuned solution choices for the library GEMMs of a training process (PyTorch's TunableOp: `torch.cuda.tunable`).

LiteMono's point-wise Linears and attention products (reference networks/depth_encoder.py:176-276) are plain library GEMMs
(hipBLASLt / rocBLAS through ATen) -- 5.6 ms of the headline step's 42.7 ms of kernel time -- and the libraries' heuristics pick
slow kernels for several of their shapes (x (11520,224) . W (224,1344): 103 us by default, 64 us for the best solution).
`gemm_db/tunableop_gfx950.csv` holds the winners TunableOp measured on an MI355X for the bench workloads (recorded by
scripts/refresh_gemm_db.sh, like miopen_db/ for the convolutions); `enable()` switches TunableOp on with tuning OFF, so a GEMM
whose shape is in the file runs the recorded solution and every other GEMM the library's default -- nothing is timed at run time.
TunableOp refuses a file whose validators (PyTorch, HIP, hipBLASLt, rocBLAS versions, GPU architecture) differ from the process's:
then everything stays on the defaults.  DD_GEMM_TUNED=0 leaves TunableOp alone; so does any PYTORCH_TUNABLEOP_* variable in the
environment (the user, or the refresh script, is in charge).  The fp32 results of two solutions differ by rounding only."""
import hashlib
import os
import shutil
import tempfile

_HERE = os.path.dirname(os.path.abspath(__file__))
SHIPPED = os.path.join(_HERE, "gemm_db", "tunableop_gfx950.csv")
STATE = {"status": "not asked", "entries": 0}


def enable():
    """Call once, after `import torch`, in a process that has a GPU.  Returns (and keeps in STATE) what happened."""
    import torch
    if os.environ.get("DD_GEMM_TUNED", "1") != "1":
        return _done("off (DD_GEMM_TUNED=0)")
    if any(k.startswith("PYTORCH_TUNABLEOP_") for k in os.environ):
        return _done("left to the PYTORCH_TUNABLEOP_* environment")
    if STATE["status"].startswith("on"):
        return STATE["status"]
    if not (torch.cuda.is_available() and os.path.isfile(SHIPPED)):
        return _done("off (no GPU or no shipped records)")
    try:
        import torch.cuda.tunable as tun
        # TunableOp writes its table back to its file when the process ends: give it a private copy, keyed by the records' content
        with open(SHIPPED, "rb") as fh:
            digest = hashlib.sha1(fh.read()).hexdigest()[:12]
        root = os.path.join(tempfile.gettempdir(), "dd_gemm_db_{}".format(os.getuid() if hasattr(os, "getuid") else 0))
        os.makedirs(root, exist_ok=True)
        dst = os.path.join(root, "{}_{}.csv".format(digest, os.getpid()))
        shutil.copy2(SHIPPED, dst)
        tun.enable(True)
        tun.tuning_enable(False)
        tun.set_filename(dst, insert_device_ordinal=False)
        if not tun.read_file(dst):
            tun.enable(False)
            return _done("off (the shipped records were recorded under other library versions: defaults)")
        STATE["entries"] = len(tun.get_results())
        _sweep(root)
        return _done("on ({} recorded solutions)".format(STATE["entries"]))
    except Exception as exc:          # an older torch without torch.cuda.tunable, a read-only temp dir, ...: the defaults are always correct
        return _done("off ({}: {})".format(type(exc).__name__, exc))


def _done(status):
    STATE["status"] = status
    return status


def _sweep(root, max_age_s=2 * 24 * 3600.0):
    import time
    now = time.time()
    for name in os.listdir(root):
        path = os.path.join(root, name)
        try:
            if now - os.path.getmtime(path) > max_age_s:
                os.remove(path)
        except OSError:
            pass
