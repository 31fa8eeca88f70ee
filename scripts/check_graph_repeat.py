"""Are repeated training runs in one process the same run?  Eager twice, replayed (per-network hipGraphs) three times: losses of the
four steps and the distance of the final weights from the first eager run (scripts/check_ms_determinism.run)."""
import importlib.util
import os
import sys

path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "check_ms_determinism.py")
spec = importlib.util.spec_from_file_location("check_ms_determinism", path)
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)
order = sys.argv[1] if len(sys.argv) > 1 else "eeggg"
ref = None
for i, kind in enumerate(order):
    extra = ["--no_hip_graph"] if kind == "e" else []
    params, bufs, losses = mod.run(True, extra=extra)
    if ref is None:
        ref = (params, bufs)
    print("run %d %s: losses %s  |w - w_run0| %.3e  |buffers - run0| %.3e" % (
        i, "eager" if kind == "e" else "graph", [round(x, 6) for x in losses], float((params - ref[0]).abs().max()), float((bufs - ref[1]).abs().max())), flush=True)
