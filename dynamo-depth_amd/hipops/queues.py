"""HIP streams on DISTINCT hardware queues.

HIP multiplexes its streams onto GPU_MAX_HW_QUEUES (4) hardware queues, and two streams that share a queue run one after the
other whatever their events say.  Round 3's timeline of the replayed step showed the pose branch waiting
behind the motion encoder in both directions although nothing orders them: their streams shared a queue -- ~4 ms of a 47 ms
step.  More queues are not the answer (GPU_MAX_HW_QUEUES=8: every branch slows down, 62 ms); choosing WHICH streams share is.

pick(n) hands out n streams from torch's pool that were MEASURED to run concurrently with each other and with the current
stream: a long chain of kernels on one, a tiny kernel on the other -- if the tiny one finishes long before the chain, the two
are on different queues (scripts/probe_stream_queues.py is the stand-alone form).  ~25 ms once per process.
"""
import os

import torch

_cache = {}


def _concurrent(a, b, big, small, chain=24):
    torch.cuda.synchronize()
    a0, a1, b1 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    with torch.cuda.stream(a):
        a0.record(a)
        for _ in range(chain):
            big.mul_(1.0)
        a1.record(a)
    with torch.cuda.stream(b):
        small.add_(1.0)
        b1.record(b)
    torch.cuda.synchronize()
    return a0.elapsed_time(b1) < 0.5 * a0.elapsed_time(a1)


def pick(n, candidates=12, report=None):
    """n streams, pairwise on different hardware queues and none on the current stream's queue, as far as the device has
    queues for them; the remainder (if any) are ordinary pool streams.  DD_STREAM_PICK=0 returns pool streams unprobed."""
    dev = torch.cuda.current_device()
    key = (dev, n)
    if key in _cache:
        return list(_cache[key])
    pool = [torch.cuda.Stream() for _ in range(max(candidates, n))]
    if os.environ.get("DD_STREAM_PICK", "1") == "0" or torch.cuda.is_current_stream_capturing():
        _cache[key] = pool[:n]
        return list(_cache[key])
    cur = torch.cuda.current_stream()
    big = torch.empty(16 << 20, dtype=torch.float32, device="cuda").fill_(1.0)        # 64 MB: ~25 us per pass
    small = torch.zeros(64, device="cuda")
    _concurrent(cur, pool[0], big, small)                 # warm-up (first launches of the two kernels)
    chosen = []
    for st in pool:
        if len(chosen) == n:
            break
        others = [cur] + chosen
        # two streams share a queue when NEITHER runs beside the other's chain (scripts/probe_stream_queues.py's criterion; asking
        # for both directions to pass found one stream of four: a single direction fails now and then for reasons of its own)
        if all(_concurrent(o, st, big, small) or _concurrent(st, o, big, small) for o in others):
            chosen.append(st)
    found = len(chosen)
    if os.environ.get("DD_STREAM_PICK_DEBUG") == "1":
        import sys
        print("[queues] current stream {}; picked {} of {} on distinct queues: pool positions {}".format(
            cur, found, n, [pool.index(st) for st in chosen]), file=sys.stderr, flush=True)
    for st in pool:
        if len(chosen) == n:
            break
        if st not in chosen:
            chosen.append(st)
    if report is not None:
        report["distinct_queues"] = found
    _cache[key] = chosen
    _cache[(dev, "found")] = found
    return list(chosen)


def found(dev=None):
    """How many of the streams handed out by the last pick() on this device sit on a hardware queue of their own."""
    return _cache.get((torch.cuda.current_device() if dev is None else dev, "found"))
