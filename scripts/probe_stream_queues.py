"""Which of torch's pool streams share a hardware queue?  HIP multiplexes its streams onto GPU_MAX_HW_QUEUES (4) hardware queues; two
streams on one queue run one after the other (the pose branch of the replayed step waits behind the motion
encoder).  For every pair (a, b) of the first N streams: a long chain of kernels on a, then one tiny kernel on b -- if b finishes
long before a's chain, they are on different queues.  Prints the groups of streams that serialise with each other.
    python scripts/probe_stream_queues.py [N=8]
Written at the end of round 3 (the GPU budget was spent): not yet run on hardware."""
import sys

import torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda")
streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(n)]
names = ["current"] + ["pool%d" % i for i in range(n)]
big = torch.empty(64 << 20, dtype=torch.float32, device=dev).fill_(1.0)       # 256 MB: ~70 us per pass
small = torch.zeros(64, device=dev)


def concurrent(a, b, chain=40):
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


concurrent(streams[0], streams[1])            # warm-up
group = list(range(len(streams)))             # union-find over "serialise with each other"
for i in range(len(streams)):
    for j in range(i + 1, len(streams)):
        if not concurrent(streams[i], streams[j]) and not concurrent(streams[j], streams[i]):
            gi, gj = group[i], group[j]
            group = [gi if g == gj else g for g in group]
for g in sorted(set(group)):
    print("one queue:", [names[i] for i in range(len(streams)) if group[i] == g])
