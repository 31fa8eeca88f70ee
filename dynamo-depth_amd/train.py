"""Entry point, launch-line compatible with the reference's train.py (train.py:1-33):

    python train.py -d kitti ...                                              # one GPU
    python -m torch.distributed.run --nproc-per-node 8 train.py --cuda_ids 0 1 2 3 4 5 6 7 ...
    python -m torch.distributed.launch --nproc_per_node=4 train.py --cuda_ids 0 1 2 3 ...   # the reference's line

One process per GPU; the `nccl` backend of PyTorch-ROCm is RCCL over xGMI.  LOCAL_RANK from torchrun is honoured
(the reference only reads --local_rank), and --cuda_ids defaults to 0..LOCAL_WORLD_SIZE-1 when omitted.
"""
import os

import miopen_env  # noqa: E402

miopen_env.setup()      # kernel arguments in device memory, MIOpen find mode, the shipped find-db records

import torch  # noqa: E402
from torch.distributed import destroy_process_group, init_process_group  # noqa: E402

from options import DynamoOptions  # noqa: E402
from Trainer import Trainer  # noqa: E402


def ddp_setup(backend="nccl"):
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this platform
    if backend == "nccl" and not torch.cuda.is_available():
        backend = "gloo"
    init_process_group(backend=backend)


def ddp_cleanup():
    destroy_process_group()


def main(argv=None):
    opt = DynamoOptions().parse(args=argv)
    opt.local_world_size = int(os.environ.get("LOCAL_WORLD_SIZE", 1))
    opt.ddp = opt.local_world_size > 1
    if "LOCAL_RANK" in os.environ:
        opt.local_rank = int(os.environ["LOCAL_RANK"])
    if opt.ddp and len(opt.cuda_ids) == 1:
        opt.cuda_ids = list(range(opt.local_world_size))
    assert len(opt.cuda_ids) == opt.local_world_size, \
        "opt.cuda_ids(={}) does not match opt.local_world_size(={})".format(opt.cuda_ids, opt.local_world_size)
    if opt.ddp:
        ddp_setup(opt.dist_backend)
    try:
        Trainer(opt).train()
    finally:
        if opt.ddp:
            ddp_cleanup()


if __name__ == "__main__":
    main()
