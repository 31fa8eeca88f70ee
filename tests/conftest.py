"""pytest configuration: registers the `gpu` marker and puts the drop-in tree on sys.path.

`-m "not gpu"` runs here (no GPU); `-m gpu` runs on an MI355X box and goes through the C-ABI HIP library.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PRODUCT = os.path.join(ROOT, "dynamo-depth_amd")
for p in (os.path.join(ROOT, "tests", "golden"), ROOT, PRODUCT):
    if p not in sys.path:
        sys.path.insert(0, p)


# MIOpen Find (torch.backends.cudnn.benchmark, the Trainer's default on a GPU) times every applicable solver the first time
# a convolution shape is seen: right for a training run, minutes per test on the dozens of one-off shapes here (the suite took
# 40 min with it against 12 without).  Inherited by the subprocesses the distributed / train.py tests start.
os.environ.setdefault("DD_MIOPEN_FIND", "0")
# hipGraph launches without the runtime's packet-capture path (dynamo-depth_amd/miopen_env.py): must be set before the first device call
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")


@pytest.fixture(autouse=True)
def _no_leftover_find_mode():
    import torch
    torch.backends.cudnn.benchmark = False
    yield


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "gpu_slow: second variants of the slowest GPU sweeps (the bf16 twins of the fp16 config-5 tests); run with "
                                       "DD_GPU_SLOW=1 -- the default -m gpu set keeps every SURVEY 8 row's parity test and fits the driver's time limit")


def pytest_collection_modifyitems(config, items):
    import torch
    if os.environ.get("DD_GPU_SLOW", "0") != "1":
        slow = pytest.mark.skip(reason="gpu_slow: run with DD_GPU_SLOW=1")
        for item in items:
            if "gpu_slow" in item.keywords:
                item.add_marker(slow)
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
