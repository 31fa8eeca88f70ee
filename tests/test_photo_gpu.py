"""Parity of the fused HIP photometric kernel (through the C ABI) against the oracle.  GPU only."""
import ctypes as C

import pytest
import torch

import photo_case as pc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from hipops import lib as L
    return L


def run_case(lib, case, materialise=True, want_grad=True, shared=False):
    args, t = case.photo_buffers("cuda", materialise=materialise, want_grad=want_grad, shared=shared)
    rc = lib.load().dd_photo_loss(C.byref(args), lib.current_stream())
    lib.check(rc, "dd_photo_loss")
    torch.cuda.synchronize()
    return t


@pytest.mark.parametrize("phase", ["disp_init", "motion_init", "mask_init"])
def test_photo_kernel_matches_oracle(lib, phase):
    ts = {0: [1, 1], -1: [1, 2], 1: [1, 2]}
    case = pc.Case(phase, 2, 64, 96, [0, 1, 2, 3], seed=7, ts=ts).run_oracle()
    t = run_case(lib, case)
    report = []
    fails = case.check(t, report=report) + case.check_grads(t, report=report)
    print("\n".join(report))
    assert not fails, fails


@pytest.mark.parametrize("materialise", [True, False])
@pytest.mark.parametrize("phase", ["motion_init", "mask_init"])
def test_photo_kernel_shared_tensors(lib, phase, materialise):
    """What networks.Model publishes: one flow field and one mask tensor for both frames (the 5-plane path), with and
    without the materialised outputs of a log step (two different instantiations of the kernel)."""
    ts = {0: [1, 1], -1: [1, 2], 1: [1, 2]}
    case = pc.Case(phase, 2, 64, 96, [0, 1, 2, 3], seed=7, ts=ts).run_oracle()
    t = run_case(lib, case, materialise=materialise, shared=True)
    report = []
    fails = case.check(t, report=report) + case.check_grads(t, report=report)
    print("\n".join(report))
    assert not fails, fails


# Full benchmark shapes of BASELINE.json against the oracle (B small enough for the CPU oracle to finish in seconds):
# tile counts and XCD remapping that only the big shapes have (320x480: 15 tile columns -> ntiles % 8 != 0).
@pytest.mark.parametrize("phase,B,H,W,scales,shared", [
    ("mask_init", 2, 192, 640, [0, 1, 2], True),
    ("fine_tune", 2, 192, 640, [0, 1, 2], True),
    ("fine_tune", 1, 320, 480, [0, 1, 2], True),
    ("fine_tune", 1, 320, 480, [0, 1, 2], False),
    ("disp_init", 1, 288, 512, [0, 1, 2, 3], False),
    ("motion_init", 1, 288, 512, [0, 1, 2, 3], True),
    ("fine_tune", 1, 288, 512, [0, 1, 2, 3], True),       # BASELINE.json config 5's loss shape with every motion term
    ("mask_init", 1, 288, 512, [0, 1, 2, 3], True),
])
def test_photo_kernel_full_size(lib, phase, B, H, W, scales, shared):
    case = pc.Case(phase, B, H, W, scales, seed=13).run_oracle(fp64=True)
    t = run_case(lib, case, materialise=False, shared=shared)
    report = []
    # per-pixel gradients: decision-masked against the fp64 oracle (<= 1e-4 on the elements no decision moved); the pose
    # gradients, sums over all pixels, in units of the fp32 oracle's own distance from fp64
    # (the masked check first: a decision the KERNEL alone flips moves the pose gradients -- sums over every pixel -- as well)
    fails = case.check(t, report=report) + case.check_grads_masked(t, report=report)
    fails += case.check_grads(t, report=report, only_T=True, t_slack=16.0 if case.kernel_flips else 4.0)
    print("\n".join(report))
    assert not fails, fails


@pytest.mark.parametrize("phase,B,H,W,scales", [
    ("disp_init", 1, 192, 640, [0, 1, 2]),       # full KITTI tile grid, several tiles per row/column
    ("mask_init", 3, 96, 160, [0, 1, 2, 3]),     # W not a multiple of the 64-wide tile
    ("motion_init", 1, 32, 32, [0, 3]),          # image smaller than a tile
])
def test_photo_kernel_shapes(lib, phase, B, H, W, scales):
    case = pc.Case(phase, B, H, W, scales, seed=11).run_oracle()
    t = run_case(lib, case)
    report = []
    fails = case.check(t, report=report) + case.check_grads(t, report=report)
    print("\n".join(report))
    assert not fails, fails


def test_photo_forward_only_and_no_materialise(lib):
    case = pc.Case("disp_init", 2, 64, 96, [0, 2], seed=5).run_oracle()
    t = run_case(lib, case, materialise=False, want_grad=False)
    assert not case.check(t)


def test_photo_deterministic_sums(lib):
    case = pc.Case("mask_init", 2, 64, 128, [0, 1], seed=2).run_oracle()
    a = run_case(lib, case)["sums"].clone()
    b = run_case(lib, case)["sums"].clone()
    assert torch.equal(a[:, 0], b[:, 0])   # photometric sums: fixed reduction order
