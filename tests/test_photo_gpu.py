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
    pytest.param("mask_init", 2, 192, 640, [0, 1, 2], True, marks=pytest.mark.gpu_slow),      # (gpu_slow: the same shape as the next row / the
    ("fine_tune", 2, 192, 640, [0, 1, 2], True),                                               #  same phase as the 288x512 row, a third case
    ("fine_tune", 1, 320, 480, [0, 1, 2], True),                                               #  of 288x512 -- each 5-15 s of float64 oracle on
    ("fine_tune", 1, 320, 480, [0, 1, 2], False),                                              #  the host; every shape, phase and the separate-
    ("disp_init", 1, 288, 512, [0, 1, 2, 3], False),                                           #  tensor instantiation stay in the default set)
    ("motion_init", 1, 288, 512, [0, 1, 2, 3], True),
    ("fine_tune", 1, 288, 512, [0, 1, 2, 3], True),       # BASELINE.json config 5's loss shape with every motion term
    pytest.param("mask_init", 1, 288, 512, [0, 1, 2, 3], True, marks=pytest.mark.gpu_slow),
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


def _flat(t):
    """Every tensor the kernel wrote, by name (the scale dictionaries flattened)."""
    out = {"g_T0": t["g_T"][0], "g_T1": t["g_T"][1], "sums": t["sums"]}
    for si, d in enumerate(t["scales"]):
        for k, v in d.items():
            if k.startswith(("g_", "out_")):
                for j, x in enumerate(v if isinstance(v, (list, tuple)) else [v]):
                    if torch.is_tensor(x):
                        out["%s[%d][%d]" % (k, si, j)] = x
    return out


def test_pack_rgb_is_a_permutation(lib):
    """dd_pack_rgb: (B,3,H,W) -> (B,H,W,3), the same values (the layout DDPhotoArgs.source_packed takes)."""
    from hipops.inputs import pack_rgb
    torch.manual_seed(3)
    for shape in [(1, 3, 2, 2), (2, 3, 6, 10), (3, 3, 64, 96), (12, 3, 192, 640)]:
        x = torch.rand(*shape, device="cuda")
        assert torch.equal(pack_rgb(x), x.permute(0, 2, 3, 1).contiguous()), shape
    with pytest.raises(lib.DynamoHipError):
        pack_rgb(torch.rand(1, 3, 3, 5, device="cuda"))          # H*W not a multiple of 4


@pytest.mark.parametrize("materialise", [True, False])
@pytest.mark.parametrize("phase,shared", [("disp_init", False), ("motion_init", True), ("mask_init", False), ("fine_tune", True)])
def test_packed_sources_give_the_planar_result_bit_for_bit(lib, phase, shared, materialise):
    """The photometric kernel gathering its source taps from the pixel-interleaved copies (DDPhotoArgs.source_packed: one 12-byte load per
    tap) against the same launch on the planar tensors of the reference boundary: the same values enter the same arithmetic, so every
    output -- sums, gradients, materialised maps -- is identical bit for bit, borders and reflect-padded rows included."""
    ts = {0: [1, 1], -1: [1, 2], 1: [1, 2]}
    case = pc.Case(phase, 2, 64, 96, [0, 1, 2, 3], seed=11, ts=ts)
    case.outputs = pc.synth.leaves_to_outputs(case.leaves, case.scales, pc.orc.pose_matrix, case.cmpflow, case.motmask)
    res = []
    for packed in (False, True):
        args, t = case.photo_buffers("cuda", materialise=materialise, want_grad=True, shared=shared, packed=packed)
        assert bool(args.source_packed[0]) == packed and bool(args.source_packed[1]) == packed
        lib.check(lib.load().dd_photo_loss(C.byref(args), lib.current_stream()), "dd_photo_loss")
        torch.cuda.synchronize()
        res.append(_flat(t))
    assert res[0].keys() == res[1].keys() and len(res[0]) >= 7
    for k in res[0]:
        assert torch.equal(res[0][k], res[1][k]), k
    assert float(res[0]["sums"].abs().sum()) > 0
