"""csrc/dd_math.h (the arithmetic every HIP kernel calls) executed on the CPU through
tests/hostmath/photo_host.cpp and compared with the oracle's autograd.  No GPU needed."""
import ctypes as C

import pytest
import torch

import photo_case as pc


@pytest.fixture(scope="module")
def host_lib():
    return pc.build_host_lib()


@pytest.mark.parametrize("phase", ["disp_init", "motion_init", "mask_init"])
def test_dd_math_matches_oracle(host_lib, phase):
    ts = {0: [1, 1], -1: [1, 2], 1: [1, 2]}
    case = pc.Case(phase, 2, 64, 96, [0, 1, 2, 3], seed=7, ts=ts).run_oracle()
    args, t = case.photo_buffers("cpu")
    rc = host_lib.dd_photo_loss_host(C.byref(args))
    assert rc == 0
    report = []
    fails = case.check(t, report=report) + case.check_grads(t, report=report)
    print("\n".join(report))
    assert not fails, fails


def test_dd_math_forward_only(host_lib):
    case = pc.Case("disp_init", 1, 32, 64, [0, 2], seed=3).run_oracle()
    args, t = case.photo_buffers("cpu", want_grad=False)
    assert host_lib.dd_photo_loss_host(C.byref(args)) == 0
    assert not case.check(t)
