"""The shipped GEMM records (dynamo-depth_amd/gemm_db/, PyTorch TunableOp with tuning off -- gemm_env.py): the recorded solutions are
in use, and they compute what the libraries' default kernels compute.  GPU only."""
import os

import pytest
import torch

from test_networks import make_opt

pytestmark = pytest.mark.gpu


def test_records_are_in_use_and_match_this_build():
    """gemm_env.enable() accepts the shipped file on the box the suite runs on (TunableOp's validators: PyTorch / HIP / hipBLASLt / rocBLAS
    versions and the GPU architecture) -- a refused file would silently put every GEMM back on the slower defaults."""
    import gemm_env
    import torch.cuda.tunable as tun
    status = gemm_env.enable()
    print("gemm_env:", status)
    assert status.startswith("on ("), status
    assert tun.is_enabled() and not tun.tuning_is_enabled()
    assert gemm_env.STATE["entries"] >= 100
    # the file TunableOp writes back at exit is a private copy, never the shipped one
    assert os.path.abspath(tun.get_filename()) != os.path.abspath(gemm_env.SHIPPED)


def test_recorded_linear_solutions_against_float64():
    """Every recorded fp32 Linear (GemmAndBiasTunableOp_float_TN: x (m,k) . W (n,k)^T + b) through F.linear with the records on,
    against float64, beside the library's default kernel on the same operands: the recorded solution is no further from float64
    than three times the default's error (both are fp32 MFMA kernels: ~1e-6 of the result's scale)."""
    import gemm_env
    import torch.cuda.tunable as tun
    assert gemm_env.enable().startswith("on (")
    rows = [ln.split(",") for ln in open(gemm_env.SHIPPED) if ln.startswith("GemmAndBiasTunableOp_float_TN,")]
    assert len(rows) >= 20
    torch.manual_seed(5)
    worst = 0.0
    for _, sig, sol, _t in rows:
        n, m, k = (int(v) for v in sig.split("_")[1:4])
        x, w, b = torch.randn(m, k, device="cuda"), torch.randn(n, k, device="cuda") / k ** 0.5, torch.randn(n, device="cuda")
        ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
        scale = float(ref.abs().max())
        tun.enable(True)
        got = torch.nn.functional.linear(x, w, b)
        tun.enable(False)
        base = torch.nn.functional.linear(x, w, b)
        tun.enable(True)
        e_got, e_base = float((got.double() - ref).abs().max()) / scale, float((base.double() - ref).abs().max()) / scale
        worst = max(worst, e_got)
        assert e_got <= max(3.0 * e_base, 2e-6), (sig, sol, e_got, e_base)
    print("recorded fp32 Linear solutions: %d shapes, worst error against float64 %.2e of the result's scale" % (len(rows), worst))


@pytest.mark.gpu_slow          # a configuration A/B at the bench shape (22 s), not a parity test: DD_GPU_SLOW=1
def test_headline_step_with_and_without_the_records():
    """One LiteMono fine_tune step at the bench shape (KITTI 192x640, batch 12: the shapes the records were taken on -- Linears AND the
    batched attention products) with the records on and off: same weights, same batch -> the same losses and gradient norms to
    fp32-kernel noise."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    import gemm_env
    import torch.cuda.tunable as tun
    from Trainer import Trainer
    assert gemm_env.enable().startswith("on (")
    res = {}
    for on in (True, False):
        torch.manual_seed(0)
        opt = make_opt("litemono", ["--synthetic", "--channels_last"])
        opt.batch_size = 12
        tr = Trainer(opt)
        tr.num_steps_per_epoch = 10
        tr.setup_phase("fine_tune")
        tr.bool_automask = False
        tr.step = 10
        tr.set_train()
        batch = bench.make_batch(tr, 0)
        tun.enable(on)
        try:
            outputs, losses = tr.process_batch(dict(batch))
            losses["loss"].backward()
            torch.cuda.synchronize()
        finally:
            tun.enable(True)
        norms = {name: sum(float((p.grad.double() ** 2).sum()) for p in getattr(tr.base_model, name).parameters() if p.grad is not None) ** 0.5
                 for name in sorted(tr.base_model.module_names)}
        res[on] = ({k: float(v.detach()) for k, v in losses.items() if torch.is_tensor(v) and v.numel() == 1}, norms)
        del tr
    for k, v in res[False][0].items():
        assert abs(res[True][0][k] - v) <= 2e-4 * max(abs(v), 1e-4), (k, res[True][0][k], v)
    for k, v in res[False][1].items():
        assert abs(res[True][1][k] - v) <= 2e-2 * max(v, 1e-8), (k, res[True][1][k], v)
    print("records on vs off: loss %.7f vs %.7f; gradient norms %s" % (res[True][0]["loss"], res[False][0]["loss"],
          ", ".join("%s %.3e/%.3e" % (k, res[True][1][k], res[False][1][k]) for k in res[False][1])))
