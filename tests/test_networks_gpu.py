"""The Abs-Rel criterion of the north star ON THE MI355X PATH (VERDICT r4 missing #3): eval-mode networks of this tree on the GPU
(MIOpen convolutions, every network-side HIP hook, NCHW and channels-last) on tiny_kitti's batch with the key-addressed weights
of the golden, `dd_depth_metrics` on the device, against the seven metrics the unmodified reference produced on the CPU
(tests/golden/make_golden_net.py; reference tools.py:16-73, eval/depth.py:60-79).  tests/test_networks.py holds the same check
with the networks on the CPU."""
import os

import numpy as np
import pytest
import torch

from fill import fill_state
from test_networks import batch_from_golden, compare_summary, make_opt

pytestmark = pytest.mark.gpu

NAMES = ["de:abs_rel", "de:sq_rel", "de:rms", "de:log_rms", "da:a1", "da:a2", "da:a3"]


@pytest.fixture(scope="module")
def z(golden_dir):
    return np.load(os.path.join(golden_dir, "net_tiny_kitti.npz"))


@pytest.mark.parametrize("channels_last", [False, True])
@pytest.mark.parametrize("depth_model", ["monodepthv2", "litemono"])
def test_abs_rel_on_the_gpu_path_matches_reference(z, depth_model, channels_last):
    import networks
    from hipops import lib as L
    from tools import DepthMetrics
    L.load()                                   # raises when libdynamo_hip.so is missing: no silent stock path
    opt = make_opt(depth_model, ["--channels_last"] if channels_last else [])
    model = networks.Model(opt)
    for name in sorted(model.module_names):
        fill_state(getattr(model, name), seed=3)
    model.to("cuda")
    if channels_last:
        model.to(memory_format=torch.channels_last)
    model.set_eval()
    inputs = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in batch_from_golden(z, opt.scales).items()}
    with torch.no_grad():
        outputs = model(inputs)
    report = []
    # MIOpen's fp32 solvers against the reference's CPU convolutions: 5e-4 relative on the network outputs
    fails = compare_summary(z, depth_model + "/eval/", outputs, 5e-4, 5e-6, report)
    lo, hi = 1 / opt.max_depth, 1 / opt.min_depth
    outputs[("disp_scaled", 0, 0)] = lo + (hi - lo) * outputs[("disp", 0, 0)]
    dm = DepthMetrics(opt.eval_img_bound, opt.eval_min_depth, opt.eval_max_depth)
    assert outputs[("disp_scaled", 0, 0)].is_cuda
    metrics = dm(inputs, outputs)              # device tensors -> dd_depth_metrics (tools.DepthMetrics.forward)
    got = np.array([float(metrics[m]) for m in NAMES])
    want = z[depth_model + "/eval/metrics"]
    # ... and the same numbers through the plain-torch restatement of the reference's loop on the same GPU outputs
    host = dm._forward_torch({k: (v.cpu() if torch.is_tensor(v) else v) for k, v in inputs.items()},
                             {("disp_scaled", 0, 0): outputs[("disp_scaled", 0, 0)].cpu()}, None)
    host = np.array([float(host[m]) for m in NAMES])
    report.append("metrics device %s\n        torch  %s\n        want   %s" % (got, host, want))
    print("\n".join(report))
    assert not fails, fails
    assert abs(got[0] - want[0]) < 0.002, (got[0], want[0])            # north star: Abs Rel within +-0.002 of the reference
    np.testing.assert_allclose(got, want, rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(got, host, rtol=2e-5, atol=2e-6)
