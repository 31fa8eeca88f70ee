"""The optimizer update (reference Trainer.py:150,492-497: torch.optim.Adam with its defaults).
CPU: the oracle's restatement beside torch.optim.Adam itself.  GPU: dd_adam_multi (one launch for every parameter tensor, through
hipops.adam.MultiTensorAdam on a torch optimizer object) beside the oracle and beside torch's own CUDA update -- odd sizes (vector
tails), a channels-last convolution weight (dense, permuted strides), unaligned gradient views (scalar path), parameters without a
gradient (skipped, their counters untouched), GradScaler's device scalars (unscaling on the fly, the skipped step)."""
import numpy as np
import pytest
import torch

from oracle.ref_adam import adam_step

SHAPES = [(1,), (3,), (4,), (7, 5), (4096,), (4097,), (64, 67, 3, 3), (24, 16, 1, 1), (33, 4099), (200_003,)]


def _tensors(seed, shapes, device="cpu"):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(s, generator=g).to(device) for s in shapes]


def test_oracle_is_torch_adam_on_the_cpu():
    params = [torch.nn.Parameter(t) for t in _tensors(0, SHAPES)]
    opt = torch.optim.Adam(params, 1e-4, foreach=False)
    mine = [(p.detach().numpy().copy(), np.zeros(p.shape, np.float32), np.zeros(p.shape, np.float32)) for p in params]
    wide = [tuple(a.copy() for a in m) for m in mine]
    for t in range(1, 11):
        grads = _tensors(100 + t, SHAPES)
        for p, g in zip(params, grads):
            p.grad = g * (10.0 ** (t % 4 - 2))               # gradients over four decades
        opt.step()
        mine = [adam_step(p_, p.grad.numpy(), m_, v_, t, 1e-4) for (p_, m_, v_), p in zip(mine, params)]
        wide = [adam_step(p_, p.grad.numpy(), m_, v_, t, 1e-4, dtype=np.float64) for (p_, m_, v_), p in zip(wide, params)]
    for (p_, m_, v_), (pw, mw, vw), p in zip(mine, wide, params):
        st = opt.state[p]
        # the evaluations differ in the roundings of the update (<= 1e-4 per step), which moves the rounding of the PARAMETER by
        # one place now and then: two units in the last place of the parameter after ten steps
        np.testing.assert_allclose(p_, p.detach().numpy(), rtol=2.4e-7, atol=1e-9)
        np.testing.assert_allclose(m_, st["exp_avg"].numpy(), rtol=2e-6, atol=4e-7 * float(np.abs(m_).max()))          # (sums that cancel)
        np.testing.assert_allclose(v_, st["exp_avg_sq"].numpy(), rtol=2e-6, atol=1e-12)
        np.testing.assert_allclose(p_, pw, rtol=2.4e-7, atol=1e-9)


def _gpu_setup(seed, shapes, unaligned=False, channels_last=True):
    dev = torch.device("cuda")
    params = []
    for i, t in enumerate(_tensors(seed, shapes, dev)):
        if channels_last and t.dim() == 4:
            t = t.contiguous(memory_format=torch.channels_last)
            if tuple(t.shape[2:]) == (1, 1):                # a 1x1 weight as `.to(memory_format=channels_last)` leaves it on the GPU: the
                t = t.as_strided(t.shape, (t.shape[1], 1, t.shape[1], t.shape[1]))          # size-1 dimensions carry the channel stride
        params.append(torch.nn.Parameter(t))
    ref = [torch.nn.Parameter(p.detach().clone(memory_format=torch.preserve_format)) for p in params]
    total = sum((p.numel() + 3) & ~3 for p in params) + 8
    flat = torch.zeros(total, device=dev)
    off = 1 if unaligned else 0
    for p in params:                                        # gradients as views of one flat buffer, the parameter's own strides
        p.grad = flat[off:off + p.numel()].as_strided(p.size(), p.stride())
        off += ((p.numel() + 3) & ~3) if not unaligned else p.numel()
    return params, ref


@pytest.mark.gpu
@pytest.mark.parametrize("unaligned", [False, True])
def test_one_launch_adam_is_the_oracles_and_torchs(unaligned):
    from hipops.adam import MultiTensorAdam, supported
    shapes = SHAPES + [(5,)]
    params, ref = _gpu_setup(1, shapes, unaligned=unaligned)
    frozen = len(params) - 1
    params[frozen].grad = None                              # a parameter the phase does not train
    opt = torch.optim.Adam(params, 1e-4, capturable=True, fused=True)
    opt_ref = torch.optim.Adam(ref, 1e-4, capturable=True, fused=True)
    for p, r in zip(params[:frozen], ref):                                      # the state has to exist (segments' warm-up makes it): one zero update
        p.grad.zero_()
        r.grad = torch.zeros_like(r)
    opt.step()
    opt_ref.step()
    for o in (opt, opt_ref):
        for st in o.state.values():
            st["step"].zero_()
    for p, r in zip(params, ref):
        with torch.no_grad():
            p.copy_(r)
    assert supported(opt)
    mt = MultiTensorAdam(opt)
    host = [(p.detach().cpu().numpy().copy(), np.zeros(p.shape, np.float32), np.zeros(p.shape, np.float32)) for p in params[:frozen]]
    before = params[frozen].detach().clone()
    for t in range(1, 7):
        grads = _tensors(200 + t, shapes, "cuda")
        for p, r, g in zip(params[:frozen], ref, grads):
            g = g * (10.0 ** (t % 4 - 2))
            p.grad.copy_(g)
            r.grad = g.clone(memory_format=torch.preserve_format) if g.dim() != 4 else g.contiguous(memory_format=torch.channels_last)
        mt.step()
        opt_ref.step()
        host = [adam_step(p_, p.grad.cpu().numpy(), m_, v_, t, 1e-4) for (p_, m_, v_), p in zip(host, params)]
    torch.cuda.synchronize()
    assert torch.equal(params[frozen], before) and params[frozen] not in opt.state or "step" not in opt.state[params[frozen]] or float(opt.state[params[frozen]]["step"]) == 0.0
    for (p_, m_, v_), p, r in zip(host, params, ref):
        st, st_r = opt.state[p], opt_ref.state[r]
        assert float(st["step"]) == 6.0
        np.testing.assert_allclose(p.detach().cpu().numpy(), p_, rtol=2.4e-7, atol=1e-9)
        torch.testing.assert_close(p.detach(), r.detach(), rtol=2.4e-7, atol=1e-9)
        np.testing.assert_allclose(st["exp_avg"].cpu().numpy(), m_, rtol=2e-6, atol=4e-7 * float(np.abs(m_).max()))
        np.testing.assert_allclose(st["exp_avg_sq"].cpu().numpy(), v_, rtol=2e-6, atol=1e-12)
        torch.testing.assert_close(st["exp_avg"], st_r["exp_avg"], rtol=2e-6, atol=4e-7 * float(np.abs(m_).max()))
        torch.testing.assert_close(st["exp_avg_sq"], st_r["exp_avg_sq"], rtol=2e-6, atol=1e-12)


@pytest.mark.gpu
def test_one_launch_adam_with_the_loss_scalers_device_scalars():
    from hipops.adam import MultiTensorAdam
    shapes = [(4097,), (64, 67, 3, 3)]
    params, ref = _gpu_setup(3, shapes)
    opt = torch.optim.Adam(params, 1e-4, capturable=True, fused=True)
    for p in params:
        p.grad.zero_()
    opt.step()
    for st in opt.state.values():
        st["step"].zero_()
    for p, r in zip(params, ref):
        with torch.no_grad():
            p.copy_(r)
    mt = MultiTensorAdam(opt)
    grads = _tensors(9, shapes, "cuda")
    scale = torch.tensor(1024.0, device="cuda")
    for p, g in zip(params, grads):
        p.grad.copy_(g * 1024.0)
    # an overflowed step: nothing moves, the counters stay
    mt.step(grad_scale=scale, found_inf=torch.ones((), device="cuda"))
    for p, r in zip(params, ref):
        assert torch.equal(p.detach(), r.detach()) and float(opt.state[p]["step"]) == 0.0 and float(opt.state[p]["exp_avg"].abs().max()) == 0.0
    mt.step(grad_scale=scale, found_inf=torch.zeros((), device="cuda"))
    for p, r, g in zip(params, ref, grads):
        p_, m_, v_ = adam_step(r.detach().cpu().numpy(), g.cpu().numpy(), np.zeros(r.shape, np.float32), np.zeros(r.shape, np.float32), 1, 1e-4)
        assert float(opt.state[p]["step"]) == 1.0
        np.testing.assert_allclose(p.detach().cpu().numpy(), p_, rtol=2.4e-7, atol=1e-9)
        np.testing.assert_allclose(opt.state[p]["exp_avg"].cpu().numpy(), m_, rtol=2e-6, atol=1e-9)
        assert torch.equal(p.grad, g * 1024.0)                    # the gradient buffers are read, not rewritten
