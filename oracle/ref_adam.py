"""CPU restatement of the optimizer update of a training step.  TEST INFRASTRUCTURE: only tests/ may import it.

Reference: Trainer.py:492-497 builds `optim.Adam(parameters_by_names(network_names), learning_rate * lr_factor)` -- torch's defaults
betas (0.9, 0.999), eps 1e-8, weight_decay 0, amsgrad False -- and Trainer.py:150 calls `optimizer.step()` once per batch.  The
algorithm lives in the reference's dependency (torch.optim.Adam, torch/optim/adam.py `_single_tensor_adam`; PyTorch 2.x):
    t += 1;  g = g + wd * p (wd != 0);  m = lerp(m, g, 1 - b1);  v = b2 * v + (1 - b2) * g * g
    p = p - (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
restated in numpy (fp32 storage, the arithmetic in `dtype`: float32 reproduces torch's roundings up to the last place, float64 is
the yardstick).  PINNED: tests/test_adam.py runs it beside torch.optim.Adam itself (the library the reference calls, on the CPU)
for ten steps on random tensors."""
import numpy as np


def adam_step(p, g, m, v, t, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, dtype=np.float32):
    """One update; p, g, m, v arrays of one shape, t the step count AFTER the increment (1 for the first update).
    Returns the new (p, m, v) as float32 arrays."""
    f = dtype
    p_, g_, m_, v_ = (np.asarray(a, dtype=np.float32).astype(f) for a in (p, g, m, v))
    if weight_decay != 0.0:
        g_ = g_ + f(weight_decay) * p_
    m_ = m_ + f(1.0 - beta1) * (g_ - m_)
    v_ = f(beta2) * v_ + f(1.0 - beta2) * g_ * g_
    bc1 = 1.0 - beta1 ** t
    bc2_sqrt = np.sqrt(1.0 - beta2 ** t)
    step_size = f(lr / bc1)
    denom = np.sqrt(v_) / f(bc2_sqrt) + f(eps)
    p_ = p_ - step_size * m_ / denom
    return p_.astype(np.float32), m_.astype(np.float32), v_.astype(np.float32)
