#!/usr/bin/env python
"""Developer micro-benchmark: torch's train-mode BatchNorm2d (+ ReLU) forward and backward on the fusion-step shapes, against
the streaming bound of a two-pass forward (read x for the statistics, read x + write y) and of the backward
(read x, read dy for the reductions; read x, read dy, write dx)."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sstem-restoration_amd"))
import hipnn.functional as HF


def _prewarm(seconds=0.6):
    """An idle MI355X needs a few hundred ms under load to reach its clocks: the first shapes of a run measured 30-50 % slow."""
    import time
    a = torch.randn(4096, 4096, device="cuda")
    t0 = time.time()
    while time.time() - t0 < seconds:
        (a @ a).sum().item()


_prewarm()

def t(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n

for shape in ((16, 32, 256, 256), (16, 64, 128, 128), (16, 128, 64, 64), (16, 256, 32, 32), (16, 64, 256, 256), (2, 32, 256, 256)):
    x = torch.randn(*shape, device="cuda"); bn = torch.nn.BatchNorm2d(shape[1]).cuda().train(); relu = torch.nn.ReLU(inplace=True)
    mb = x.numel() * 4 / 1e6
    with torch.no_grad():
        f = t(lambda: relu(bn(x)))
        fb = t(lambda: bn(x))
    xg = x.clone().requires_grad_(); y = relu(bn(xg)); g = torch.randn_like(y)
    b = t(lambda: torch.autograd.grad(y, xg, g, retain_graph=True))
    bn2 = torch.nn.BatchNorm2d(shape[1]).cuda().train()
    with torch.no_grad():
        nf = t(lambda: HF.batchnorm_train_act(bn2, x, HF.ACT_RELU, 0.0))
    xn = x.clone().requires_grad_(); yn = HF.batchnorm_train_act(bn2, xn, HF.ACT_RELU, 0.0)
    nb = t(lambda: torch.autograd.grad(yn, xn, g, retain_graph=True))
    print("%-20s %6.1f MB  torch: BN+ReLU fwd %.3f ms (%.0f GB/s over 3 passes), bwd %.3f ms (%.0f GB/s over 5)   native: fwd %.3f ms (%.0f GB/s), bwd %.3f ms (%.0f GB/s)" %
          (shape, mb, f, 3 * mb / f, b, 5 * mb / b, nf, 3 * mb / nf, nb, 5 * mb / nb))
