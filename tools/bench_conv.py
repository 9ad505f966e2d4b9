#!/usr/bin/env python
"""Developer micro-benchmark of the fused conv3x3 kernel (HIP-event timed): TFLOP/s per layer shape,
optionally against torch's own conv2d (MIOpen) on the same tensors as a same-box reference point."""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "sstem-restoration_amd"))
import torch  # noqa: E402
import hipnn.functional as HF  # noqa: E402


def _prewarm(seconds=0.6):
    """An idle MI355X needs a few hundred ms under load to reach its clocks: the first shapes of a run measured 30-50 % slow."""
    import time
    a = torch.randn(4096, 4096, device="cuda")
    t0 = time.time()
    while time.time() - t0 < seconds:
        (a @ a).sum().item()


_prewarm()

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--torch", action="store_true")
ap.add_argument("--bwd", action="store_true")
ap.add_argument("--bf16", action="store_true", help="also time the opt-in bf16-operand id and print the HBM floor of the layer")
ap.add_argument("--split", action="store_true", help="also time the split-bf16 ids (fp32 operands as 3 / 2 bf16 pieces: ALGO_MFMA_BF16X6 / X3) "
                "and print each id's max error against float64 torch on one image's first 64 rows, relative to max|ref|")
ap.add_argument("--set", default="c2c3", help="c2c3 | c5 (IFNet layers of the 256x256 training step at 8 per GPU)")
a = ap.parse_args()

# (N, Cin, H, W, Cout): IFNet layers at B=8 1024^2 (C2) and the fusion nets at B=16 256^2 (C3)
SHAPES = [(8, 64, 512, 512, 64), (8, 128, 256, 256, 128), (8, 256, 128, 128, 256), (8, 512, 64, 64, 512),
          (8, 51, 1024, 1024, 51), (8, 6, 1024, 1024, 6), (8, 32, 1024, 1024, 32),
          (16, 32, 256, 256, 32), (16, 64, 256, 256, 32), (16, 64, 128, 128, 64), (16, 256, 32, 32, 256)]
if a.set == "c3b2":      # the SFF fusion step's layers at 2 samples per GPU (UNet 6-32-64-128-256, FusionNet 32..512)
    SHAPES = [(2, 6, 256, 256, 32), (2, 32, 256, 256, 32), (2, 32, 128, 128, 64), (2, 64, 128, 128, 64), (2, 64, 64, 64, 128),
              (2, 128, 64, 64, 128), (2, 128, 32, 32, 256), (2, 256, 32, 32, 256), (2, 256, 16, 16, 512), (2, 512, 16, 16, 512),
              (2, 64, 256, 256, 32), (2, 128, 128, 128, 64), (2, 256, 64, 64, 128), (2, 512, 32, 32, 256)]
if a.set == "c5":
    SHAPES = [(8, 6, 256, 256, 32), (8, 32, 256, 256, 32), (8, 32, 128, 128, 64), (8, 64, 128, 128, 64), (8, 64, 64, 64, 128),
              (8, 128, 64, 64, 128), (8, 128, 32, 32, 256), (8, 256, 32, 32, 256), (8, 256, 16, 16, 512), (8, 512, 16, 16, 512),
              (8, 512, 8, 8, 512), (8, 64, 128, 128, 51), (8, 51, 256, 256, 51)]


def timeit(fn, n):
    fn(); fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for (N, Cin, H, W, Cout) in SHAPES:
    x = torch.randn(N, Cin, H, W, device="cuda"); w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05
    b = torch.randn(Cout, device="cuda")
    flop = 2.0 * N * Cout * H * W * Cin * 9
    with torch.no_grad():
        ms = timeit(lambda: HF.conv2d_fused(x, w, b, None, None, HF.ACT_RELU, 0.0), a.iters)
        line = "conv3x3 N%d Cin%d %dx%d Cout%d: %.3f ms  %.1f TFLOP/s" % (N, Cin, H, W, Cout, ms, flop / ms / 1e9)
        if a.bf16:
            with HF.algorithm(HF.ALGO_MFMA_BF16):
                mb = timeit(lambda: HF.conv2d_fused(x, w, b, None, None, HF.ACT_RELU, 0.0), a.iters)
            floor_ms = 4.0 * N * H * W * (Cin + Cout) / 8e12 * 1e3
            line += "   | bf16 operands %.3f ms  %.1f TFLOP/s  (x%.1f; fp32-tensor HBM floor %.3f ms)" % (mb, flop / mb / 1e9, ms / mb, floor_ms)
        if a.split:
            hh = min(H, 64)
            xs = x[:1, :, :hh].contiguous()
            ref = torch.relu(torch.nn.functional.conv2d(xs.double().cpu(), w.double().cpu(), b.double().cpu(), padding=1))[:, :, :hh - 1]
            scale = ref.abs().max().item()

            def err_of():
                return (HF.conv2d_fused(xs, w, b, None, None, HF.ACT_RELU, 0.0)[:, :, :hh - 1].double().cpu() - ref).abs().max().item() / scale
            line += "   | fp32 err %.1e" % err_of()
            for name, algo in (("bf16x6", HF.ALGO_MFMA_BF16X6), ("bf16x3", HF.ALGO_MFMA_BF16X3), ("f16x3", HF.ALGO_MFMA_F16X3)):
                with HF.algorithm(algo):
                    HF.measured_amax_word(x)       # the bound a producing layer would have left (the fp16 id: no measuring pass in the timing)
                    m2 = timeit(lambda: HF.conv2d_fused(x, w, b, None, None, HF.ACT_RELU, 0.0), a.iters)
                    line += "   | %s %.3f ms  %.1f TFLOP/s (x%.2f) err %.1e" % (name, m2, flop / m2 / 1e9, ms / m2, err_of())
        if a.torch:
            mt = timeit(lambda: torch.relu_(torch.nn.functional.conv2d(x, w, b, padding=1)), a.iters)
            line += "   | torch conv2d+relu %.3f ms  %.1f TFLOP/s" % (mt, flop / mt / 1e9)
    if a.bwd:
        xg = x.clone().requires_grad_(); wg = w.clone().requires_grad_(); bg = b.clone().requires_grad_()
        go = torch.randn(N, Cout, H, W, device="cuda")

        def fb():
            xg.grad = wg.grad = bg.grad = None
            HF.conv2d_fused(xg, wg, bg, None, None, HF.ACT_RELU, 0.0).backward(go)
        ms = timeit(fb, max(2, a.iters // 3))
        line += "   | fwd+bwd %.3f ms (%.1f TFLOP/s on 3x fwd flops)" % (ms, 3 * flop / ms / 1e9)
    print(line, flush=True)
