#!/usr/bin/env python
"""Developer micro-benchmark: 3x3 weight gradient (+ bias gradient) through the C-ABI, fp32 MFMA id against the opt-in bf16-operand id.
Usage: python tools/bench_wgrad_bf16.py [N Cin H W Cout] ..."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "sstem-restoration_amd"))
import torch  # noqa: E402
import sstem_native  # noqa: E402


def _prewarm(seconds=0.6):
    """An idle MI355X needs a few hundred ms under load to reach its clocks: the first shapes of a run measured 30-50 % slow."""
    import time
    a = torch.randn(4096, 4096, device="cuda")
    t0 = time.time()
    while time.time() - t0 < seconds:
        (a @ a).sum().item()


_prewarm()

lib = sstem_native.load_library()
MFMA, BF16 = 2, 3
shapes = [(8, 6, 256, 256, 32), (8, 32, 256, 256, 32), (8, 32, 128, 128, 64), (8, 64, 128, 128, 64), (8, 128, 64, 64, 128),
          (8, 256, 32, 32, 256), (8, 512, 16, 16, 512), (8, 512, 8, 8, 512), (8, 64, 128, 128, 51), (8, 51, 256, 256, 51),
          (16, 64, 256, 256, 64), (16, 128, 128, 128, 128)]
args = [int(v) for v in sys.argv[1:]]
if args:
    shapes = [tuple(args[i:i + 5]) for i in range(0, len(args), 5)]


def timeit(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for dims in shapes:
    N, Cin, H, W, Cout = dims
    x = torch.randn(N, Cin, H, W, device="cuda"); g = torch.randn(N, Cout, H, W, device="cuda")
    res = {}
    for algo in (MFMA, BF16):
        gw = torch.empty(Cout, Cin, 3, 3, device="cuda"); gb = torch.empty(Cout, device="cuda")
        ws_n = int(lib.sstem_conv3x3_wgrad_workspace_floats_algo(N, Cin, H, W, Cout, algo)); ws = torch.empty(max(ws_n, 1), device="cuda")

        def run():
            rc = lib.sstem_conv2d_backward_weight_bias_f32(x.data_ptr(), g.data_ptr(), gw.data_ptr(), gb.data_ptr(), ws.data_ptr(), ws_n,
                                                           N, Cin, H, W, Cout, 3, 3, 1, 1, torch.cuda.current_stream().cuda_stream, algo)
            sstem_native.check(rc, "wgrad")
        res[algo] = (timeit(run), gw, gb, ws_n)
    a, b = res[MFMA], res[BF16]
    err = float((a[1] - b[1]).abs().max() / a[1].abs().max()); berr = float((a[2] - b[2]).abs().max() / a[2].abs().max())
    flop = 2.0 * 9 * N * H * W * Cin * Cout
    floor = 4.0 * N * H * W * (Cin + Cout) / 8e12 * 1e3
    print("wgrad N%d %d->%d %dx%d: fp32 %.3f ms (%.0f TF)   bf16 %.3f ms (%.0f TF, x%.1f; HBM floor %.3f ms; slabs %.1f MB)   rel diff gw %.1e gb %.1e"
          % (N, Cin, Cout, H, W, a[0], flop / a[0] / 1e9, b[0], flop / b[0] / 1e9, a[0] / b[0], floor, b[3] * 4 / 1e6, err, berr), flush=True)
