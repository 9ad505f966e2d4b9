#!/usr/bin/env python
"""Developer micro-benchmark: 3x3 weight gradient through the C-ABI, MFMA (split-K slabs + reduce) against the direct kernel
(one workgroup per (co, ci) pair), on given layer shapes.  Usage: python tools/bench_wgrad.py [N Cin H W Cout] ..."""
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
MFMA, DIRECT = 2, 1
shapes = [(8, 6, 256, 256, 6), (8, 6, 256, 256, 32), (8, 32, 128, 128, 32), (8, 32, 128, 128, 64), (8, 51, 256, 256, 51),
          (8, 64, 128, 128, 64), (8, 64, 128, 128, 51), (16, 6, 256, 256, 32), (16, 32, 256, 256, 32), (2, 6, 256, 256, 32)]
args = [int(v) for v in sys.argv[1:]]
if args:
    shapes = [tuple(args[i:i + 5]) for i in range(0, len(args), 5)]


def run(algo, x, g, gw, ws, ws_n, dims):
    N, Cin, H, W, Cout = dims
    rc = lib.sstem_conv2d_backward_weight_f32(x.data_ptr(), g.data_ptr(), gw.data_ptr(), ws.data_ptr() if ws is not None else None, ws_n,
                                              N, Cin, H, W, Cout, 3, 3, 1, 1, torch.cuda.current_stream().cuda_stream, algo)
    sstem_native.check(rc, "wgrad")


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
    gw1 = torch.empty(Cout, Cin, 3, 3, device="cuda"); gw2 = torch.empty_like(gw1)
    ws_n = int(lib.sstem_conv3x3_wgrad_workspace_floats(N, Cin, H, W, Cout)); ws = torch.empty(max(ws_n, 1), device="cuda")
    a = timeit(lambda: run(MFMA, x, g, gw1, ws, ws_n, dims))
    b = timeit(lambda: run(DIRECT, x, g, gw2, None, 0, dims))
    err = float((gw1 - gw2).abs().max() / gw2.abs().max())
    print("wgrad N%d %d->%d %dx%d: mfma %.3f ms   direct %.3f ms   (max rel diff %.1e)" % (N, Cin, Cout, H, W, a, b, err))
