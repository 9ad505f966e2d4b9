#!/usr/bin/env python
"""Developer micro-benchmark: 3x3 weight gradient through the C-ABI under the fp32 MFMA id and the split-bf16 ids (X6 / X3), with
each id's max error against float64 torch (relative to max|ref|).  Usage: python tools/bench_wgrad_split.py [--set c3|c5|c3b2] [N Cin H W Cout ...]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "sstem-restoration_amd"))
import torch  # noqa: E402
import sstem_native  # noqa: E402


def _prewarm(seconds=0.6):
    import time
    a = torch.randn(4096, 4096, device="cuda")
    t0 = time.time()
    while time.time() - t0 < seconds:
        (a @ a).sum().item()


_prewarm()
lib = sstem_native.load_library()
MFMA, X3, X6, F16X3 = 2, 4, 5, 6
SETS = {
    "sp": [(16, 64, 256, 256, 64), (16, 128, 128, 128, 128), (16, 256, 64, 64, 256), (16, 512, 32, 32, 512), (16, 1024, 32, 32, 512),
           (16, 128, 256, 256, 64), (16, 64, 256, 256, 51)],
    "c3": [(16, 6, 256, 256, 32), (16, 32, 256, 256, 32), (16, 32, 128, 128, 64), (16, 64, 128, 128, 64), (16, 64, 64, 64, 128),
           (16, 128, 64, 64, 128), (16, 128, 32, 32, 256), (16, 256, 32, 32, 256), (16, 64, 256, 256, 32), (16, 128, 128, 128, 64)],
    "c5": [(8, 6, 256, 256, 32), (8, 32, 256, 256, 32), (8, 32, 128, 128, 64), (8, 64, 128, 128, 64), (8, 64, 64, 64, 128),
           (8, 128, 64, 64, 128), (8, 128, 32, 32, 256), (8, 256, 32, 32, 256), (8, 256, 16, 16, 512), (8, 512, 16, 16, 512),
           (8, 512, 8, 8, 512), (8, 64, 128, 128, 51), (8, 51, 256, 256, 51)],
    "c3b2": [(2, 32, 256, 256, 32), (2, 64, 128, 128, 64), (2, 128, 64, 64, 128), (2, 256, 32, 32, 256), (2, 64, 256, 256, 32)],
}
argv = sys.argv[1:]
shapes = SETS["c5"]
if argv and argv[0] == "--set":
    shapes = SETS[argv[1]]; argv = argv[2:]
if argv:
    a = [int(v) for v in argv]
    shapes = [tuple(a[i:i + 5]) for i in range(0, len(a), 5)]


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
    xs, gs = x[:1].double().cpu().requires_grad_(False), g[:1].double().cpu()
    wref = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
    torch.nn.functional.conv2d(xs, wref, padding=1).backward(gs)
    ref = wref.grad
    line = "wgrad N%d %d->%d %dx%d:" % (N, Cin, Cout, H, W)
    base = None
    xw = torch.zeros(1024, device="cuda"); gw_word = torch.zeros(1024, device="cuda")
    lib.sstem_amax_f32(x.data_ptr(), x.numel(), xw.data_ptr(), None); lib.sstem_amax_f32(g.data_ptr(), g.numel(), gw_word.data_ptr(), None)
    for name, algo in (("fp32", MFMA), ("x6", X6), ("x3", X3), ("f16x3", F16X3)):
        ws_n = int(lib.sstem_conv3x3_wgrad_workspace_floats_algo(N, Cin, H, W, Cout, algo)); ws = torch.empty(max(ws_n, 1), device="cuda")
        gw = torch.empty(Cout, Cin, 3, 3, device="cuda"); gb = torch.empty(Cout, device="cuda")

        def run(n_img=N, xx=x, gg=g):
            if algo == F16X3:       # the recorded launches' two-piece fp16 weight gradient (bounds of the whole tensors: upper bounds of any slice)
                rc = lib.sstem_conv3x3_backward_weight_scaled_masked_f32(xx.data_ptr(), xw.data_ptr(), gg.data_ptr(), gw_word.data_ptr(), None,
                                                                         gw.data_ptr(), gb.data_ptr(), ws.data_ptr(), ws_n, n_img, Cin, H, W, Cout,
                                                                         0, torch.cuda.current_stream().cuda_stream)
            else:
                rc = lib.sstem_conv2d_backward_weight_bias_f32(xx.data_ptr(), gg.data_ptr(), gw.data_ptr(), gb.data_ptr(), ws.data_ptr(), ws_n,
                                                               n_img, Cin, H, W, Cout, 3, 3, 1, 1, torch.cuda.current_stream().cuda_stream, algo)
            sstem_native.check(rc, "wgrad")
        ms = timeit(run)
        base = base or ms
        x1, g1 = x[:1].contiguous(), g[:1].contiguous()
        ws1_n = int(lib.sstem_conv3x3_wgrad_workspace_floats_algo(1, Cin, H, W, Cout, algo))
        if ws1_n > ws_n:
            ws = torch.empty(ws1_n, device="cuda"); ws_n = ws1_n
        run(1, x1, g1)
        err = float((gw.double().cpu() - ref).abs().max() / ref.abs().max())
        line += "   %s %.3f ms (x%.2f) err %.1e" % (name, ms, base / ms, err)
    print(line, flush=True)
