#!/usr/bin/env python
"""Developer sweep (python tools/sweep_wgrad.py [ifnet] [batch ...]): every configuration of the fp32 3x3 weight-gradient kernels (SSTEM_WGRAD_FORCE) x a ladder of slab counts on
the layer shapes of the SFF fusion step, against what the plan's cost model picks.  Used to calibrate the constants of
sstem::wgrad_plan (csrc/conv_kernels.hip).  Usage: python tools/sweep_wgrad.py [batch ...]   (default 2 16)"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "sstem-restoration_amd"))
import torch  # noqa: E402
import sstem_native  # noqa: E402

lib = sstem_native.load_library()
LAYERS = [(6, 256, 32), (32, 256, 32), (32, 128, 64), (64, 128, 64), (64, 64, 128), (128, 64, 128), (128, 32, 256), (256, 32, 128),
          (256, 64, 128), (128, 128, 64), (64, 256, 32), (32, 256, 1)]
IFNET = [(6, 256, 6), (6, 256, 32), (32, 128, 32), (32, 128, 64), (64, 64, 64), (64, 64, 128), (128, 32, 128), (128, 32, 256), (256, 16, 256),
         (256, 16, 512), (512, 8, 512), (512, 16, 512), (512, 16, 256), (256, 32, 256), (256, 32, 128), (128, 64, 128), (128, 64, 64),
         (64, 128, 64), (64, 128, 51), (51, 256, 51)]
CONFIGS = [(2, 2, 1, 2), (2, 2, 2, 2), (2, 2, 2, 1), (1, 2, 4, 1), (2, 1, 4, 1), (1, 1, 8, 1)]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    a = torch.randn(4096, 4096, device="cuda")
    for _ in range(30):
        (a @ a).sum().item()                      # clocks
    args = sys.argv[1:]
    layers = LAYERS
    if args and args[0] == "ifnet":              # the SFF IFNet's layers (BASELINE config 5: 8 per GPU at 256x256)
        layers, args = IFNET, args[1:] or ["8"]
    for N in [int(v) for v in args] or [2, 16]:
        for Cin, S, Cout in layers:
            x = torch.randn(N, Cin, S, S, device="cuda"); g = torch.randn(N, Cout, S, S, device="cuda")
            gw = torch.empty(Cout, Cin, 3, 3, device="cuda"); gb = torch.empty(Cout, device="cuda")

            def run():
                ws_n = int(lib.sstem_conv3x3_wgrad_workspace_floats(N, Cin, S, S, Cout))
                ws = torch.empty(max(ws_n, 1), device="cuda")
                rc = lib.sstem_conv2d_backward_weight_bias_f32(x.data_ptr(), g.data_ptr(), gw.data_ptr(), gb.data_ptr(), ws.data_ptr(), ws_n,
                                                               N, Cin, S, S, Cout, 3, 3, 1, 1, torch.cuda.current_stream().cuda_stream, 2)
                sstem_native.check(rc, "wgrad")
            os.environ.pop("SSTEM_WGRAD_FORCE", None)
            t_plan = timeit(run)
            res = []
            for c in CONFIGS:
                if (c[0] == 1 and Cout > 32 and False):
                    continue
                for ks in (1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024):     # (the plan clamps to the tile count)
                    os.environ["SSTEM_WGRAD_FORCE"] = "%d,%d,%d,%d,%d" % (c + (ks,))
                    try:
                        res.append((timeit(run, 10), c, ks))
                    except RuntimeError:
                        pass
            os.environ.pop("SSTEM_WGRAD_FORCE", None)
            res.sort()
            best = res[0]
            print("N=%d %3d->%3d %3dx%3d: plan %.1f us | best %.1f us %s slabs %d | next %s" % (
                N, Cin, Cout, S, S, t_plan, best[0], best[1], best[2], "; ".join("%.1f %s/%d" % r for r in res[1:4])), flush=True)


if __name__ == "__main__":
    main()
