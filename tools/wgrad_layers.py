#!/usr/bin/env python
"""Developer tool: the fp32 weight gradient (plan's kernel + reduce) of each layer of a network, a few calls each -- meant to be run
under `rocprofv3 --kernel-trace` so that the kernel and the reduce launch of every layer can be read separately.
Usage: python tools/wgrad_layers.py [ifnet|fusion] [batch]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "sstem-restoration_amd"))
sys.path.insert(0, os.path.join(REPO, "tools"))
import torch  # noqa: E402
import sstem_native  # noqa: E402
import sweep_wgrad as SW  # noqa: E402

lib = sstem_native.load_library()
which = sys.argv[1] if len(sys.argv) > 1 else "ifnet"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8
layers = SW.IFNET if which == "ifnet" else SW.LAYERS
a = torch.randn(4096, 4096, device="cuda")
for _ in range(30):
    (a @ a).sum().item()
for Cin, S, Cout in layers:
    x = torch.randn(N, Cin, S, S, device="cuda"); g = torch.randn(N, Cout, S, S, device="cuda")
    gw = torch.empty(Cout, Cin, 3, 3, device="cuda"); gb = torch.empty(Cout, device="cuda")
    ws_n = int(lib.sstem_conv3x3_wgrad_workspace_floats(N, Cin, S, S, Cout)); ws = torch.empty(max(ws_n, 1), device="cuda")
    for _ in range(5):
        rc = lib.sstem_conv2d_backward_weight_bias_f32(x.data_ptr(), g.data_ptr(), gw.data_ptr(), gb.data_ptr(), ws.data_ptr(), ws_n,
                                                       N, Cin, S, S, Cout, 3, 3, 1, 1, torch.cuda.current_stream().cuda_stream, 2)
        sstem_native.check(rc, "wgrad")
    torch.cuda.synchronize()
    print("layer N=%d %d->%d %dx%d slab %.1f MB" % (N, Cin, Cout, S, S, ws_n * 4 / 1e6), flush=True)
