#!/usr/bin/env python
"""Run one 3x3 weight gradient a few times (for rocprofv3 --pmc runs): python tools/wgrad_one.py N Cin H W Cout [x6|f16x3|fp32]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "sstem-restoration_amd"))
import torch
import sstem_native
N, Cin, H, W, Cout = [int(v) for v in sys.argv[1:6]]
kind = sys.argv[6] if len(sys.argv) > 6 else "f16x3"
lib = sstem_native.load_library()
x = torch.randn(N, Cin, H, W, device="cuda"); g = torch.randn(N, Cout, H, W, device="cuda")
algo = {"fp32": 2, "x6": 5, "f16x3": 6}[kind]
ws_n = int(lib.sstem_conv3x3_wgrad_workspace_floats_algo(N, Cin, H, W, Cout, algo)); ws = torch.empty(max(ws_n, 1), device="cuda")
gw = torch.empty(Cout, Cin, 3, 3, device="cuda"); gb = torch.empty(Cout, device="cuda")
xw = torch.zeros(1024, device="cuda"); gword = torch.zeros(1024, device="cuda")
lib.sstem_amax_f32(x.data_ptr(), x.numel(), xw.data_ptr(), None); lib.sstem_amax_f32(g.data_ptr(), g.numel(), gword.data_ptr(), None)
for _ in range(int(os.environ.get("ITERS", "6"))):
    if kind == "f16x3":
        rc = lib.sstem_conv3x3_backward_weight_scaled_masked_f32(x.data_ptr(), xw.data_ptr(), g.data_ptr(), gword.data_ptr(), None, gw.data_ptr(),
                                                                 gb.data_ptr(), ws.data_ptr(), ws_n, N, Cin, H, W, Cout, 0, None)
    else:
        rc = lib.sstem_conv2d_backward_weight_bias_f32(x.data_ptr(), g.data_ptr(), gw.data_ptr(), gb.data_ptr(), ws.data_ptr(), ws_n,
                                                       N, Cin, H, W, Cout, 3, 3, 1, 1, None, algo)
    sstem_native.check(rc, "wgrad")
torch.cuda.synchronize()
