#!/usr/bin/env python
"""Time inference ConvTranspose2d(k3,s2,p1,op1) launches (hipnn routes: the sub-pixel form on the fp16 two-piece id, or the exact-fp32 native kernels with
SSTEM_CONVT_SUBPIXEL=0).   python tools/time_convt.py 8,64,512,512,32 ...   -> median ms of 7 batches of 20 calls per shape (N,Cin,H,W,C)"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "sstem-restoration_amd"))
import torch
import hipnn.functional as HF
out = []
for spec in sys.argv[1:]:
    N, Cin, H, W, C = [int(v) for v in spec.split(",")]
    x = torch.randn(N, Cin, H, W, device="cuda")
    m = torch.nn.ConvTranspose2d(Cin, C, 3, stride=2, padding=1, output_padding=1).cuda().requires_grad_(False)
    sc = torch.rand(C, device="cuda") + 0.5; sh = torch.randn(C, device="cuda")
    with torch.no_grad():
        for _ in range(5):
            HF.conv_transpose3x3s2_fused(x, m.weight, m.bias, sc, sh, HF.ACT_RELU, 0.0, owner=m)
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                HF.conv_transpose3x3s2_fused(x, m.weight, m.bias, sc, sh, HF.ACT_RELU, 0.0, owner=m)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20)
    ts.sort()
    out.append("%s %.3f" % (spec, ts[3]))
print(os.environ.get("SSTEM_NATIVE_LIB", "product").split("/")[-1], "walk_ct=" + os.environ.get("SSTEM_SPLIT_WALK_CT", "default"), "subpixel=" + os.environ.get("SSTEM_CONVT_SUBPIXEL", "1"), " | ".join(out))
