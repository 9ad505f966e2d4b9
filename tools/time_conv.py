#!/usr/bin/env python
"""Time conv3x3 layers of one algorithm id precisely (same-box A/B of library builds: SSTEM_NATIVE_LIB=...).
   python tools/time_conv.py f16x3 8,64,512,512,64 8,32,1024,1024,32 ...   -> median ms of 7 batches of 30 calls per layer"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "sstem-restoration_amd"))
import torch
import hipnn.functional as HF
ids = {"direct": HF.ALGO_DIRECT, "fp32": HF.ALGO_MFMA, "bf16": HF.ALGO_MFMA_BF16, "x6": HF.ALGO_MFMA_BF16X6, "x3": HF.ALGO_MFMA_BF16X3, "f16x3": HF.ALGO_MFMA_F16X3}
HF.set_algorithm(ids[sys.argv[1]])
out = []
for spec in sys.argv[2:]:
    N, Cin, H, W, Cout = [int(v) for v in spec.split(",")]
    x = torch.randn(N, Cin, H, W, device="cuda"); w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05; b = torch.randn(Cout, device="cuda")
    with torch.no_grad():
        for _ in range(10):
            HF.conv2d_fused(x, w, b, None, None, HF.ACT_RELU, 0.0)
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                HF.conv2d_fused(x, w, b, None, None, HF.ACT_RELU, 0.0)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 30)
    ts.sort()
    out.append("%s %.3f" % (spec, ts[3]))
print(os.environ.get("SSTEM_NATIVE_LIB", "product").split("/")[-1], os.environ.get("SSTEM_F16_WALK", ""), " | ".join(out))
