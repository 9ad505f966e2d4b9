#!/usr/bin/env python
"""Run one conv3x3 layer shape a few times (for rocprofv3 --pmc runs)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "sstem-restoration_amd"))
import torch
import hipnn.functional as HF
N, Cin, H, W, Cout = [int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (8, 128, 256, 256, 128))]
if "bf16" in sys.argv[6:]:
    HF.set_algorithm(HF.ALGO_MFMA_BF16)
if "x6" in sys.argv[6:]:
    HF.set_algorithm(HF.ALGO_MFMA_BF16X6)
if "x3" in sys.argv[6:]:
    HF.set_algorithm(HF.ALGO_MFMA_BF16X3)
if "f16x3" in sys.argv[6:]:
    HF.set_algorithm(HF.ALGO_MFMA_F16X3)
x = torch.randn(N, Cin, H, W, device="cuda"); w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05; b = torch.randn(Cout, device="cuda")
with torch.no_grad():
    for _ in range(5):
        y = HF.conv2d_fused(x, w, b, None, None, HF.ACT_RELU, 0.0)
torch.cuda.synchronize()
