#!/usr/bin/env python
"""Time 3x3 layers with few output channels under ALGO_AUTO (the streaming kernel) and under the fp16 two-piece id.
python tools/time_stream_small.py 8,32,1024,1024,1 ..."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sstem-restoration_amd"))
import torch, hipnn.functional as HF
res = []
for spec in sys.argv[1:]:
    N, Cin, H, W, Cout = [int(v) for v in spec.split(",")]
    x = torch.randn(N, Cin, H, W, device="cuda"); w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05; b = torch.randn(Cout, device="cuda")
    row = [spec]
    for algo in (HF.ALGO_AUTO, HF.ALGO_MFMA_F16X3):
        owner = torch.nn.Module()
        with HF.algorithm(algo), torch.no_grad():
            for _ in range(5):
                HF.conv2d_fused(x, w, b, None, None, HF.ACT_RELU, 0.0, owner=owner)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(40):
                HF.conv2d_fused(x, w, b, None, None, HF.ACT_RELU, 0.0, owner=owner)
            e1.record(); torch.cuda.synchronize()
        row.append("%.3f" % (e0.elapsed_time(e1) / 40))
    res.append(" ".join(row))
print("variant=%s (auto / f16x3 ms): " % os.environ.get("SSTEM_STREAM_SMALL_VARIANT", "0") + " | ".join(res))
