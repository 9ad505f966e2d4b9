#!/usr/bin/env python
"""Does the time of an F16X3 layer depend on the DATA?  (the >= 64-channel layers run at the socket's power cap: operands full of zeros
or of one repeated value toggle fewer matrix-core inputs.)  python tools/time_conv_data.py 8,64,512,512,64"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sstem-restoration_amd"))
import torch, hipnn.functional as HF
for spec in sys.argv[1:]:
    N, Cin, H, W, Cout = [int(v) for v in spec.split(",")]
    w0 = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05; b = torch.randn(Cout, device="cuda")
    row = [spec]
    for name, x, w in (("randn", torch.randn(N, Cin, H, W, device="cuda"), w0),
                       ("relu(randn)", torch.relu(torch.randn(N, Cin, H, W, device="cuda")), w0),
                       ("zeros", torch.zeros(N, Cin, H, W, device="cuda"), w0),
                       ("ones", torch.ones(N, Cin, H, W, device="cuda"), w0),
                       ("randn, w=0", torch.randn(N, Cin, H, W, device="cuda"), torch.zeros_like(w0)),
                       ("exact fp16 values", torch.randn(N, Cin, H, W, device="cuda").half().float(), w0.half().float())):
        owner = torch.nn.Module()
        with HF.algorithm(HF.ALGO_MFMA_F16X3), torch.no_grad():
            for _ in range(10):
                HF.conv2d_fused(x, w, b, None, None, HF.ACT_RELU, 0.0, owner=owner)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(100):
                HF.conv2d_fused(x, w, b, None, None, HF.ACT_RELU, 0.0, owner=owner)
            e1.record(); torch.cuda.synchronize()
        row.append("%s %.3f" % (name, e0.elapsed_time(e1) / 100))
    print(" | ".join(row))
