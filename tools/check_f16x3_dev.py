#!/usr/bin/env python
"""Developer aid for dev builds of the split kernels (SSTEM_NATIVE_LIB=build_ablate/libsstem_split_<name>.so): F16X3 layers against fp64
torch (plain, and with the skip addition in the store), then the time of both forms."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sstem-restoration_amd"))
import torch, hipnn.functional as HF
torch.manual_seed(3)
F = torch.nn.functional
for (N, Cin, H, W, Cout) in [(2, 32, 256, 256, 32), (2, 64, 128, 128, 64), (1, 6, 64, 64, 32), (2, 32, 40, 96, 32), (1, 128, 64, 64, 128), (8, 32, 512, 512, 32)]:
    x = torch.randn(N, Cin, H, W, device="cuda"); w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05; b = torch.randn(Cout, device="cuda")
    res = torch.randn(N, Cout, H, W, device="cuda")
    with HF.algorithm(HF.ALGO_MFMA_F16X3), torch.no_grad():
        y = HF.conv2d_fused(x, w, b, None, None, HF.ACT_RELU, 0.0)
        z = HF.conv2d_fused(x, w, b, None, None, HF.ACT_LEAKY, 0.2, residual=res, res_scale=0.5)
    ref = torch.relu(F.conv2d(x.double(), w.double(), b.double(), padding=1))
    refz = (F.leaky_relu(F.conv2d(x.double(), w.double(), b.double(), padding=1), 0.2) + res.double()) * 0.5
    print((N, Cin, H, W, Cout), "max err / max: plain %.2e, residual %.2e" % (float((y.double() - ref).abs().max() / ref.abs().max()),
                                                                             float((z.double() - refz).abs().max() / refz.abs().max())))
for (N, Cin, H, W, Cout) in [(8, 32, 1024, 1024, 32), (8, 64, 512, 512, 64), (8, 128, 256, 256, 128)]:
    x = torch.randn(N, Cin, H, W, device="cuda"); w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05; b = torch.randn(Cout, device="cuda")
    res = torch.randn(N, Cout, H, W, device="cuda"); owner = torch.nn.Module()
    out = []
    for r in (None, res):
        with HF.algorithm(HF.ALGO_MFMA_F16X3), torch.no_grad():
            for _ in range(5):
                HF.conv2d_fused(x, w, b, None, None, HF.ACT_LEAKY, 0.2, residual=r, res_scale=0.5, owner=owner)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                HF.conv2d_fused(x, w, b, None, None, HF.ACT_LEAKY, 0.2, residual=r, res_scale=0.5, owner=owner)
            e1.record(); torch.cuda.synchronize()
            out.append(e0.elapsed_time(e1) / 50)
    print((N, Cin, H, W, Cout), "plain %.3f ms, with residual %.3f ms" % tuple(out))
