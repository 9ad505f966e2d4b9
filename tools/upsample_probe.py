"""Developer probe: where does the native bilinear x2 kernel differ most from torch (CPU fp32, GPU fp32, fp64)?"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "sstem-restoration_amd"))
import numpy as np, torch, torch.nn.functional as F
import hipnn.functional as HF
g = torch.Generator().manual_seed(9)
x = torch.randn(1, 51, 16, 32, generator=g)
got = HF.upsample_bilinear2x(x.cuda()).cpu()
cpu32 = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
gpu32 = F.interpolate(x.cuda(), scale_factor=2, mode="bilinear", align_corners=True).cpu()
f64 = F.interpolate(x.double(), scale_factor=2, mode="bilinear", align_corners=True)
for name, ref in (("torch cpu fp32", cpu32), ("torch gpu fp32", gpu32), ("fp64", f64)):
    d = (got.double() - ref.double()).abs()
    i = int(d.argmax()); idx = np.unravel_index(i, d.shape)
    print("%-15s max |native - ref| %.3e at %s: native %.9g ref %.9g" % (name, float(d.max()), idx, float(got[idx]), float(ref[idx])))
d = (cpu32.double() - f64).abs(); print("torch cpu fp32 vs fp64: %.3e;  torch gpu fp32 vs fp64: %.3e;  native vs fp64: %.3e" % (float(d.max()), float((gpu32.double() - f64).abs().max()), float((got.double() - f64).abs().max())))
c, oy, ox = idx[1], idx[2], idx[3]
H, W = 16, 32
ry = np.float32(H - 1) / np.float32(2 * H - 1); rx = np.float32(W - 1) / np.float32(2 * W - 1)
print("worst column ox=%d: sx fp32 %.9g (fp64 %.12g), oy=%d: sy fp32 %.9g (fp64 %.12g)" % (ox, float(np.float32(rx * np.float32(ox))), (W - 1) / (2 * W - 1) * ox, oy, float(np.float32(ry * np.float32(oy))), (H - 1) / (2 * H - 1) * oy))
