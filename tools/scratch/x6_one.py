import sys, torch, torch.nn as nn, torch.nn.functional as F
sys.path.insert(0, "sstem-restoration_amd"); sys.path.insert(0, "tests")
import hipnn.functional as HF
def rel(a, r): return (a.double().cpu() - r).abs().max().item() / r.abs().max().item()
torch.manual_seed(5)
N, Cin, H, W, Cout = 2, 128, 24, 32, 160
x = torch.randn(N, Cin, H, W, device="cuda"); w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05; b = torch.randn(Cout, device="cuda")
g = torch.randn(N, Cout, H, W, device="cuda")
for act in (HF.ACT_NONE, HF.ACT_RELU):
    for sparse in (False, True):
        gg = g * (torch.rand_like(g) > 0.5) if sparse else g
        xd = x.double().cpu().requires_grad_(True); wd = w.double().cpu().requires_grad_(True); bd = b.double().cpu()
        yd = F.conv2d(xd, wd, bd, padding=1)
        if act == HF.ACT_RELU: yd = F.relu(yd)
        yd.backward(gg.double().cpu())
        for algo in (HF.ALGO_MFMA_BF16X6, HF.ALGO_MFMA):
            for fusion in (True, False):
                HF._AUTO_F16_TRAIN = False; HF._MASK_FUSION = fusion; HF.set_algorithm(algo)
                xg = x.clone().requires_grad_(True); wg = w.clone().requires_grad_(True)
                y = HF.conv2d_fused(xg, wg, b, None, None, act, 0.0)
                y.backward(gg)
                print("act", act, "sparse", sparse, "algo", algo, "fusion", fusion, "y %.1e gx %.1e gw %.1e" % (rel(y.detach(), yd.detach()), rel(xg.grad, xd.grad), rel(wg.grad, wd.grad)))
