import sys, torch, torch.nn as nn, torch.nn.functional as F
sys.path.insert(0, "sstem-restoration_amd"); sys.path.insert(0, "tests")
import hipnn.functional as HF
torch.manual_seed(1)
def rel(a, r): return (a.double().cpu() - r).abs().max().item() / r.abs().max().item()
for (N, Cin, H, W, Cout) in [(2, 160, 24, 32, 128), (2, 128, 24, 32, 128), (2, 128, 24, 32, 160), (2, 128, 32, 32, 256), (2, 64, 24, 64, 64)]:
    x = torch.randn(N, Cin, H, W, device="cuda"); w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05; b = torch.randn(Cout, device="cuda")
    g = torch.randn(N, Cout, H, W, device="cuda")
    xd = x.double().cpu().requires_grad_(True); wd = w.double().cpu().requires_grad_(True); bd = b.double().cpu().requires_grad_(True)
    yd = F.conv2d(xd, wd, bd, padding=1); yd.backward(g.double().cpu())
    for algo, pair in ((HF.ALGO_MFMA_BF16X6, True), (HF.ALGO_MFMA_BF16X6, False), (HF.ALGO_MFMA, True), (HF.ALGO_AUTO, True)):
        HF.set_algorithm(algo); HF._PACK_PAIR = pair
        xg = x.clone().requires_grad_(True); wg = w.clone().requires_grad_(True); bg = b.clone().requires_grad_(True)
        y = HF.conv2d_fused(xg, wg, bg)
        y.backward(g)
        print((N, Cin, H, W, Cout), "algo", algo, "pair", pair, "y %.1e gx %.1e gw %.1e gb %.1e" % (rel(y.detach(), yd.detach()), rel(xg.grad, xd.grad), rel(wg.grad, wd.grad), rel(bg.grad, bd.grad)))
