import sys, torch, torch.nn as nn, torch.nn.functional as F
sys.path.insert(0, "sstem-restoration_amd"); sys.path.insert(0, "tests")
import hipnn.functional as HF
from hipnn import FusedSequential
chans = [(128, 128), (128, 128), (128, 160), (160, 128)]
def _net(seed):
    torch.manual_seed(seed)
    return [nn.Conv2d(ci, co, 3, padding=1) for ci, co in chans]
x0 = torch.randn(2, 128, 24, 32, generator=torch.Generator().manual_seed(3))
def rel(a, r): return (a.double().cpu() - r).abs().max().item() / r.abs().max().item()
for lossname in ("mean_sq", "sum", "sum_sq"):
    def loss(o): return o.square().mean() if lossname == "mean_sq" else (o.sum() if lossname == "sum" else o.square().sum())
    convs = _net(71)
    for cv in convs: cv.double()
    xd = x0.double().requires_grad_(True)
    a = F.relu(convs[0](xd)); bb = F.relu(convs[1](a)); s = a + bb; c = F.relu(convs[2](s)); out = convs[3](c)
    for t in (a, bb, s, c, out): t.retain_grad()
    loss(out).backward()
    ref = [t.grad for t in (out, c, s, bb, a, xd)]
    for algo in (HF.ALGO_MFMA_BF16X6, HF.ALGO_MFMA):
        HF._AUTO_F16_TRAIN = False; HF.set_algorithm(algo)
        convs = _net(71)
        net = [FusedSequential(cv, nn.ReLU()).cuda() for cv in convs[:3]] + [FusedSequential(convs[3]).cuda()]
        x = x0.cuda().requires_grad_(True)
        a = net[0](x); bb = net[1](a); s = a + bb; c = net[2](s); out = net[3](c)
        for t in (a, bb, s, c, out): t.retain_grad()
        loss(out).backward()
        got = [t.grad for t in (out, c, s, bb, a, x)]
        print(lossname, "algo", algo, " ".join("%s %.1e" % (n, rel(g, r)) for n, g, r in zip(("g_out", "g_c", "g_s", "g_bb", "g_a", "g_x"), got, ref)),
              "| max|g_out| %.2e" % ref[0].abs().max().item())
