import sys, torch, torch.nn as nn, torch.nn.functional as F
sys.path.insert(0, "sstem-restoration_amd"); sys.path.insert(0, "tests")
import hipnn.functional as HF
from hipnn import FusedSequential
W = int(sys.argv[1]) if len(sys.argv) > 1 else 32
chans = [(128, 128), (128, 128), (128, 160), (160, 128)]
def _net(seed):
    torch.manual_seed(seed)
    return [nn.Conv2d(ci, co, 3, padding=1) for ci, co in chans]
x0 = torch.randn(2, 128, 24, W, generator=torch.Generator().manual_seed(3))
res = {}
for on in (True, False):
    HF._AUTO_F16_TRAIN = on
    convs = _net(71)
    net = [FusedSequential(c, nn.ReLU()).cuda() for c in convs[:3]] + [FusedSequential(convs[3]).cuda()]
    x = x0.cuda().requires_grad_(True)
    a = net[0](x); bb = net[1](a); c = net[2](a + bb); out = net[3](c)
    out.square().mean().backward()
    res[on] = [out.detach(), x.grad] + [p.grad for m in net for p in m.parameters()]
convs = _net(71)
for cv in convs: cv.double()
x = x0.double().requires_grad_(True)
a = F.relu(convs[0](x)); bb = F.relu(convs[1](a)); c = F.relu(convs[2](a + bb)); out = convs[3](c)
out.square().mean().backward()
ref = [out, x.grad] + [p.grad for cv in convs for p in cv.parameters()]
names = ["out", "x.grad"] + ["%s%d" % (k, i) for i in range(4) for k in ("w", "b")]
for n, g, x6, r in zip(names, res[True], res[False], ref):
    s = r.abs().max().item()
    print("%-7s max|ref| %.3e  f16 err %.2e  x6 err %.2e (of max)" % (n, s, (g.double().cpu() - r).abs().max().item() / s, (x6.double().cpu() - r).abs().max().item() / s))
