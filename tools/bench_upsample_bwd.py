import os, sys
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "sstem-restoration_amd"))
import torch, hipnn.functional as HF, torch.nn.functional as F
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n
for shape in ((16,128,128,128),(16,64,128,128),(8,51,128,128),(8,128,64,64),(8,256,32,32),(8,512,16,16),(2,7,9,14)):
    x = torch.randn(*shape, device="cuda", requires_grad=True)
    y = HF._UpsampleBilinear2x.apply(x)
    g = torch.randn_like(y)
    a = t(lambda: torch.autograd.grad(y, x, g, retain_graph=True))
    xr = x.detach().clone().requires_grad_(True)
    yr = F.interpolate(xr, scale_factor=2, mode="bilinear", align_corners=True)
    gr, = torch.autograd.grad(yr, xr, g)
    gn, = torch.autograd.grad(y, x, g, retain_graph=True)
    err = (gn - gr).abs().max().item() / gr.abs().max().item()
    byt = x.numel() * 4 * 5
    print("%s native backward %.3f ms (%.0f GB/s) err vs aten %.1e" % (shape, a, byt / a / 1e6, err))
