import os, sys, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "sstem-restoration_amd"))
import hipnn.functional as HF
import torch.nn.functional as F


def _prewarm(seconds=0.6):
    """An idle MI355X needs a few hundred ms under load to reach its clocks: the first shapes of a run measured 30-50 % slow."""
    import time
    a = torch.randn(4096, 4096, device="cuda")
    t0 = time.time()
    while time.time() - t0 < seconds:
        (a @ a).sum().item()


_prewarm()
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n
for shape in ((8,51,512,512),(8,64,256,256),(8,512,32,32)):
    x=torch.randn(*shape,device="cuda")
    a=t(lambda: HF.upsample_bilinear2x(x)); b=t(lambda: F.interpolate(x,scale_factor=2,mode="bilinear",align_corners=True))
    byt=x.numel()*4*5
    print("%s native %.3f ms (%.0f GB/s)  torch %.3f ms (%.0f GB/s)"%(shape,a,byt/a/1e6,b,byt/b/1e6))

# backward of torch's op (what training uses today): grad_out [N,C,2H,2W] -> grad_in [N,C,H,W]
print("-- backward (torch autograd)")
for shape in ((8, 51, 128, 128), (8, 64, 128, 128), (8, 128, 64, 64), (8, 256, 32, 32), (8, 512, 16, 16), (8, 51, 512, 512)):
    x = torch.randn(*shape, device="cuda", requires_grad=True)
    y = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
    g = torch.randn_like(y)
    b = t(lambda: torch.autograd.grad(y, x, g, retain_graph=True))
    byt = x.numel() * 4 * 5
    print("%s torch backward %.3f ms (%.0f GB/s of in+out bytes)" % (shape, b, byt / b / 1e6))
