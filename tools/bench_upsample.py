import os, sys, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "sstem-restoration_amd"))
import hipnn.functional as HF
import torch.nn.functional as F
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
