import os, sys, time
REPO = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(REPO, "sstem-restoration_amd")); sys.path.insert(0, REPO)
import torch
from libs.sepconv.fused import interp_apply_gray
B, S = 8, 1024
torch.manual_seed(0)
def t(fn, n=100):
    t0 = time.time()
    while time.time() - t0 < 0.6: fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
g1 = torch.rand(B, 1, S, S, device="cuda"); g2 = torch.rand(B, 1, S, S, device="cuda")
ks = [torch.softmax(torch.randn(B, 51, S, S, device="cuda"), 1) for _ in range(4)]
with torch.no_grad():
    print("random softmax coefficients, random frames: %.4f ms" % t(lambda: interp_apply_gray(g1, g2, *ks)))
    z1 = torch.zeros_like(g1); z2 = torch.zeros_like(g2)
    print("random coefficients, ZERO frames:            %.4f ms" % t(lambda: interp_apply_gray(z1, z2, *ks)))
    kz = [torch.zeros_like(k) for k in ks]
    print("ZERO coefficients, random frames:            %.4f ms" % t(lambda: interp_apply_gray(g1, g2, *kz)))
    print("ZERO coefficients, ZERO frames:              %.4f ms" % t(lambda: interp_apply_gray(z1, z2, *kz)))
    kc = [torch.full_like(k, 1.0 / 51) for k in ks]
    print("constant 1/51 coefficients, random frames:   %.4f ms" % t(lambda: interp_apply_gray(g1, g2, *kc)))
