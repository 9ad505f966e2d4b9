import os, sys
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "sstem-restoration_amd"))
import torch, hipnn.functional as HF, steps
orig = HF._auto_algo
seen = []
def logged(N, Cin, H, W, Cout):
    a = orig(N, Cin, H, W, Cout); seen.append(((N, Cin, H, W, Cout), a)); return a
HF._auto_algo = logged
fw = steps.IFNetForward(torch.device("cuda:0"), batch=8, size=1024)
fw.step(); torch.cuda.synchronize()
for k, a in seen:
    if a != HF.ALGO_MFMA_BF16X6: print(k, a)
print(len(seen), "3x3 launches resolved under AUTO")
