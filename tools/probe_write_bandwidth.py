import torch, time
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n
a = torch.randn(4096, 4096, device="cuda"); t0=time.time()
while time.time()-t0 < 0.6: (a@a).sum().item()
y = torch.empty(8, 51, 1024, 1024, device="cuda"); x = torch.randn(8, 51, 512, 512, device="cuda")
print("fill 1.7 GB: %.3f ms" % t(lambda: y.fill_(1.0)))
z = torch.empty_like(y)
print("copy 1.7 GB -> 1.7 GB: %.3f ms" % t(lambda: z.copy_(y)))
print("nearest upsample (torch): %.3f ms" % t(lambda: torch.nn.functional.interpolate(x, scale_factor=2, mode="nearest")))
print("bilinear ac=True (torch): %.3f ms" % t(lambda: torch.nn.functional.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)))
print("relu 1.7 GB (read+write): %.3f ms" % t(lambda: torch.relu(y)))
