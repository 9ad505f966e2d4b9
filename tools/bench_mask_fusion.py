import os, sys
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "sstem-restoration_amd"))
import torch, sstem_native
lib = sstem_native.load_library()
def timeit(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
a = torch.randn(4096, 4096, device="cuda")
import time; t0 = time.time()
while time.time() - t0 < 0.6: (a @ a).sum().item()
X6 = 5
for (N, Cin, H, W, Cout) in [(8, 51, 256, 256, 51), (8, 64, 128, 128, 64), (8, 128, 64, 64, 128), (16, 64, 128, 128, 64)]:
    x = torch.randn(N, Cin, H, W, device="cuda"); g = torch.randn(N, Cout, H, W, device="cuda")
    mask = torch.rand(N, Cout, H, W, device="cuda") > 0.5
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.1
    ws_n = lib.sstem_conv3x3_wgrad_workspace_floats_algo(N, Cin, H, W, Cout, X6); ws = torch.empty(ws_n, device="cuda")
    gw = torch.empty(Cout, Cin, 3, 3, device="cuda"); gb = torch.empty(Cout, device="cuda")
    t_plain = timeit(lambda: lib.sstem_conv2d_backward_weight_bias_f32(x.data_ptr(), g.data_ptr(), gw.data_ptr(), gb.data_ptr(), ws.data_ptr(), ws_n, N, Cin, H, W, Cout, 3, 3, 1, 1, None, X6))
    t_mask = timeit(lambda: lib.sstem_conv3x3_backward_weight_masked_f32(x.data_ptr(), g.data_ptr(), mask.data_ptr(), gw.data_ptr(), gb.data_ptr(), ws.data_ptr(), ws_n, N, Cin, H, W, Cout, 0, None, X6))
    t_where = timeit(lambda: torch.where(mask, g, torch.zeros((), device="cuda")))
    fn = lib.sstem_conv3x3_forward_workspace_floats_algo(N, Cout, H, W, Cin, X6); fws = torch.empty(fn, device="cuda"); gx = torch.empty(N, Cin, H, W, device="cuda")
    d_plain = timeit(lambda: lib.sstem_conv2d_forward_f32(g.data_ptr(), w.data_ptr(), None, None, None, gx.data_ptr(), fws.data_ptr(), fn, N, Cout, H, W, Cin, 3, 3, 1, 1, 1, 0, 0.0, None, X6))
    d_mask = timeit(lambda: lib.sstem_conv3x3_forward_masked_f32(g.data_ptr(), mask.data_ptr(), w.data_ptr(), None, None, None, gx.data_ptr(), None, fws.data_ptr(), fn, N, Cout, H, W, Cin, 1, 0, 0.0, None, X6))
    print("N%d %d->%d %dx%d: wgrad plain %.3f masked %.3f | dgrad plain %.3f masked %.3f | where pass %.3f ms" % (N, Cin, Cout, H, W, t_plain, t_mask, d_plain, d_mask, t_where))
