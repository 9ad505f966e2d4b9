#!/usr/bin/env python
"""Phase times of the split weight-gradient kernel from a -DSSTEM_WGRAD_STAMPS=1 build (tools/build_wgrad_dev.sh stamps ...):
SSTEM_NATIVE_LIB=build_ablate/libsstem_wgrad_stamps.so python tools/wgrad_stamps.py N Cin H W Cout [f16x3|x6]"""
import ctypes, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "sstem-restoration_amd"))
import torch
import sstem_native
N, Cin, H, W, Cout = [int(v) for v in sys.argv[1:6]]
kind = sys.argv[6] if len(sys.argv) > 6 else "f16x3"
lib = sstem_native.load_library()
raw = ctypes.CDLL(os.environ["SSTEM_NATIVE_LIB"])
x = torch.randn(N, Cin, H, W, device="cuda"); g = torch.randn(N, Cout, H, W, device="cuda")
algo = {"x6": 5, "f16x3": 6}[kind]
ws_n = int(lib.sstem_conv3x3_wgrad_workspace_floats_algo(N, Cin, H, W, Cout, algo)); ws = torch.empty(max(ws_n, 1), device="cuda")
gw = torch.empty(Cout, Cin, 3, 3, device="cuda"); gb = torch.empty(Cout, device="cuda")
xw = torch.zeros(1024, device="cuda"); gword = torch.zeros(1024, device="cuda")
lib.sstem_amax_f32(x.data_ptr(), x.numel(), xw.data_ptr(), None); lib.sstem_amax_f32(g.data_ptr(), g.numel(), gword.data_ptr(), None)
out = (ctypes.c_ulonglong * 8)()
def run():
    if kind == "f16x3":
        rc = lib.sstem_conv3x3_backward_weight_scaled_masked_f32(x.data_ptr(), xw.data_ptr(), g.data_ptr(), gword.data_ptr(), None, gw.data_ptr(),
                                                                 gb.data_ptr(), ws.data_ptr(), ws_n, N, Cin, H, W, Cout, 0, None)
    else:
        rc = lib.sstem_conv2d_backward_weight_bias_f32(x.data_ptr(), g.data_ptr(), gw.data_ptr(), gb.data_ptr(), ws.data_ptr(), ws_n,
                                                       N, Cin, H, W, Cout, 3, 3, 1, 1, None, algo)
    sstem_native.check(rc, "wgrad")
for _ in range(3):
    run()
raw.sstem_debug_wgrad_stamps(out)
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    run()
e1.record(); torch.cuda.synchronize()
raw.sstem_debug_wgrad_stamps(out)
wgs = out[7]
names = ["first issue", "commit", "barrier 1", "issue next", "MFMA phase", "barrier 2", "slab stores"]
tot = sum(out[k] for k in range(7))
print("wgrad %s %s: %.1f us per call (kernel + reduce), %d workgroup-runs" % (kind, (N, Cin, H, W, Cout), e0.elapsed_time(e1) * 200, wgs))
for k, n in enumerate(names):
    print("  %-12s %9.0f clocks per workgroup  %5.1f %%" % (n, out[k] / wgs, 100.0 * out[k] / tot))
