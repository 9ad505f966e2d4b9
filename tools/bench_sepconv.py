#!/usr/bin/env python
"""Developer micro-benchmark: forward and backward sepconv launches at one shape, HIP-event timed.
Prints ms per launch and achieved algorithmic GB/s.  (bench.py is the contract benchmark.)"""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "sstem-restoration_amd"))
import torch  # noqa: E402
import libs.sepconv._ext.cunnex as cunnex  # noqa: E402


def _prewarm(seconds=0.6):
    """An idle MI355X needs a few hundred ms under load to reach its clocks: the first shapes of a run measured 30-50 % slow."""
    import time
    a = torch.randn(4096, 4096, device="cuda")
    t0 = time.time()
    while time.time() - t0 < seconds:
        (a @ a).sum().item()


_prewarm()

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--size", type=int, default=1024)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--algo", type=int, default=0)
ap.add_argument("--what", default="fwd,bwd,fused")
ap.add_argument("--gray", action="store_true", help="grayscale frames replicated to 3 identical channels")
a = ap.parse_args()
cunnex.set_algorithm(a.algo)
lib = cunnex.load_library()
B, S = a.batch, a.size
torch.manual_seed(555)
inp = torch.rand(B, 1 if a.gray else 3, S + 50, S + 50, device="cuda").expand(B, 3, S + 50, S + 50).contiguous()
ver = torch.softmax(torch.randn(B, 51, S, S, device="cuda"), 1)
hor = torch.softmax(torch.randn(B, 51, S, S, device="cuda"), 1)
out = torch.empty(B, 3, S, S, device="cuda")
g = torch.randn(B, 3, S, S, device="cuda")
gv = torch.empty_like(ver)
gh = torch.empty_like(hor)


def timeit(fn, n):
    fn(); fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


tag = "%s tile=%s dbg=%s" % ("gray" if a.gray else "rgb",os.environ.get("SSTEM_TILE", "default"), os.environ.get("SSTEM_DEBUG_FLAGS", "0"))
if "fwd" in a.what:
    ms = timeit(lambda: cunnex.SeparableConvolution_cuda_forward(inp, ver, hor, out), a.iters)
    by = lib.sstem_sepconv_forward_bytes(B, 3, S, S)
    print("fwd  %s: %.4f ms  %.0f GB/s (%.1f%% of 8 TB/s)" % (tag, ms, by / ms / 1e6, by / ms / 1e6 / 80))
if "bwd" in a.what:
    ms = timeit(lambda: cunnex.SeparableConvolution_cuda_backward(g, inp, ver, hor, None, gv, gh), a.iters)
    by = lib.sstem_sepconv_backward_bytes(B, 3, S, S)
    print("bwd  %s: %.4f ms  %.0f GB/s (%.1f%% of 8 TB/s)" % (tag, ms, by / ms / 1e6, by / ms / 1e6 / 80))
if "fused" in a.what:
    from libs.sepconv.fused import interp_apply
    i1 = torch.rand(B, 1 if a.gray else 3, S, S, device="cuda").expand(B, 3, S, S).contiguous()
    i2 = torch.rand(B, 1 if a.gray else 3, S, S, device="cuda").expand(B, 3, S, S).contiguous()
    ms = timeit(lambda: interp_apply(i1, i2, ver, hor, hor, ver), a.iters)
    by = 4 * (2 * B * 3 * S * S + 4 * B * 51 * S * S + B * S * S)
    print("fused interp apply %s: %.4f ms  %.0f GB/s (%.1f%% of 8 TB/s)  [unfused: 2 fwd + add + mean]" % (tag, ms, by / ms / 1e6, by / ms / 1e6 / 80))
