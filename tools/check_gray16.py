#!/usr/bin/env python
"""Developer check of the 16x16x4 fused-apply kernel (csrc/sepconv_kernels.hip, sepconv_gray16_mfma; SSTEM_GRAY16=0 selects the 4x4x1
kernel): small shapes against the CPU oracle (both coefficient layouts, which must agree bit for bit), then the C2 launch timed with
HIP events.  Run once per setting of SSTEM_GRAY16 (the knob is read once per process):
    SSTEM_GRAY16=1 python tools/check_gray16.py ; SSTEM_GRAY16=0 python tools/check_gray16.py"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "sstem-restoration_amd"), REPO]
from libs.sepconv.fused import coef_to_blocked, interp_apply_gray, interp_apply_gray_blocked   # noqa: E402
from oracle import sepconv_c                                                                    # noqa: E402  (checker)


def oracle(g1, g2, ks):
    pad = ((0, 0), (0, 0), (25, 25), (25, 25))
    r1 = np.repeat(g1, 3, 1); r2 = np.repeat(g2, 3, 1)
    y = sepconv_c.forward(np.pad(r2, pad, mode="edge"), ks[2], ks[3]) + sepconv_c.forward(np.pad(r1, pad, mode="edge"), ks[0], ks[1])
    return y.mean(axis=1, keepdims=True)


def main():
    print("SSTEM_GRAY16 =", os.environ.get("SSTEM_GRAY16", "(default)"))
    rng = np.random.default_rng(5)
    worst = 0.0
    for (B, H, W) in () if "--time-only" in sys.argv else ((1, 24, 64), (2, 40, 72), (1, 64, 64), (1, 100, 130), (2, 7, 200), (1, 131, 63)):
        g1 = rng.random((B, 1, H, W), dtype=np.float32); g2 = rng.random((B, 1, H, W), dtype=np.float32)
        ks = [rng.standard_normal((B, 51, H, W), dtype=np.float32) for _ in range(4)]
        ref = oracle(g1, g2, ks)
        t = [torch.from_numpy(a).cuda() for a in [g1, g2] + ks]
        a = interp_apply_gray(*t)
        bl = interp_apply_gray_blocked(t[0], t[1], *(coef_to_blocked(k) for k in t[2:]))
        torch.cuda.synchronize()
        err = float(np.abs(a.cpu().numpy() - ref).max() / np.abs(ref).max())
        same = bool(torch.equal(a, bl))
        worst = max(worst, err)
        print("B %d H %3d W %3d: rel err vs oracle %.2e, blocked == nchw: %s" % (B, H, W, err, same), flush=True)
        assert err < 2e-5 and same
    # one-hot taps: an exact gather
    B, H, W = 1, 32, 64
    if "--time-only" in sys.argv:
        return time_c2()
    g1 = rng.random((B, 1, H, W), dtype=np.float32); g2 = rng.random((B, 1, H, W), dtype=np.float32)
    ks = [np.zeros((B, 51, H, W), np.float32) for _ in range(4)]
    for k, tap in zip(ks, (3, 50, 25, 0)):
        k[:, tap] = 1.0
    ref = oracle(g1, g2, ks)
    t = [torch.from_numpy(a).cuda() for a in [g1, g2] + ks]
    got = interp_apply_gray(*t).cpu().numpy()
    print("one-hot taps: max |diff| %.2e (%s)" % (np.abs(got - ref).max(), "bit-exact" if np.array_equal(got, ref) else "not bit-exact"))
    time_c2()


def time_c2():
    B, S = 8, 1024
    g = torch.Generator(device="cuda"); g.manual_seed(555)
    p1 = torch.rand(B, 1, S, S, device="cuda", generator=g); p2 = torch.rand(B, 1, S, S, device="cuda", generator=g)
    kk = [coef_to_blocked(torch.softmax(torch.randn(B, 51, S, S, device="cuda", generator=g), 1)) for _ in range(4)]
    skew = int(os.environ.get("SSTEM_CHECK_SKEW_BYTES", "0"))          # experiment: the four coefficient tensors at staggered offsets
    if skew:
        kk2 = []
        for i, k in enumerate(kk):
            buf = torch.empty(k.numel() + 4 * skew, dtype=torch.float32, device="cuda")
            off = (i * skew) // 4
            v = buf[off:off + k.numel()].view(k.shape)
            v.copy_(k)
            kk2.append(v)
        kk = kk2
        print("coefficient tensors staggered by %d bytes" % skew)
    for _ in range(300):
        interp_apply_gray_blocked(p1, p2, *kk)
    torch.cuda.synchronize()
    n = 200
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); interp_apply_gray_blocked(p1, p2, *kk); b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    nbytes = 4 * (2 * B * S * S + 4 * B * 51 * S * S + B * S * S)
    print("C2 blocked: median %.4f ms, mean %.4f ms -> %.3f of 8 TB/s (mean)" % (ms[n // 2], sum(ms) / n, nbytes / (sum(ms) / n * 1e-3) / 8e12))


if __name__ == "__main__":
    main()
