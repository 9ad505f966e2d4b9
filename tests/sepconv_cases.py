"""Seeded input generators shared by the CPU (oracle) and GPU (parity) tests."""
import numpy as np


def make_case(seed, B, C, H, W, kind="randn"):
    rng = np.random.default_rng(seed)
    inp = rng.random((B, C, H + 50, W + 50), dtype=np.float32)
    if kind == "randn":
        ver = rng.standard_normal((B, 51, H, W), dtype=np.float32)
        hor = rng.standard_normal((B, 51, H, W), dtype=np.float32)
    elif kind == "softmax":  # normalised positive kernels, like a kernel-prediction net's output
        def sm(a):
            e = np.exp(a - a.max(axis=1, keepdims=True))
            return (e / e.sum(axis=1, keepdims=True)).astype(np.float32)
        ver = sm(rng.standard_normal((B, 51, H, W), dtype=np.float32))
        hor = sm(rng.standard_normal((B, 51, H, W), dtype=np.float32))
    elif kind == "box":
        ver = np.full((B, 51, H, W), 1.0 / 51.0, np.float32)
        hor = np.full((B, 51, H, W), 1.0 / 51.0, np.float32)
    elif kind == "onehot":  # per-pixel random tap: the output is a gather of input pixels
        fy = rng.integers(0, 51, (B, H, W))
        fx = rng.integers(0, 51, (B, H, W))
        ver = np.zeros((B, 51, H, W), np.float32)
        hor = np.zeros((B, 51, H, W), np.float32)
        bb, yy, xx = np.meshgrid(np.arange(B), np.arange(H), np.arange(W), indexing="ij")
        ver[bb, fy, yy, xx] = 1.0
        hor[bb, fx, yy, xx] = 1.0
    else:
        raise ValueError(kind)
    grad = rng.standard_normal((B, C, H, W), dtype=np.float32)
    return inp, ver, hor, grad


def onehot_expected(inp, ver, hor):
    """Exact result for one-hot kernels: out[b,c,y,x] = in[b,c,y+fy*,x+fx*]."""
    B, C = inp.shape[:2]
    H, W = ver.shape[2:]
    fy = ver.argmax(axis=1)
    fx = hor.argmax(axis=1)
    bb, yy, xx = np.meshgrid(np.arange(B), np.arange(H), np.arange(W), indexing="ij")
    out = np.empty((B, C, H, W), np.float32)
    for c in range(C):
        out[:, c] = inp[bb, c, yy + fy, xx + fx]
    return out
