"""CPU restatement of the reference's bilinear back-warp (test infrastructure only).

Follows ``sff_scripts_fusion/utils/image_warp_torch.py:35-112`` step by step in float32 numpy: NHWC
view, 1-pixel zero border (:37), x+1 / y+1 (:44-45), floor (:50-53), clamp of x0,x1,y0,y1 to the padded
range (:55-58), weights from the clamped x1,y1 (:87-93), four gathers and the stacked sum (:95).
Pinned by goldens generated from the reference module itself, which imports cleanly here (pure torch):
tests/golden/make_warp_golden.py.
"""
import numpy as np


def warp(img, flow):
    """img [B,C,H,W] float32, flow [B,2,H,W] float32 (channel 0 = dx, 1 = dy) -> [B,C,H,W] float32."""
    img = np.asarray(img, np.float32)
    flow = np.asarray(flow, np.float32)
    B, C, H, W = img.shape
    pad = np.zeros((B, H + 2, W + 2, C), np.float32)
    pad[:, 1:-1, 1:-1, :] = img.transpose(0, 2, 3, 1)
    cols = np.arange(W, dtype=np.float32)[None, None, :]
    rows = np.arange(H, dtype=np.float32)[None, :, None]
    x = (flow[:, 0] + cols).astype(np.float32) + np.float32(1)
    y = (flow[:, 1] + rows).astype(np.float32) + np.float32(1)
    x0 = np.floor(x).astype(np.int64); x1 = x0 + 1
    y0 = np.floor(y).astype(np.int64); y1 = y0 + 1
    x0 = np.clip(x0, 0, W + 1); x1 = np.clip(x1, 0, W + 1)
    y0 = np.clip(y0, 0, H + 1); y1 = np.clip(y1, 0, H + 1)
    dx = x1.astype(np.float32) - x
    dy = y1.astype(np.float32) - y
    one = np.float32(1)
    wa = dx * dy; wb = dx * (one - dy); wc = (one - dx) * dy; wd = (one - dx) * (one - dy)
    bi = np.arange(B)[:, None, None]
    Ia = pad[bi, y0, x0]; Ib = pad[bi, y1, x0]; Ic = pad[bi, y0, x1]; Id = pad[bi, y1, x1]
    out = ((wa[..., None] * Ia + wb[..., None] * Ib) + wc[..., None] * Ic) + wd[..., None] * Id
    return out.transpose(0, 3, 1, 2).astype(np.float32)
