"""``interp_apply`` -- the IFNet epilogue as one native launch (include/sstem_sepconv.h).

    out = mean_c( sepconv(ReplicationPad2d(25)(i2), k2v, k2h) + sepconv(ReplicationPad2d(25)(i1), k1v, k1h) )

i.e. ``sff_scripts_interp/model/model_interp.py:90-97`` (and each output channel of
``sp_scripts_train/networks.py:116-124``).  Forward only: the models use it when autograd is off and keep
the separate ``SeparableConvolution`` op (which has a backward) for training.  GPU tensors only.
"""
import torch

import sstem_native


def interp_apply(i1, i2, k1v, k1h, k2v, k2h):
    ts = [i1, i2, k1v, k1h, k2v, k2h]
    for t in ts:
        if not t.is_cuda:
            raise NotImplementedError("interp_apply is GPU-only")
        if t.dtype != torch.float32:
            raise TypeError("interp_apply needs float32 tensors")
    B, C, H, W = i1.shape
    if C != 3 or tuple(i2.shape) != (B, 3, H, W) or any(tuple(k.shape) != (B, 51, H, W) for k in ts[2:]):
        raise RuntimeError("interp_apply: inconsistent shapes")
    i1, i2, k1v, k1h, k2v, k2h = (t.contiguous() for t in ts)
    out = i1.new_empty((B, 1, H, W))
    lib = sstem_native.load_library()
    with torch.cuda.device(i1.device):
        rc = lib.sstem_sepconv_interp_apply_f32(i1.data_ptr(), i2.data_ptr(), k1v.data_ptr(), k1h.data_ptr(),
                                                k2v.data_ptr(), k2h.data_ptr(), out.data_ptr(), B, H, W,
                                                torch.cuda.current_stream().cuda_stream)
    sstem_native.check(rc, "sstem_sepconv_interp_apply_f32")
    return out


def interp_apply_gray_supported(B, H, W):
    return bool(sstem_native.load_library().sstem_sepconv_interp_apply_gray_supported(B, H, W))


def interp_apply_gray(g1, g2, k1v, k1h, k2v, k2h):
    """The same apply for callers that built the x3 channel replication themselves (every caller of the reference does:
    inference_singleImage.py:55-61, test_fusion.py:105-106): g1, g2 are the single planes [B,1,H,W].  Bit-identical to
    ``interp_apply`` on the replicated frames; one launch, no channel comparison (include/sstem_sepconv.h)."""
    ts = [g1, g2, k1v, k1h, k2v, k2h]
    for t in ts:
        if not t.is_cuda:
            raise NotImplementedError("interp_apply_gray is GPU-only")
        if t.dtype != torch.float32:
            raise TypeError("interp_apply_gray needs float32 tensors")
    B, C, H, W = g1.shape
    if C != 1 or tuple(g2.shape) != (B, 1, H, W) or any(tuple(k.shape) != (B, 51, H, W) for k in ts[2:]):
        raise RuntimeError("interp_apply_gray: inconsistent shapes")
    g1, g2, k1v, k1h, k2v, k2h = (t.contiguous() for t in ts)
    out = g1.new_empty((B, 1, H, W))
    lib = sstem_native.load_library()
    with torch.cuda.device(g1.device):
        rc = lib.sstem_sepconv_interp_apply_gray_f32(g1.data_ptr(), g2.data_ptr(), k1v.data_ptr(), k1h.data_ptr(),
                                                     k2v.data_ptr(), k2h.data_ptr(), out.data_ptr(), B, H, W,
                                                     torch.cuda.current_stream().cuda_stream)
    sstem_native.check(rc, "sstem_sepconv_interp_apply_gray_f32")
    return out


# ---- blocked coefficients (include/sstem_sepconv.h): [B, H, ceil(W/64), 51, 64] -----------------------------------------------

def coef_blocked_shape(B, H, W):
    return (B, H, (W + 63) // 64, 51, 64)


def interp_apply_gray_blocked_supported(B, H, W):
    return bool(sstem_native.load_library().sstem_sepconv_interp_apply_gray_blocked_supported(B, H, W))


def coef_to_blocked(coef):
    """NCHW coefficients [B,51,H,W] -> the row-segment layout the blocked apply reads (tests, foreign producers; the IFNet's
    kernel heads store that layout themselves)."""
    if not coef.is_cuda:
        raise NotImplementedError("coef_to_blocked is GPU-only")
    if coef.dtype != torch.float32 or coef.dim() != 4 or coef.shape[1] != 51:
        raise RuntimeError("coef_to_blocked needs a float32 [B,51,H,W] tensor")
    coef = coef.contiguous()
    B, _, H, W = coef.shape
    out = coef.new_empty(coef_blocked_shape(B, H, W))
    with torch.cuda.device(coef.device):
        rc = sstem_native.load_library().sstem_sepconv_coef_to_blocked_f32(coef.data_ptr(), out.data_ptr(), B, H, W,
                                                                           torch.cuda.current_stream().cuda_stream)
    sstem_native.check(rc, "sstem_sepconv_coef_to_blocked_f32")
    return out


def interp_apply_gray_blocked(g1, g2, k1v, k1h, k2v, k2h):
    """``interp_apply_gray`` on coefficient tensors in the blocked layout: bit-identical output, the coefficient streams walk
    consecutive addresses."""
    B, C, H, W = g1.shape
    ts = [g1, g2, k1v, k1h, k2v, k2h]
    for t in ts:
        if not t.is_cuda:
            raise NotImplementedError("interp_apply_gray_blocked is GPU-only")
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise TypeError("interp_apply_gray_blocked needs contiguous float32 tensors")
    if C != 1 or tuple(g2.shape) != (B, 1, H, W) or any(tuple(k.shape) != coef_blocked_shape(B, H, W) for k in ts[2:]):
        raise RuntimeError("interp_apply_gray_blocked: inconsistent shapes")
    out = g1.new_empty((B, 1, H, W))
    lib = sstem_native.load_library()
    with torch.cuda.device(g1.device):
        rc = lib.sstem_sepconv_interp_apply_gray_blocked_f32(g1.data_ptr(), g2.data_ptr(), k1v.data_ptr(), k1h.data_ptr(),
                                                             k2v.data_ptr(), k2h.data_ptr(), out.data_ptr(), B, H, W,
                                                             torch.cuda.current_stream().cuda_stream)
    sstem_native.check(rc, "sstem_sepconv_interp_apply_gray_blocked_f32")
    return out


# ---- bfloat16 coefficient tensors (include/sstem_sepconv.h, ..._bf16coef) ------------------------------------------------------

def interp_apply_gray_bf16coef_supported(B, H, W):
    return bool(sstem_native.load_library().sstem_sepconv_interp_apply_gray_bf16coef_supported(B, H, W))


def interp_apply_gray_bf16coef(g1, g2, k1v, k1h, k2v, k2h):
    """``interp_apply_gray`` on bfloat16 coefficient tensors [B,51,H,W] (the kernel heads' outputs handed over in bf16: half the
    coefficient bytes); planes, sums and the result float32 -- bit for bit what ``interp_apply_gray`` returns on ``k.float()``."""
    ts = [g1, g2, k1v, k1h, k2v, k2h]
    for t in ts:
        if not t.is_cuda:
            raise NotImplementedError("interp_apply_gray_bf16coef is GPU-only")
    if any(t.dtype != torch.float32 for t in ts[:2]) or any(t.dtype != torch.bfloat16 for t in ts[2:]):
        raise TypeError("interp_apply_gray_bf16coef needs float32 planes and bfloat16 coefficient tensors")
    B, C, H, W = g1.shape
    if C != 1 or tuple(g2.shape) != (B, 1, H, W) or any(tuple(k.shape) != (B, 51, H, W) for k in ts[2:]):
        raise RuntimeError("interp_apply_gray_bf16coef: inconsistent shapes")
    g1, g2, k1v, k1h, k2v, k2h = (t.contiguous() for t in ts)
    out = g1.new_empty((B, 1, H, W))
    lib = sstem_native.load_library()
    with torch.cuda.device(g1.device):
        rc = lib.sstem_sepconv_interp_apply_gray_bf16coef(g1.data_ptr(), g2.data_ptr(), k1v.data_ptr(), k1h.data_ptr(),
                                                          k2v.data_ptr(), k2h.data_ptr(), out.data_ptr(), B, H, W,
                                                          torch.cuda.current_stream().cuda_stream)
    sstem_native.check(rc, "sstem_sepconv_interp_apply_gray_bf16coef")
    return out


# ---- the uint8 image stored by the apply itself (include/sstem_sepconv.h, sstem_sepconv_interp_apply_gray_u8_f32) --------------

def interp_apply_gray_u8(g1, g2, k1v, k1h, k2v, k2h):
    """``interp_apply_gray`` / ``interp_apply_gray_blocked`` (by the coefficient tensors' shape) that ALSO returns
    ``(out * 255).astype(uint8)`` -- numpy's truncation, no clamp (inference_singleImage.py:76) -- stored by the same launch:
    (out float32 [B,1,H,W], image uint8 [B,H,W])."""
    ts = [g1, g2, k1v, k1h, k2v, k2h]
    for t in ts:
        if not t.is_cuda:
            raise NotImplementedError("interp_apply_gray_u8 is GPU-only")
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise TypeError("interp_apply_gray_u8 needs contiguous float32 tensors")
    B, C, H, W = g1.shape
    blocked = k1v.dim() == 5
    want = coef_blocked_shape(B, H, W) if blocked else (B, 51, H, W)
    if C != 1 or tuple(g2.shape) != (B, 1, H, W) or any(tuple(k.shape) != want for k in ts[2:]):
        raise RuntimeError("interp_apply_gray_u8: inconsistent shapes")
    out = g1.new_empty((B, 1, H, W))
    img = torch.empty((B, H, W), dtype=torch.uint8, device=g1.device)
    lib = sstem_native.load_library()
    with torch.cuda.device(g1.device):
        rc = lib.sstem_sepconv_interp_apply_gray_u8_f32(g1.data_ptr(), g2.data_ptr(), k1v.data_ptr(), k1h.data_ptr(), k2v.data_ptr(),
                                                        k2h.data_ptr(), out.data_ptr(), img.data_ptr(), B, H, W, 1 if blocked else 0,
                                                        torch.cuda.current_stream().cuda_stream)
    sstem_native.check(rc, "sstem_sepconv_interp_apply_gray_u8_f32")
    return out, img
