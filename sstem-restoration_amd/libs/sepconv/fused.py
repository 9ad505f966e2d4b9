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
