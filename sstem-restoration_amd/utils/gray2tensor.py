"""uint8 image <-> fp32 tensor edge on the GPU.

Same function names and results as the reference ``sp_scripts_test/utils/gray2tensor.py:7-31`` and the
conversions inlined in ``sff_scripts_interp/inference_singleImage.py:55-66,76``; the arithmetic runs in two
small native kernels (``include/sstem_io.h``) so only one uint8 plane crosses PCIe per image instead of
six fp32 channels, and the ``*255`` truncation (no clamp!) is reproduced bit for bit.
"""
import numpy as np
import torch

import sstem_native


def _stream():
    return torch.cuda.current_stream().cuda_stream


def gray_to_tensor(img_u8, replicas=1, device="cuda"):
    """uint8 [H,W] (numpy or tensor) -> float32 [1,replicas,H,W] = img/255 on the GPU (identical channels)."""
    if isinstance(img_u8, np.ndarray):
        if img_u8.dtype != np.uint8 or img_u8.ndim != 2:
            raise TypeError("expected a 2-D uint8 image, got %s %s" % (img_u8.dtype, img_u8.shape))
        img_u8 = torch.from_numpy(np.ascontiguousarray(img_u8))
    src = img_u8.to(device).contiguous()
    if not src.is_cuda:
        raise NotImplementedError("the conversion kernels are GPU-only")
    H, W = src.shape
    out = torch.empty((1, replicas, H, W), dtype=torch.float32, device=src.device)
    lib = sstem_native.load_library()
    with torch.cuda.device(src.device):
        rc = lib.sstem_gray_u8_to_f32(src.data_ptr(), out.data_ptr(), H * W, replicas, _stream())
    sstem_native.check(rc, "sstem_gray_u8_to_f32")
    return out


def tensor_to_gray(t, clamp01=False):
    """float32 GPU tensor [..., H, W] (first image/channel is used when 4-D) -> uint8 numpy [H,W] =
    (t*255).astype(uint8): truncation, NO clamp unless clamp01."""
    if t.dim() == 4:
        t = t[0, 0]
    if not t.is_cuda:
        raise NotImplementedError("the conversion kernels are GPU-only")
    t = t.detach().float().contiguous()
    out = torch.empty(t.shape, dtype=torch.uint8, device=t.device)
    lib = sstem_native.load_library()
    with torch.cuda.device(t.device):
        rc = lib.sstem_f32_to_gray_u8(t.data_ptr(), out.data_ptr(), t.numel(), 1 if clamp01 else 0, _stream())
    sstem_native.check(rc, "sstem_f32_to_gray_u8")
    return out.cpu().numpy()


# ---- the reference's names (gray2tensor.py) ----------------------------------------------------
def Gray2Tensor(im):
    return gray_to_tensor(np.asarray(im), 1)


def Tensor2Gray(tensor):
    return tensor_to_gray(tensor)


def TrainTensor2Gray(tensor):
    return tensor_to_gray(tensor)


def TrainTensor2mask(tensor):
    return tensor_to_gray(tensor, clamp01=True)
