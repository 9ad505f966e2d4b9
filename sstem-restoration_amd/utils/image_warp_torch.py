"""``SpatialTransformation`` -- the reference's bilinear back-warp module on MI355X.

Same class, constructor flag and call contract as ``sff_scripts_fusion/utils/image_warp_torch.py:5-112``:
``forward(moving_image[B,C,H,W], deformation_matrix[B,H,W,2]) -> warped[B,C,H,W]`` with
``deformation_matrix[..., 0] = dx`` (columns) and ``[..., 1] = dy`` (rows).  One native gather kernel
(``include/sstem_warp.h``) instead of ~20 torch ops; no gradient is defined, as in the way the reference
uses it (the flow network is frozen and runs under ``no_grad``, ``main_fusion.py:227-235``).
GPU tensors only -- there is no CPU fallback.
"""
import torch
import torch.nn as nn

import sstem_native


class SpatialTransformation(nn.Module):
    def __init__(self, use_gpu=False):
        self.use_gpu = use_gpu
        super(SpatialTransformation, self).__init__()

    @torch.no_grad()
    def forward(self, moving_image, deformation_matrix):
        if not moving_image.is_cuda or not deformation_matrix.is_cuda:
            raise NotImplementedError("the warp kernel is GPU-only")
        B, C, H, W = moving_image.shape
        assert tuple(deformation_matrix.shape) == (B, H, W, 2)
        img = moving_image.float().contiguous()
        # [B,H,W,2] -> [B,2,H,W]; free when the caller made it by permuting the flow network's output
        flow = deformation_matrix.permute(0, 3, 1, 2).float().contiguous()
        out = torch.empty_like(img)
        lib = sstem_native.load_library()
        with torch.cuda.device(img.device):
            rc = lib.sstem_warp_bilinear_f32(img.data_ptr(), flow.data_ptr(), out.data_ptr(), B, C, H, W,
                                             torch.cuda.current_stream().cuda_stream)
        sstem_native.check(rc, "sstem_warp_bilinear_f32")
        return out
