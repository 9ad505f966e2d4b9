"""``FunctionSepconv`` / ``ModuleSepconv``: the reference's alternate (cupy) spelling of the op
(``sff_scripts_interp/model/sepconv.py:76-164``) on the native kernels.

As there, the filter length is whatever the coefficient tensors carry -- ``min(vertical.shape[1], horizontal.shape[1])`` (``:85``), with
the same shape and contiguity assertions (``:88-93``) and ``NotImplementedError`` for CPU tensors (``:111-112``).  51 taps go through
``SeparableConvolution`` (the MFMA kernels); any other length runs on the one-lane-per-element kernels through
``sstem_sepconv_forward_taps_f32`` (include/sstem_sepconv.h).  The reference's version is forward-only (its backward raises,
``:140-144``); this one has the gradient for every length."""
import torch

import sstem_native
from libs.sepconv.SeparableConvolution import SeparableConvolution


class _FunctionSepconvTaps(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, vertical, horizontal):
        taps = vertical.shape[1]
        B, C, Hin, Win = input.shape
        H, W = vertical.shape[2], vertical.shape[3]
        output = input.new_empty((B, C, H, W))
        lib = sstem_native.load_library()
        with torch.cuda.device(input.device):
            rc = lib.sstem_sepconv_forward_taps_f32(input.data_ptr(), vertical.data_ptr(), horizontal.data_ptr(), output.data_ptr(),
                                                    B, C, H, W, taps, torch.cuda.current_stream().cuda_stream)
        sstem_native.check(rc, "sstem_sepconv_forward_taps_f32")
        ctx.save_for_backward(input, vertical, horizontal)
        return output

    @staticmethod
    def backward(ctx, grad_output):
        input, vertical, horizontal = ctx.saved_tensors
        if not grad_output.is_cuda:
            raise NotImplementedError()
        grad_output = grad_output.contiguous()
        B, C, H, W = grad_output.shape
        gv, gh = torch.empty_like(vertical), torch.empty_like(horizontal)
        lib = sstem_native.load_library()
        with torch.cuda.device(input.device):
            rc = lib.sstem_sepconv_backward_taps_f32(grad_output.data_ptr(), input.data_ptr(), vertical.data_ptr(), horizontal.data_ptr(),
                                                     gv.data_ptr(), gh.data_ptr(), B, C, H, W, vertical.shape[1],
                                                     torch.cuda.current_stream().cuda_stream)
        sstem_native.check(rc, "sstem_sepconv_backward_taps_f32")
        return torch.zeros_like(input), gv, gh          # grad_input: zeros, as the compiled op of the reference leaves it (kernel.cu:152-206)


def FunctionSepconv(tenInput, tenVertical, tenHorizontal):
    taps = min(tenVertical.shape[1], tenHorizontal.shape[1])
    out_h = min(tenVertical.shape[2], tenHorizontal.shape[2])
    out_w = min(tenVertical.shape[3], tenHorizontal.shape[3])
    assert tenInput.shape[2] - taps == out_h - 1
    assert tenInput.shape[3] - taps == out_w - 1
    assert tenInput.is_contiguous()
    assert tenVertical.is_contiguous()
    assert tenHorizontal.is_contiguous()
    if taps == SeparableConvolution.FILTER and tenVertical.shape[1] == taps and tenHorizontal.shape[1] == taps:
        return SeparableConvolution.apply(tenInput, tenVertical, tenHorizontal)
    if not tenInput.is_cuda:
        raise NotImplementedError()                      # as the reference: no CPU version of the op
    if any(t.dtype != torch.float32 for t in (tenInput, tenVertical, tenHorizontal)):
        raise TypeError("FunctionSepconv: float32 tensors (bfloat16 coefficients: 51 taps only)")
    if tuple(tenVertical.shape) != tuple(tenHorizontal.shape):      # tensors of unequal extents: the common part, as the reference's min()
        tenVertical = tenVertical[:, :taps, :out_h, :out_w].contiguous()
        tenHorizontal = tenHorizontal[:, :taps, :out_h, :out_w].contiguous()
    return _FunctionSepconvTaps.apply(tenInput, tenVertical, tenHorizontal)


class ModuleSepconv(torch.nn.Module):
    def forward(self, tenInput, tenVertical, tenHorizontal):
        return FunctionSepconv(tenInput, tenVertical, tenHorizontal)
