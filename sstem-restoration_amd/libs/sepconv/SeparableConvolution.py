"""``SeparableConvolution`` -- the reference's operator API on MI355X.

Mirrors ``libs/sepconv/SeparableConvolution.py:11-78`` of the reference:

* ``SeparableConvolution.apply(input[B,C,H+50,W+50], vertical[B,51,H,W],
  horizontal[B,51,H,W]) -> output[B,C,H,W]``
* the same shape/contiguity assertions (reference ``:29-35``),
* GPU tensors go to the native library, CPU tensors raise
  ``NotImplementedError`` (reference ``:47-48``) -- there is deliberately no CPU or
  PyTorch fallback in the product path,
* ``backward`` returns ``(grad_input, grad_vertical, grad_horizontal)`` where
  ``grad_input`` is all zeros: the reference's launcher never writes it
  (``src/SeparableConvolution_kernel.cu:152-206``).

Extension (round 5): ``vertical`` / ``horizontal`` may be bfloat16 tensors (BASELINE config 5, "bf16 activations with fp32 sepconv
accumulate"): frames, output and all sums stay float32.

Differences: the native kernels overwrite every output element, so the outputs
are allocated with ``empty`` instead of three extra zero-fill passes (``:37,60-62``);
``grad_input`` is still a zero tensor.  A non-GPU ``grad_output`` raises instead of
silently returning zero gradients (reference ``:64,76``).
"""
import torch

import libs.sepconv._ext as _ext  # noqa: F401  (same import shape as the reference, :7-8)
import libs.sepconv._ext.cunnex


class SeparableConvolution(torch.autograd.Function):
    FILTER = 51

    @staticmethod
    def forward(context, input, vertical, horizontal):
        context.save_for_backward(input, vertical, horizontal)

        batches, depth, in_h, in_w = input.shape
        taps = min(vertical.size(1), horizontal.size(1))
        out_h = min(vertical.size(2), horizontal.size(2))
        out_w = min(vertical.size(3), horizontal.size(3))

        assert in_h - 51 == out_h - 1
        assert in_w - 51 == out_w - 1
        assert taps == 51

        assert input.is_contiguous()
        assert vertical.is_contiguous()
        assert horizontal.is_contiguous()

        if not input.is_cuda:
            raise NotImplementedError()  # as the reference: no CPU version of the op

        output = input.new_empty((batches, depth, out_h, out_w))
        _ext.cunnex.SeparableConvolution_cuda_forward(input, vertical, horizontal, output)
        return output

    @staticmethod
    def backward(context, grad_output):
        _input, vertical, horizontal = context.saved_tensors

        if not grad_output.is_cuda:
            raise NotImplementedError()

        grad_output = grad_output.contiguous()
        grad_input = torch.zeros_like(_input)
        # bfloat16 coefficient tensors (BASELINE config 5; include/sstem_sepconv.h, ..._bf16coef): the kernels read them as they
        # are and write fp32 gradients; autograd wants a gradient of its input's dtype, so they are rounded on the way out
        grad_vertical = torch.empty_like(vertical, dtype=torch.float32)
        grad_horizontal = torch.empty_like(horizontal, dtype=torch.float32)

        _ext.cunnex.SeparableConvolution_cuda_backward(
            grad_output, _input, vertical, horizontal,
            grad_input, grad_vertical, grad_horizontal)

        if vertical.dtype != torch.float32:
            grad_vertical, grad_horizontal = grad_vertical.to(vertical.dtype), grad_horizontal.to(horizontal.dtype)
        return grad_input, grad_vertical, grad_horizontal
