"""Host-side glue between torch.nn module trees and the gfx950 convolution kernels.

``functional``  autograd Functions over the C-ABI of include/sstem_conv.h (no torch/CPU fallback).
``fused``       ``FusedSequential``: an nn.Sequential with the reference's child indices (so
                state_dict keys are unchanged) whose forward runs Conv(+BN eval)(+activation) runs as
                single fused launches.
"""
from .functional import conv2d_fused, conv_transpose3x3s2_fused  # noqa: F401
from .fused import FusedSequential, invalidate_caches  # noqa: F401
