"""``FunctionSepconv`` / ``ModuleSepconv``: the reference's alternate (cupy) spelling of the op
(``sff_scripts_interp/model/sepconv.py:152-164``) routed to the same native kernels.  The reference
version is forward-only (its backward raises, ``:140-144``); this one inherits the full backward."""
import torch

from libs.sepconv.SeparableConvolution import SeparableConvolution


def FunctionSepconv(tenInput, tenVertical, tenHorizontal):
    return SeparableConvolution.apply(tenInput, tenVertical, tenHorizontal)


class ModuleSepconv(torch.nn.Module):
    def forward(self, tenInput, tenVertical, tenHorizontal):
        return SeparableConvolution.apply(tenInput, tenVertical, tenHorizontal)
