"""CPU-side checks of the drop-in boundary: the shared library loads, exports every symbol the
public header declares, and refuses bad arguments without touching a GPU."""
import ctypes
import glob
import os
import re

import pytest
import torch

import libs.sepconv._ext.cunnex as cunnex
from libs.sepconv.SeparableConvolution import SeparableConvolution


def _declared_functions(repo_root):
    names = []
    for h in glob.glob(os.path.join(repo_root, "include", "*.h")):
        text = open(h).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names += re.findall(r"\b(sstem_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_library_exports_every_declared_symbol(repo_root):
    lib = ctypes.CDLL(cunnex.library_path())
    declared = _declared_functions(repo_root)
    assert "sstem_sepconv_forward_f32" in declared and "sstem_sepconv_backward_f32" in declared
    for name in declared:
        assert hasattr(lib, name), "libsstem_hip.so does not export %s" % name
    # and the Python binding table covers exactly the declared functions
    assert sorted(cunnex.C_ABI) == declared


def test_version_and_byte_model():
    lib = cunnex.load_library()
    assert lib.sstem_version() >= 100
    # 3,633,949,056 B per C2 forward call; 7,056.5 MB backward
    assert lib.sstem_sepconv_forward_bytes(8, 3, 1024, 1024) == 3633949056
    assert lib.sstem_sepconv_backward_bytes(8, 3, 1024, 1024) == 4 * (8 * 3 * 1024 * 1024 + 8 * 3 * 1074 * 1074 + 4 * 8 * 51 * 1024 * 1024)
    assert lib.sstem_status_string(0) == b"ok"


def test_argument_validation_without_gpu():
    lib = cunnex.load_library()
    # empty tensors: successful no-op, nothing is launched
    assert lib.sstem_sepconv_forward_f32(None, None, None, None, 0, 3, 4, 4, None) == 0
    assert lib.sstem_sepconv_backward_f32(None, None, None, None, None, None, None, 0, 3, 4, 4, None) == 0
    # null pointers / bad shapes / unsupported channel count are refused before any HIP call
    assert lib.sstem_sepconv_forward_f32(None, None, None, None, 1, 3, 4, 4, None) == 1
    assert lib.sstem_sepconv_forward_f32(None, None, None, None, -1, 3, 4, 4, None) == 2
    assert lib.sstem_sepconv_backward_f32(None, None, None, None, None, None, None, 1, 4, 4, 4, None) == 3
    assert b"three channels" in lib.sstem_last_error()


def test_operator_refuses_cpu_tensors_like_the_reference():
    # SeparableConvolution.py:47-48 of the reference: CPU -> NotImplementedError (no fallback)
    with pytest.raises(NotImplementedError):
        SeparableConvolution.apply(torch.zeros(1, 3, 52, 52), torch.zeros(1, 51, 2, 2), torch.zeros(1, 51, 2, 2))


def test_operator_shape_asserts():
    with pytest.raises(AssertionError):  # :29 input height mismatch
        SeparableConvolution.apply(torch.zeros(1, 3, 53, 52), torch.zeros(1, 51, 2, 2), torch.zeros(1, 51, 2, 2))
    with pytest.raises(AssertionError):  # :31 filter size must be 51
        SeparableConvolution.apply(torch.zeros(1, 3, 52, 52), torch.zeros(1, 50, 2, 2), torch.zeros(1, 51, 2, 2))
    with pytest.raises(AssertionError):  # :33 contiguity
        SeparableConvolution.apply(torch.zeros(1, 3, 52, 104)[..., ::2], torch.zeros(1, 51, 2, 2), torch.zeros(1, 51, 2, 2))


def test_binding_refuses_cpu_tensors():
    with pytest.raises(RuntimeError):
        cunnex.SeparableConvolution_cuda_forward(torch.zeros(1, 3, 52, 52), torch.zeros(1, 51, 2, 2),
                                                 torch.zeros(1, 51, 2, 2), torch.zeros(1, 3, 2, 2))
