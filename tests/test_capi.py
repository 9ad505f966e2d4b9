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


def test_conv_workspace_plans_without_gpu():
    """Host-side planning of the 3x3 convolution launches (pure functions of the sizes; nothing is launched)."""
    lib = cunnex.load_library()
    AUTO, DIRECT, MFMA, BF16 = 0, 1, 2, 3
    q = lib.sstem_conv3x3_forward_workspace_floats_algo
    # fp32 id: packed weights [cb][chunk of 8 ci][72][32 co]; the _algo query with AUTO / MFMA is the plain query
    assert lib.sstem_conv3x3_workspace_floats(64, 64) == 2 * 8 * 72 * 32
    for dims in ((8, 64, 512, 512, 64), (2, 512, 16, 16, 512), (1, 6, 9, 9, 51)):
        assert q(*dims, AUTO) == q(*dims, MFMA) == lib.sstem_conv3x3_forward_workspace_floats(*dims)
        assert q(*dims, DIRECT) == 0
    # bf16 id: packed weights [cb][chunk of 16 ci][9 taps][co block][16] bf16 = half a float each; 64-channel blocks above 32 outputs
    assert q(8, 64, 512, 512, 64, BF16) == 1 * 4 * 9 * 64 * 16 // 2              # large grid: no split-K slices
    assert q(8, 6, 512, 512, 32, BF16) == 1 * 1 * 9 * 32 * 16 // 2               # 6 channels padded to one chunk, 32-channel block
    assert q(8, 51, 512, 512, 51, BF16) == 1 * 4 * 9 * 64 * 16 // 2
    packed = 8 * 32 * 9 * 64 * 16 // 2
    deep = q(1, 512, 16, 16, 512, BF16)                                          # 8 workgroups per slice: split over K
    assert deep > packed and (deep - packed) % (1 * 512 * 16 * 16) == 0 and (deep - packed) // (512 * 16 * 16) in (2, 4, 8)
    # weight gradient: slabs [slices][9][CoutP][CinP] + one row of bias partial sums per slice (bf16) / four rows (fp32)
    w = lib.sstem_conv3x3_wgrad_workspace_floats_algo
    for dims in ((8, 64, 128, 128, 64), (16, 64, 256, 256, 64), (8, 6, 256, 256, 32), (1, 3, 1, 1, 2)):
        assert w(*dims, AUTO) == lib.sstem_conv3x3_wgrad_workspace_floats(*dims) and w(*dims, DIRECT) == 0
        n = w(*dims, BF16)
        cinp, coutp = (dims[1] + 63) // 64 * 64, (dims[4] + 63) // 64 * 64
        per_slice = 9 * coutp * cinp + coutp
        assert n > 0 and n % per_slice == 0
        slices = n // per_slice
        tiles = dims[0] * ((dims[3] + 31) // 32) * ((dims[2] + 1) // 2)
        assert 1 <= slices <= max(1, tiles // 8) and slices <= 256
    # empty and invalid sizes
    assert q(0, 64, 8, 8, 64, BF16) >= 0 and w(0, 64, 8, 8, 64, BF16) == 0 and q(1, -1, 8, 8, 64, BF16) == 0


def test_conv_algorithm_ranges_without_gpu():
    """sstem_conv3x3_algo_supported: the ranges the launchers enforce (include/sstem_conv.h), as pure functions of the sizes."""
    lib = cunnex.load_library()
    sup = lib.sstem_conv3x3_algo_supported
    MFMA, BF16, X3, X6, F16X3 = 2, 3, 4, 5, 6
    for algo in (MFMA, BF16, X3, X6, F16X3):
        assert sup(8, 64, 512, 512, 64, algo) == 1 and sup(8, 51, 1024, 1024, 51, algo) == 1 and sup(0, 64, 8, 8, 64, algo) == 0
    # 16-bit ids, W % 4 == 0: one channel plane below 2^31 bytes; fp16 pieces: EIGHT planes below 2^31 bytes (the staging loads' buffer
    # resource) unless the whole image is (dword staging)
    assert sup(1, 64, 8192, 8192, 64, X6) == 1 and sup(1, 64, 8192, 8192, 64, F16X3) == 0
    assert sup(1, 64, 8000, 8000, 64, F16X3) == 1 and sup(1, 4, 8192, 8192, 4, F16X3) == 1
    # W % 4 != 0: the whole image below 2^31 bytes
    assert sup(1, 64, 4097, 4097, 64, X6) == 0 and sup(1, 8, 4097, 4097, 8, X6) == 1 and sup(1, 8, 4097, 4097, 8, F16X3) == 1
    assert sup(1, 64, 8, 8, 64, 99) == 0


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


def test_every_environment_knob_is_documented(repo_root):
    """Every SSTEM_* variable the library, the Python package or bench.py reads is listed in INTEGRATION.md."""
    import glob
    import re
    names = set()
    files = glob.glob(os.path.join(repo_root, "sstem-restoration_amd", "**", "*.hip"), recursive=True)
    files += glob.glob(os.path.join(repo_root, "sstem-restoration_amd", "**", "*.py"), recursive=True)
    files.append(os.path.join(repo_root, "bench.py"))
    for f in files:
        text = open(f, errors="ignore").read()
        names |= set(re.findall(r'getenv\("(SSTEM_[A-Z0-9_]+)"\)', text))
        names |= set(re.findall(r'environ\.get\("(SSTEM_[A-Z0-9_]+)"', text))
    doc = open(os.path.join(repo_root, "INTEGRATION.md")).read()
    assert len(names) >= 20
    missing = sorted(n for n in names if n not in doc)
    assert not missing, "undocumented environment knobs: %s" % missing
