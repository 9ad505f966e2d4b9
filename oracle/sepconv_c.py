"""ctypes binding of oracle/sepconv_oracle.c (test infrastructure only).

Each function mirrors one reference kernel; see the C file for file:line cites.
Arrays are numpy float32, contiguous NCHW; outputs are allocated zero-filled
here exactly as the reference's Python does (SeparableConvolution.py:37,60-62).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}

_i64 = ctypes.c_int64
_fp = ctypes.POINTER(ctypes.c_float)


def build():
    """Compile both oracle libraries (serial checker + OpenMP baseline)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])


def _lib(omp=False):
    name = "libsepconv_oracle_omp.so" if omp else "libsepconv_oracle.so"
    if name not in _LIBS:
        path = os.path.join(_HERE, name)
        if not os.path.exists(path):
            build()
        lib = ctypes.CDLL(path)
        lib.sepconv_oracle_forward.argtypes = [_fp] * 4 + [_i64] * 4
        lib.sepconv_oracle_forward.restype = ctypes.c_int
        lib.sepconv_oracle_grad_vertical.argtypes = [_fp] * 4 + [_i64] * 4
        lib.sepconv_oracle_grad_vertical.restype = ctypes.c_int
        lib.sepconv_oracle_grad_horizontal.argtypes = [_fp] * 4 + [_i64] * 4
        lib.sepconv_oracle_grad_horizontal.restype = ctypes.c_int
        lib.sepconv_oracle_backward.argtypes = [_fp] * 7 + [_i64] * 4
        lib.sepconv_oracle_backward.restype = ctypes.c_int
        lib.sepconv_oracle_num_threads.restype = ctypes.c_int
        _LIBS[name] = lib
    return _LIBS[name]


def _p(a):
    return a.ctypes.data_as(_fp)


def _check(inp, ver, hor):
    for a in (inp, ver, hor):
        assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"] and a.ndim == 4
    B, C, Hin, Win = inp.shape
    F = min(ver.shape[1], hor.shape[1])
    H = min(ver.shape[2], hor.shape[2])
    W = min(ver.shape[3], hor.shape[3])
    # SeparableConvolution.py:29-31
    assert Hin - 51 == H - 1
    assert Win - 51 == W - 1
    assert F == 51
    return B, C, H, W


def forward(inp, ver, hor, omp=False):
    B, C, H, W = _check(inp, ver, hor)
    out = np.zeros((B, C, H, W), np.float32)
    rc = _lib(omp).sepconv_oracle_forward(_p(inp), _p(ver), _p(hor), _p(out), B, C, H, W)
    if rc:
        raise RuntimeError("sepconv_oracle_forward rc=%d" % rc)
    return out


def backward(grad_out, inp, ver, hor, omp=False):
    """Returns (grad_input == zeros, grad_vertical, grad_horizontal)."""
    B, C, H, W = _check(inp, ver, hor)
    assert grad_out.dtype == np.float32 and grad_out.shape == (B, C, H, W)
    grad_out = np.ascontiguousarray(grad_out)
    gi = np.zeros_like(inp)
    gv = np.zeros_like(ver)
    gh = np.zeros_like(hor)
    rc = _lib(omp).sepconv_oracle_backward(_p(grad_out), _p(inp), _p(ver), _p(hor),
                                           _p(gi), _p(gv), _p(gh), B, C, H, W)
    if rc:
        raise RuntimeError("sepconv_oracle_backward rc=%d" % rc)
    return gi, gv, gh


def num_threads(omp=True):
    return int(_lib(omp).sepconv_oracle_num_threads())
