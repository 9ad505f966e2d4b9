"""Independent float64 restatement of the sepconv op (test infrastructure only).

Deliberately written a different way from sepconv_oracle.c -- sliding windows
and einsum in float64 instead of per-element loops in float32 -- so that an
indexing slip in one is not repeated in the other.  Formulas are the ones of
libs/sepconv/src/SeparableConvolution_kernel.cu:45-49 (forward), :99-109
(gradVertical) and :137-147 (gradHorizontal); gradInput is identically zero
(kernel.cu:152-206 never writes it).
"""
import numpy as np

K = 51


def _patches(inp, k=K):
    # [B,C,H,W,K(fy),K(fx)] view: patches[b,c,y,x,fy,fx] = inp[b,c,y+fy,x+fx]
    return np.lib.stride_tricks.sliding_window_view(inp, (k, k), axis=(2, 3))


def forward(inp, ver, hor):
    """Any filter length (ver.shape[1] == hor.shape[1]): the reference's cupy spelling takes it from the tensors
    (sff_scripts_interp/model/sepconv.py:15-30,85-90); the compiled op fixes 51 (kernel.cu:9)."""
    assert ver.shape[1] == hor.shape[1]
    p = _patches(inp.astype(np.float64), ver.shape[1])
    v = ver.astype(np.float64)
    h = hor.astype(np.float64)
    # out[b,c,y,x] = sum_{fy,fx} p[b,c,y,x,fy,fx] v[b,fy,y,x] h[b,fx,y,x]
    t = np.einsum("bcyxij,bjyx->bciyx", p, h, optimize=True)
    return np.einsum("bciyx,biyx->bcyx", t, v, optimize=True)


def backward(grad_out, inp, ver, hor):
    assert ver.shape[1] == hor.shape[1]
    p = _patches(inp.astype(np.float64), ver.shape[1])
    g = grad_out.astype(np.float64)
    v = ver.astype(np.float64)
    h = hor.astype(np.float64)
    gp = np.einsum("bcyx,bcyxij->byxij", g, p, optimize=True)  # sum over channels
    gv = np.einsum("byxij,bjyx->biyx", gp, h, optimize=True)
    gh = np.einsum("byxij,biyx->bjyx", gp, v, optimize=True)
    return np.zeros(inp.shape, np.float64), gv, gh
