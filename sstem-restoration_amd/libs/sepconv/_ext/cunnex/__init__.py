"""ctypes binding of ``csrc/libsstem_hip.so`` under the reference's FFI module name.

The reference exposes two callables here, produced by ``torch.utils.ffi._wrap_function``
around the cffi module ``_cunnex`` (``libs/sepconv/_ext/cunnex/__init__.py:1-15``):

    SeparableConvolution_cuda_forward(input, vertical, horizontal, output)
    SeparableConvolution_cuda_backward(gradLoss, input, vertical, horizontal,
                                       gradInput, gradVertical, gradHorizontal)

Both take torch tensors, write their outputs in place on the current stream and
return 1.  The same two names with the same argument order live here; they unwrap
the tensors to raw device pointers + sizes and call the C-ABI
(``sstem_sepconv_forward_f32`` / ``sstem_sepconv_backward_f32``).

There is NO fallback: if the shared library is missing or fails to load, every
call raises (``ImportError``/``RuntimeError``); a CPU tensor is refused by the
caller exactly as in the reference (``SeparableConvolution.py:47-48``).
"""
import torch

__all__ = [
    "SeparableConvolution_cuda_forward",
    "SeparableConvolution_cuda_backward",
    "library_path",
    "load_library",
]

from sstem_native import C_ABI, library_path, load_library  # noqa: F401  (one loader for the whole C-ABI)

ALGO_AUTO, ALGO_DIRECT, ALGO_MFMA = 0, 1, 2
_forced_algo = ALGO_AUTO


def set_algorithm(algo):
    """Force a kernel family (tests / A-B benchmarks).  0 auto, 1 direct, 2 mfma."""
    global _forced_algo
    if algo not in (ALGO_AUTO, ALGO_DIRECT, ALGO_MFMA):
        raise ValueError("unknown sepconv algorithm id %r" % (algo,))
    _forced_algo = algo


def _raise_status(lib, rc, what):
    detail = lib.sstem_last_error().decode("utf-8", "replace")
    name = lib.sstem_status_string(rc).decode("utf-8", "replace")
    raise RuntimeError("%s failed: %s (%d)%s" % (what, name, rc, (": " + detail) if detail else ""))


def _dev_tensor(t, name, coef=False):
    """coef: a coefficient tensor (vertical / horizontal) -- float32, or bfloat16 for the ..._bf16coef entry points
    (include/sstem_sepconv.h: BASELINE config 5)."""
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise RuntimeError("%s must live on the GPU (got %s)" % (name, t.device))
    if t.dtype != torch.float32 and not (coef and t.dtype == torch.bfloat16):
        raise TypeError("%s must be float32%s (got %s)" % (name, " or bfloat16" if coef else "", t.dtype))
    if not t.is_contiguous():
        raise RuntimeError("%s must be contiguous" % name)
    return t


def _same_device(tensors):
    dev = tensors[0].device
    for t in tensors[1:]:
        if t.device != dev:
            raise RuntimeError("all tensors must be on the same device (%s vs %s)" % (dev, t.device))
    return dev


def SeparableConvolution_cuda_forward(input, vertical, horizontal, output):
    lib = load_library()
    ts = [_dev_tensor(input, "input"), _dev_tensor(vertical, "vertical", coef=True),
          _dev_tensor(horizontal, "horizontal", coef=True), _dev_tensor(output, "output")]
    if vertical.dtype != horizontal.dtype:
        raise TypeError("vertical and horizontal must have one dtype (%s vs %s)" % (vertical.dtype, horizontal.dtype))
    bf16 = vertical.dtype == torch.bfloat16
    dev = _same_device(ts)
    B, C, H, W = output.shape
    if tuple(input.shape) != (B, C, H + 50, W + 50) or tuple(vertical.shape) != (B, 51, H, W) \
            or tuple(horizontal.shape) != (B, 51, H, W):
        raise RuntimeError("sepconv forward: inconsistent shapes in=%s v=%s h=%s out=%s" % (
            tuple(input.shape), tuple(vertical.shape), tuple(horizontal.shape), tuple(output.shape)))
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream().cuda_stream
        if bf16:
            rc = lib.sstem_sepconv_forward_bf16coef(input.data_ptr(), vertical.data_ptr(), horizontal.data_ptr(), output.data_ptr(),
                                                    B, C, H, W, stream)
        else:
            rc = lib.sstem_sepconv_forward_f32_algo(
                input.data_ptr(), vertical.data_ptr(), horizontal.data_ptr(), output.data_ptr(),
                B, C, H, W, stream, _forced_algo)
    if rc != 0:
        _raise_status(lib, rc, "sstem_sepconv_forward_bf16coef" if bf16 else "sstem_sepconv_forward_f32")
    return 1


def SeparableConvolution_cuda_backward(gradLoss, input, vertical, horizontal,
                                       gradInput, gradVertical, gradHorizontal):
    lib = load_library()
    ts = [_dev_tensor(gradLoss, "gradLoss"), _dev_tensor(input, "input"),
          _dev_tensor(vertical, "vertical", coef=True), _dev_tensor(horizontal, "horizontal", coef=True),
          _dev_tensor(gradVertical, "gradVertical"), _dev_tensor(gradHorizontal, "gradHorizontal")]      # gradients: always float32
    if vertical.dtype != horizontal.dtype:
        raise TypeError("vertical and horizontal must have one dtype (%s vs %s)" % (vertical.dtype, horizontal.dtype))
    bf16 = vertical.dtype == torch.bfloat16
    if gradInput is not None:
        ts.append(_dev_tensor(gradInput, "gradInput"))
    dev = _same_device(ts)
    B, C, H, W = gradLoss.shape
    if tuple(input.shape) != (B, C, H + 50, W + 50) or tuple(vertical.shape) != (B, 51, H, W) \
            or tuple(horizontal.shape) != (B, 51, H, W) \
            or gradVertical.shape != vertical.shape or gradHorizontal.shape != horizontal.shape:
        raise RuntimeError("sepconv backward: inconsistent shapes")
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream().cuda_stream
        if bf16:
            rc = lib.sstem_sepconv_backward_bf16coef(
                gradLoss.data_ptr(), input.data_ptr(), vertical.data_ptr(), horizontal.data_ptr(),
                gradInput.data_ptr() if gradInput is not None else None,
                gradVertical.data_ptr(), gradHorizontal.data_ptr(), B, C, H, W, stream)
        else:
            rc = lib.sstem_sepconv_backward_f32_algo(
                gradLoss.data_ptr(), input.data_ptr(), vertical.data_ptr(), horizontal.data_ptr(),
                gradInput.data_ptr() if gradInput is not None else None,
                gradVertical.data_ptr(), gradHorizontal.data_ptr(),
                B, C, H, W, stream, _forced_algo)
    if rc != 0:
        _raise_status(lib, rc, "sstem_sepconv_backward_bf16coef" if bf16 else "sstem_sepconv_backward_f32")
    return 1
