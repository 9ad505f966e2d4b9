"""Private instances of libsstem_hip.so for in-process A/B parity checks.

The library reads its developer knobs (SSTEM_GRAY_KERNEL, SSTEM_GRAY_SHAPE, SSTEM_GRAY_GV_SHAPE, ...) once, at the
first call that needs them.  A COPY of the .so under another file name is a separate dlopen with its own statics, so a
test can hold several differently configured instances at once and run them on the same device tensors through the
C-ABI (include/sstem_sepconv.h) -- e.g. the generic build (gray dispatch off) beside the product dispatch.
"""
import ctypes
import os
import shutil
import tempfile

import torch

import sstem_native

_p, _i64 = ctypes.c_void_p, ctypes.c_int64
_PROTOS = ("sstem_sepconv_forward_f32", "sstem_sepconv_backward_f32", "sstem_sepconv_interp_apply_f32",
           "sstem_last_error")
_keep = []   # temp dirs stay alive for the process


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


class Instance:
    def __init__(self, env):
        d = tempfile.TemporaryDirectory(prefix="sstem_inst_")
        _keep.append(d)
        path = os.path.join(d.name, "libsstem_inst_%d.so" % len(_keep))
        shutil.copy(sstem_native.library_path(), path)
        saved = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            self.lib = ctypes.CDLL(path)
            for name in _PROTOS:
                fn = getattr(self.lib, name)
                fn.restype, fn.argtypes = sstem_native.C_ABI[name]
            self._warm_up()      # every knob is read (and cached) while the environment is in place
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed (%d): %s" % (what, rc, self.lib.sstem_last_error().decode("utf-8", "replace")))

    def _warm_up(self):
        g = torch.rand(1, 1, 58, 58, device="cuda").expand(1, 3, 58, 58).contiguous()
        k = torch.rand(1, 51, 8, 8, device="cuda")
        self.forward(g, k, k)
        self.backward(torch.rand(1, 3, 8, 8, device="cuda"), g, k, k)
        u = torch.rand(1, 1, 8, 8, device="cuda").expand(1, 3, 8, 8).contiguous()
        self.interp_apply(u, u, k, k, k, k)
        torch.cuda.synchronize()

    def forward(self, inp, ver, hor):
        B, C, Hp, Wp = inp.shape
        H, W = ver.shape[2:]
        out = torch.empty(B, C, H, W, device=inp.device)
        s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        self._check(self.lib.sstem_sepconv_forward_f32(_ptr(inp), _ptr(ver), _ptr(hor), _ptr(out), B, C, H, W, s), "forward")
        return out

    def backward(self, grad, inp, ver, hor):
        B, C = inp.shape[:2]
        H, W = ver.shape[2:]
        gv, gh = torch.empty_like(ver), torch.empty_like(hor)
        s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        self._check(self.lib.sstem_sepconv_backward_f32(_ptr(grad), _ptr(inp), _ptr(ver), _ptr(hor), None, _ptr(gv),
                                                        _ptr(gh), B, C, H, W, s), "backward")
        return gv, gh

    def interp_apply(self, i1, i2, k1v, k1h, k2v, k2h):
        B, _, H, W = i1.shape
        out = torch.empty(B, 1, H, W, device=i1.device)
        s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        self._check(self.lib.sstem_sepconv_interp_apply_f32(_ptr(i1), _ptr(i2), _ptr(k1v), _ptr(k1h), _ptr(k2v), _ptr(k2h),
                                                            _ptr(out), B, H, W, s), "interp_apply")
        return out


_cache = {}


def instance(**env):
    """Instance configured by environment knobs, e.g. instance(SSTEM_GRAY_KERNEL="0"); cached per configuration."""
    key = tuple(sorted(env.items()))
    if key not in _cache:
        _cache[key] = Instance({k: str(v) for k, v in env.items()})
    return _cache[key]
