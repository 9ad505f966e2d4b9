"""Autograd Functions over the native convolution entry points (include/sstem_conv.h).

``conv2d_fused(x, w, b, scale, shift, act, slope)``
    out = act((conv(x, w) + b) * scale + shift), stride 1, "same" padding -- what the reference spells
    as nn.Conv2d [+ nn.BatchNorm2d(eval)] [+ nn.ReLU | nn.LeakyReLU] (e.g. model_interp.py:121-127,
    networks.py:179-186, model_fusionnet.py:12-18).
``conv_transpose3x3s2_fused(...)``
    the same around nn.ConvTranspose2d(k=3, s=2, p=1, output_padding=1) (model_unet.py:32,70,
    model_fusionnet.py:21-27).

Backward (training mode, scale/shift absent): the activation mask is applied to grad_output, the data
gradient is the same MFMA kernel with transposed+flipped weights, the weight gradient is a native
reduction kernel, the bias gradient a sum.  CPU tensors raise NotImplementedError: there is no
fallback path.
"""
import ctypes
import os

import torch

import sstem_native

ACT_NONE, ACT_RELU, ACT_LEAKY = 0, 1, 2
ALGO_AUTO, ALGO_DIRECT, ALGO_MFMA, ALGO_MFMA_BF16, ALGO_MFMA_BF16X3, ALGO_MFMA_BF16X6, ALGO_MFMA_F16X3 = 0, 1, 2, 3, 4, 5, 6
_SPLIT_ALGOS = (ALGO_MFMA_BF16X3, ALGO_MFMA_BF16X6)     # fp32 operands split into 2 / 3 bf16 pieces (include/sstem_conv.h)
# ALGO_MFMA_F16X3: two fp16 pieces under per-tensor power-of-two scales (include/sstem_conv.h), launches nothing is recorded for only;
# as a forced id every other launch (recording, data / weight gradients) runs as under ALGO_MFMA_BF16X6
_ALL_ALGOS = (ALGO_AUTO, ALGO_DIRECT, ALGO_MFMA, ALGO_MFMA_BF16) + _SPLIT_ALGOS + (ALGO_MFMA_F16X3,)
_forced_algo = ALGO_AUTO


def set_algorithm(algo):
    """ALGO_MFMA_BF16 is the opt-in reduced-precision id (BASELINE config 5, "bf16 activations"): 3x3 forward and data
    gradient round both operands to bf16 while staging and sum in fp32; tensors, parameters, BatchNorm, sepconv and the
    optimiser stay fp32.  Everything that is not a 3x3 convolution runs as under ALGO_AUTO.
    ALGO_MFMA_BF16X6 / _BF16X3 are fp32 convolutions on the bf16 matrix cores: each fp32 operand is split into three / two bf16
    pieces whose exact products are summed in fp32 (six / three MFMAs per product term; csrc/conv_split_kernels.hip).  X6
    reproduces fp32 products to 2^-26 relative -- same tests and tolerances as ALGO_MFMA --, X3 to 1.1e-5.  Forward and data
    gradient; weight gradients, ConvTranspose and everything else run as under ALGO_AUTO."""
    global _forced_algo
    if algo not in _ALL_ALGOS:
        raise ValueError("unknown conv algorithm id %r" % (algo,))
    _forced_algo = algo


def get_algorithm():
    return _forced_algo


def _algorithm_from_env():
    """SSTEM_CONV_ALGO = auto | direct | mfma | bf16: the process-wide default (e.g. for the unmodified CLI); unknown values raise."""
    v = os.environ.get("SSTEM_CONV_ALGO")
    if not v:
        return
    names = {"auto": ALGO_AUTO, "direct": ALGO_DIRECT, "mfma": ALGO_MFMA, "bf16": ALGO_MFMA_BF16, "bf16x3": ALGO_MFMA_BF16X3,
             "bf16x6": ALGO_MFMA_BF16X6, "f16x3": ALGO_MFMA_F16X3}
    if v.lower() not in names:
        raise ValueError("SSTEM_CONV_ALGO=%r: expected one of %s" % (v, sorted(names)))
    set_algorithm(names[v.lower()])


_algorithm_from_env()


class algorithm(object):
    """``with algorithm(ALGO_MFMA_BF16): ...`` -- scoped set_algorithm."""

    def __init__(self, algo):
        self.algo = algo

    def __enter__(self):
        self.prev = _forced_algo
        set_algorithm(self.algo)

    def __exit__(self, *exc):
        set_algorithm(self.prev)
        return False


def _ptr(t):
    return t.data_ptr() if t is not None else None


# ---- host-side cost per launch (the eager 2-sample training step is host-bound: ~280 launches at ~18 us of Python each) ----------
class _NullCtx(object):
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NULL_CTX = _NullCtx()


def _on(device):
    """``with _on(t.device):`` -- torch.cuda.device(...) only when the tensor lives on another device than the current one (the
    context manager costs several microseconds of device switching per launch on the common single-device path)."""
    return _NULL_CTX if device.index == torch.cuda.current_device() else torch.cuda.device(device)


_size_queries = {}


def _q(name, *args):
    """A size query of the C-ABI (workspace floats, partial counts, support predicates: pure functions of their arguments),
    answered from a dictionary after the first call."""
    key = (name,) + args
    v = _size_queries.get(key)
    if v is None:
        v = _size_queries[key] = int(getattr(sstem_native.load_library(), name)(*args))
    return v


def _check(t, name):
    if not t.is_cuda:
        raise NotImplementedError("%s is a CPU tensor: the convolution blocks have no CPU path" % name)
    if t.dtype != torch.float32:
        raise TypeError("%s must be float32 (got %s)" % (name, t.dtype))
    c = t.contiguous()
    if c is not t:                       # a copy of a view (a channel slice of a concatenation's gradient): the bound travels with it
        hand_on_amax(t, c)
    return c


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """The current HIP stream of the current device as an integer handle.  torch.cuda.current_stream() builds a Stream object
    through three Python layers (9 us per call in the profile of the eager 2-sample step, once per launch); the raw getter is
    the same lookup without the object."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def conv3x3_bf16io(x, w, b=None, scale=None, shift=None, act=ACT_NONE, slope=0.0, out_bf16=False, owner=None, prepacked_ws=None,
                   in_mask=None, out_mask=None, out=None):
    """Inference-only spelling of the bf16-operand 3x3 convolution with bf16 ACTIVATION tensors: x may be float32 or bfloat16
    (NCHW, contiguous), the result is bfloat16 when out_bf16.  No autograd (FusedSequential uses it under no_grad for the
    convolutions inside one block); the caller has checked bf16io_ok."""
    lib = sstem_native.load_library()
    if not x.is_cuda:
        raise NotImplementedError("input is a CPU tensor: the convolution blocks have no CPU path")
    if x.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError("input must be float32 or bfloat16 (got %s)" % (x.dtype,))
    x = x.contiguous(); w = _check(w, "weight")
    b = _check(b, "bias") if b is not None else None
    scale = _check(scale, "scale") if scale is not None else None
    shift = _check(shift, "shift") if shift is not None else None
    N, Cin, H, W = x.shape
    Cout = w.shape[0]
    assert w.shape[1] == Cin and tuple(w.shape[2:]) == (3, 3)
    if out is None:
        out = torch.empty((N, Cout, H, W), dtype=torch.bfloat16 if out_bf16 else torch.float32, device=x.device)
    else:           # the caller's tensor (run_fused(out=...))
        assert tuple(out.shape) == (N, Cout, H, W) and out.dtype == (torch.bfloat16 if out_bf16 else torch.float32) \
            and out.device == x.device and out.is_contiguous()
    ws_n = _q("sstem_conv3x3_forward_workspace_floats_algo", N, Cin, H, W, Cout, ALGO_MFMA_BF16)
    prepacked = False
    if prepacked_ws is not None:                  # a workspace whose head already holds this call's packed weights
        ws, prepacked = prepacked_ws, True
        ws_n = ws.numel()
    elif owner is not None:
        ws, prepacked = _cached_workspace(owner, w, (False, ALGO_MFMA_BF16, N, Cin, H, W, Cout), ws_n, w)
    else:
        ws = w.new_empty((max(ws_n, 1),))
    with _on(x.device):
        if in_mask is not None or out_mask is not None:      # the ReLU mask written / applied by the launch (see _MASK_FUSION)
            rc = lib.sstem_conv3x3_forward_bf16io_masked(x.data_ptr(), 1 if x.dtype == torch.bfloat16 else 0, _ptr(in_mask), w.data_ptr(),
                                                         _ptr(b), _ptr(scale), _ptr(shift), out.data_ptr(), 1 if out_bf16 else 0,
                                                         _ptr(out_mask), ws.data_ptr(), ws_n, N, Cin, H, W, Cout,
                                                         2 if prepacked else 0, act, float(slope), _stream())
        else:
            rc = lib.sstem_conv3x3_forward_bf16io(x.data_ptr(), 1 if x.dtype == torch.bfloat16 else 0, w.data_ptr(), _ptr(b), _ptr(scale),
                                                  _ptr(shift), out.data_ptr(), 1 if out_bf16 else 0, ws.data_ptr(), ws_n, N, Cin, H, W, Cout,
                                                  2 if prepacked else 0, act, float(slope), _stream())
    sstem_native.check(rc, "sstem_conv3x3_forward_bf16io")
    return out


def bf16io_ok(x, conv, out_bf16):
    """Can conv3x3_bf16io take this call?  (bf16 id selected, no backward possible, 3x3, sizes the kernel supports)"""
    if _forced_algo != ALGO_MFMA_BF16 or torch.is_grad_enabled() or not _BF16_IO:
        return False
    if not x.is_cuda or x.dim() != 4 or tuple(conv.weight.shape[2:]) != (3, 3):
        return False
    N, Cin, H, W = x.shape
    return bool(_q("sstem_conv3x3_bf16io_supported", N, Cin, H, W, conv.weight.shape[0], 1 if out_bf16 else 0))


_BF16_IO = os.environ.get("SSTEM_BF16_IO", "1") != "0"        # developer knob (A/B runs): bf16 tensors between the convs of a block


# ---- amax words (include/sstem_conv.h): the per-tensor bounds the fp16 split id scales by -----------------------------------------
# A word is 1024 floats, zero when handed out; launches only ever raise slots.  Words come from pools of 256 zeroed by ONE fill
# launch; a pool allocated while a HIP graph is being captured belongs to that capture (its fill is a node of the graph, so every
# replay starts from zero again) -- pools are never shared between captures or between captured and eager launches.
# A tensor carries its word as ``t._sstem_amax = (word, t._version, capture generation)``: an in-place edit of the tensor makes the pair stale and the
# next consumer measures the tensor itself (one pass).  2 x 2 pooling and bilinear up-sampling hand the word on (convex combinations).
_AMAX_FLOATS = 1024
_AMAX_POOL_WORDS = 256
_amax_pools = {}           # (device, capture generation or None) -> [pool tensor, next index]
capture_generation = 0     # train_utils.GraphedCallable bumps it around every capture


def _new_amax_word(device):
    # per stream: the pool's zero fill is ordered with the launches that use its words only on the stream that allocated it
    key = (device, capture_generation if torch.cuda.is_current_stream_capturing() else None, _stream())
    ent = _amax_pools.get(key)
    if ent is None or ent[1] >= _AMAX_POOL_WORDS:
        for k in [k for k in _amax_pools if k[0] == device and k[1] is not None and k[1] != capture_generation]:
            del _amax_pools[k]                              # pools of finished captures: their words live on with the tensors that hold them
        ent = _amax_pools[key] = [torch.zeros((_AMAX_POOL_WORDS, _AMAX_FLOATS), dtype=torch.float32, device=device), 0]
    w = ent[0][ent[1]]
    ent[1] += 1
    return w


def _capture_gen():
    return capture_generation if torch.cuda.is_current_stream_capturing() else None


def tag_amax(t, word):
    t._sstem_amax = (word, t._version, _capture_gen())
    return t


def amax_word_of(t):
    """The amax word of a tensor if it still describes it (not edited in place since), else None.  While a HIP graph is being
    captured only words tagged under THIS capture count: a persistent input tagged by the eager warm-up runs carries a word from an
    eager pool that no node of the graph ever refreshes -- replays on refilled data would scale by the warm-up batch's bound
    (round-3 advisor finding); rejecting it makes the consumer record its own measuring pass into the graph."""
    tag = getattr(t, "_sstem_amax", None)
    if tag is None and t._base is not None:      # a view (slice, narrow, reshape) holds a subset of its base's elements: the base's bound holds
        tag = getattr(t._base, "_sstem_amax", None)          # (views share their base's version counter)
    if tag is not None and tag[1] == t._version and tag[0].device == t.device:
        if torch.cuda.is_current_stream_capturing() and tag[2] != capture_generation:
            return None
        return tag[0]
    return None


def tag_concat_amax(cat, *parts):
    """`cat` holds the channel-wise concatenation of `parts` (written in place by their producers): its bound is the slot-wise maximum
    of theirs -- when every part carries a valid word; otherwise the consumer measures."""
    words = [amax_word_of(p) for p in parts]
    if all(w is not None for w in words):
        w = words[0]
        for v in words[1:]:
            w = torch.maximum(w, v)
        tag_amax(cat, w)
    return cat


def tag_sum_amax(s, a, b, scale=1.0):
    """s = (a + b) * scale: |s| <= 2 |scale| max(bound of a, bound of b), slot by slot -- when both carry a valid word."""
    wa, wb = amax_word_of(a), amax_word_of(b)
    if wa is not None and wb is not None:
        tag_amax(s, torch.maximum(wa, wb) * (2.0 * abs(float(scale))))
    return s


def hand_on_amax(src, dst):
    """dst is made of convex combinations (or a selection) of src's elements: src's bound holds for it."""
    w = amax_word_of(src)
    if w is not None:
        tag_amax(dst, w)
    return dst


def measured_amax_word(x):
    """x's amax word; measured by one pass over x (sstem_amax_f32) when no producer has bounded it."""
    w = amax_word_of(x)
    if w is None:
        w = _new_amax_word(x.device)
        with _on(x.device):
            rc = sstem_native.load_library().sstem_amax_f32(x.data_ptr(), x.numel(), w.data_ptr(), _stream())
        sstem_native.check(rc, "sstem_amax_f32")
        tag_amax(x, w)
    return w


_PACK_CACHE_SLOTS = 4      # distinct (orientation, algorithm, sizes) workspaces kept per module
_touch_log = None          # train_utils.GraphedCallable, while it captures: the tensors every module-level cache that the body reads
                           # (packed weights here, folded BatchNorm in hipnn.fused) was built from


def _cached_workspace(owner, w, key, ws_n, like):
    """Workspace of a 3x3 MFMA launch whose weights cannot change before the next call with the same key (frozen or
    inference weights): kept on the owning module together with the weight's version counter and address, so that the next
    call finds its packed weights in place (SSTEM_CONV_WEIGHT_PREPACKED).  Returns (workspace, prepacked)."""
    if _touch_log is not None:
        _touch_log.append((w,))
    store = owner.__dict__.setdefault("_sstem_packs", {})
    sig = (w._version, w.data_ptr(), ws_n)
    ent = store.get(key)
    if ent is not None and ent[0] == sig and ent[1].device == like.device:
        return ent[1], True
    ws = like.new_empty((max(ws_n, 1),))
    if key not in store and len(store) >= _PACK_CACHE_SLOTS:
        store.pop(next(iter(store)))                      # oldest entry
    store[key] = (sig, ws)
    return ws, False


_bf16_fallback_logged = set()

# ALGO_AUTO for a 3x3 layer: the split-bf16 X6 id (fp32 operands as three bf16 pieces, six exact products per term summed in fp32: the
# fp32 ids' arithmetic at 6/16 of the fp32 MFMA's pipe time, measured at least as close to float64 as the fp32 MFMA kernel on every
# tested shape, and it passes that kernel's tests unchanged) where it is the faster kernel -- at least one 16-channel chunk of input
# channels, a map at least 12 pixels wide (16 x 16 tiles up to 16 columns; 8 x 8 maps stay on the fp32 kernel) and either at least 128 workgroups of its 8 x 32 x
# 64-channel tiles (256 until round 4: with this round's split kernels 128 wins on the 2-sample fusion step, 3.56 -> 3.44 ms replayed, and
# leaves the other steps where they were: gpurun r4bg / r4bh) or at least 128 input channels (a K loop long enough to split; below both the fp32 kernel's 4-row tiles win;
# tools/bench_conv.py --split, sets c2c3 / c5 / c3b2: 1.2-1.7x above the line, 0.7-1.0x below it).  Everything else stays on the fp32 MFMA kernel.  SSTEM_CONV_AUTO_SPLIT=0: AUTO
# never picks a split id (the round-1 behaviour).
_AUTO_SPLIT = os.environ.get("SSTEM_CONV_AUTO_SPLIT", "1") != "0"
_AUTO_SPLIT_MIN_WGS = int(os.environ.get("SSTEM_CONV_AUTO_SPLIT_MIN_WGS", "128"))
_AUTO_SPLIT_MIN_W = int(os.environ.get("SSTEM_CONV_AUTO_SPLIT_MIN_W", "12"))      # narrowest map the split kernels' 16 x 16 tiles are given
_AUTO_SPLIT_WGRAD_MIN_PIXELS = int(os.environ.get("SSTEM_CONV_AUTO_SPLIT_WGRAD_MIN_PIXELS", "0"))


def _auto_algo(N, Cin, H, W, Cout):
    """What ALGO_AUTO resolves to for a 3x3 layer of this size inside hipnn (the C-ABI's own AUTO stays the fp32 MFMA kernel)."""
    if N * ((Cout + 31) // 32) >= 65536:
        return ALGO_AUTO                          # the library decides (direct kernel)
    if _AUTO_SPLIT and Cin >= 16 and W >= _AUTO_SPLIT_MIN_W and (W > 16 or W % 4 == 0):
        # (maps of 12 / 16 columns: the split kernel's 16 x 16 tiles, 1.15-1.26x the fp32 kernel's on the 16 x 16 levels; 8 x 8 maps stay)
        tw, th = (32, 8) if W > 16 else (16, 16)
        wgs = ((W + tw - 1) // tw) * ((H + th - 1) // th) * N * ((Cout + 63) // 64 if Cout > 32 else 1)
        if W <= 16 and wgs < 64:
            return ALGO_MFMA                      # 2 x 512 channels at 16 x 16 is 16 workgroups: the fp32 kernel splits finer (fusion step at 2 samples 3.7 vs 4.2 ms)
        if (wgs >= _AUTO_SPLIT_MIN_WGS or Cin >= 128) and _q("sstem_conv3x3_algo_supported", N, Cin, H, W, Cout, ALGO_MFMA_BF16X6):
            return ALGO_MFMA_BF16X6
    return ALGO_MFMA


# ALGO_AUTO, launches nothing is recorded for (inference networks, the frozen flow net of the fusion step): the fp16 two-piece id where
# X6 would run -- half the matrix instructions, 2^-22 per product.  SSTEM_CONV_AUTO_F16X3=0 keeps X6 there.
_AUTO_F16 = os.environ.get("SSTEM_CONV_AUTO_F16X3", "1") != "0"


# ... and (round 5) RECORDED launches too: forward, data gradient and weight gradient of the layers X6 would serve run on the fp16
# two-piece id (sstem_conv3x3_forward_scaled_masked_f32 / sstem_conv3x3_backward_weight_scaled_masked_f32: half the matrix instructions,
# the ReLU masks inside the launches as under X6; gradients carry amax words like activations do).  SSTEM_CONV_AUTO_F16_TRAIN=0 keeps X6.
_AUTO_F16_TRAIN = os.environ.get("SSTEM_CONV_AUTO_F16_TRAIN", "1") != "0"


def _bounds_wanted():
    """Do recorded launches run on fp16 pieces at all (then producers of activations and gradients leave bounds behind)?"""
    return _AUTO_F16 and _AUTO_F16_TRAIN and _AUTO_SPLIT and _forced_algo in (ALGO_AUTO, ALGO_MFMA_F16X3)


_F16_WGRAD_NARROW = os.environ.get("SSTEM_CONV_F16_WGRAD_NARROW", "1") != "0"    # A/B knob: 0 keeps the narrow layers' weight gradients on the fp32 MFMA kernels
_F16_WGRAD_MIN_PIXELS = int(os.environ.get("SSTEM_CONV_F16_WGRAD_MIN_PIXELS", "4096"))     # below: only when both bounds are already there


def _train_f16(algo, N, Cin, H, W, Cout, both=True):
    """The id a recorded X6 launch of this size runs under: the fp16 two-piece id when the knob is on and its kernels take the layer
    (both = also the (Cout -> Cin) data gradient)."""
    if algo != ALGO_MFMA_BF16X6 or not (_AUTO_F16 and _AUTO_F16_TRAIN) or _forced_algo not in (ALGO_AUTO, ALGO_MFMA_F16X3):
        return algo
    if not _q("sstem_conv3x3_algo_supported", N, Cin, H, W, Cout, ALGO_MFMA_F16X3):
        return algo
    if both and not _q("sstem_conv3x3_algo_supported", N, Cout, H, W, Cin, ALGO_MFMA_F16X3):
        return algo
    return ALGO_MFMA_F16X3


_AUTO_F16_SMALL_CIN = os.environ.get("SSTEM_CONV_AUTO_F16_SMALL_CIN", "1") != "0"      # A/B knob: 0 keeps layers with < 16 input channels on the fp32 MFMA


def _inference_algo(N, Cin, H, W, Cout):
    """What ALGO_AUTO resolves to for a 3x3 launch nothing is recorded for: the fp16 two-piece id wherever X6 would run, and (round 4)
    for the full-resolution layers with FEWER than 16 input channels too (the first layers of every network: 6 -> 6, 6 -> 32 at
    1024^2, 1 -> 64 at 2048^2).  Padded to one 16-channel chunk they are one stream step per tile of the tile-walking instance: 8 x 6 ->
    32 at 1024^2 0.63 -> 0.38 ms, 8 x 6 -> 6 0.52 -> 0.30 ms (1 x 1 -> 64 at 2048^2 on the 64-channel instance 0.59 -> 0.32 ms), and
    their outputs carry a bound (no measuring pass behind them).  Recorded launches keep
    _auto_algo's answer (the fp32 MFMA kernel for these layers)."""
    algo = _auto_algo(N, Cin, H, W, Cout)
    if algo == ALGO_MFMA_BF16X6 and _AUTO_F16:
        return ALGO_MFMA_F16X3
    if algo == ALGO_MFMA and _AUTO_SPLIT and _AUTO_F16 and _AUTO_F16_SMALL_CIN and Cin < 16 and Cout <= 64 and W > 16 and W % 4 == 0 \
            and ((W + 31) // 32) * ((H + 7) // 8) * N >= 4096 and N * ((Cout + 31) // 32) < 65536:
        return ALGO_MFMA_F16X3
    return algo


# 3x3 layers with a handful of output channels (32 -> 2, 32 -> 1, 6 -> 6 at full resolution): the streaming fp32 kernel through the scaled
# entry (include/sstem_conv.h, SSTEM_CONV_DIRECT there) instead of a 32-channel output block that is 3-20 % occupied -- launches nothing
# is recorded for, under ALGO_AUTO, large enough to fill the chip.  SSTEM_CONV_AUTO_STREAM_SMALL=0 turns it off (A/B runs).
_AUTO_STREAM_SMALL = os.environ.get("SSTEM_CONV_AUTO_STREAM_SMALL", "1") != "0"


def _stream_small_ok(N, Cin, H, W, Cout):
    return _AUTO_STREAM_SMALL and Cout <= 8 and Cin * Cout <= 128 and N * H * W >= (1 << 20) \
        and bool(_q("sstem_conv3x3_stream_small_supported", N, Cin, H, W, Cout))


def _layer_algo(N, Cin, H, W, Cout, algo):
    """The algorithm id a 3x3 layer of this size runs under: a forced bf16 id falls back to the fp32 MFMA id for the layers its
    kernels cannot take (W % 4 != 0 with an image of 2 GiB or more; said once per shape) instead of failing the whole model."""
    if (algo == ALGO_MFMA_BF16 or algo in _SPLIT_ALGOS or algo == ALGO_MFMA_F16X3) and not _q("sstem_conv3x3_algo_supported", N, Cin, H, W, Cout, algo):
        key = (N, Cin, H, W, Cout)
        if key not in _bf16_fallback_logged:
            _bf16_fallback_logged.add(key)
            import warnings
            warnings.warn("hipnn: 3x3 layer N=%d %d->%d at %dx%d is outside the bf16 kernels' range; it runs under the fp32 MFMA id" % (N, Cin, Cout, H, W))
        return ALGO_MFMA
    return algo


def _resolved_algo(N, Cin, H, W, Cout, bn_part=None):
    """The algorithm id _raw_conv runs a 3x3 layer of this size under when it is given no prepacked workspace."""
    algo = _forced_algo
    if algo == ALGO_MFMA_F16X3:
        algo = ALGO_MFMA_BF16X6                   # recording launches and gradients of the forced fp16 id
    if algo == ALGO_AUTO:
        algo = ALGO_MFMA if bn_part is not None else _auto_algo(N, Cin, H, W, Cout)
    return _layer_algo(N, Cin, H, W, Cout, algo)


# ReLU bookkeeping of a training step inside the split-bf16 launches (sstem_conv3x3_forward_masked_f32 /
# sstem_conv3x3_backward_weight_masked_f32): the forward launch writes the (output > 0) mask itself and the data- and weight-gradient
# launches apply it while they stage the incoming gradient -- one compare and one select pass per Conv+ReLU layer and step less (9 % +
# 1.4 % of the fp32 IFNet training step).  SSTEM_MASK_FUSION=0 turns it off (A/B runs).
_MASK_FUSION = os.environ.get("SSTEM_MASK_FUSION", "1") != "0"


def _mask_fusable(algo, W):
    """Can a 3x3 launch under this id write / apply the ReLU mask itself?  (the split-bf16 ids; the bf16-operand id on its 16-byte
    staging path: W % 4 == 0 -- torch's allocations are 16-byte aligned)"""
    return algo in _SPLIT_ALGOS or (algo in (ALGO_MFMA_BF16, ALGO_MFMA_F16X3) and W % 4 == 0)


def blocked_store_ok(x, conv):
    """Can the launch of the 3x3 `conv` on x store its output in the row-segment layout [N, H, ceil(W/64), Cout, 64] the sepconv
    apply reads (include/sstem_conv.h, SSTEM_LAYOUT_ROW_SEGMENTS)?  Nothing recorded for a backward, a launch that runs under a
    split id through the scaled entry and is not split over K, one image of the result below 4 GiB."""
    if not (x.is_cuda and x.dim() == 4 and x.dtype == torch.float32) or _recording(x, conv.weight, conv.bias):
        return False
    if tuple(conv.weight.shape[2:]) != (3, 3):
        return False
    N, Cin, H, W = x.shape
    Cout = conv.weight.shape[0]
    algo = _forced_algo
    if algo == ALGO_AUTO:
        algo = _inference_algo(N, Cin, H, W, Cout)
    algo = _layer_algo(N, Cin, H, W, Cout, algo)
    scaled_entry = algo == ALGO_MFMA_F16X3 or (algo in _SPLIT_ALGOS and _AUTO_F16)
    if not scaled_entry or H * ((W + 63) // 64) * Cout * 256 >= (1 << 32):
        return False
    # ... and one the NCHW launch would not split over K either (a blocked store never is: the two spellings must give the same bits)
    return _q("sstem_conv3x3_forward_workspace_floats_algo", N, Cin, H, W, Cout, algo) == _q("sstem_conv3x3_packed_floats", Cin, Cout, algo)


LAYOUT_NCHW, LAYOUT_ROW_SEGMENTS, LAYOUT_CONVT_PARITY = 0, 1, 2        # include/sstem_conv.h, SSTEM_LAYOUT_*


def _channel_block_stride(out):
    """0 for a contiguous NCHW tensor, the image stride (floats) for a channel block of a larger contiguous NCHW tensor; anything
    else is a caller's error."""
    if out.is_contiguous():
        return 0
    N, C, H, W = out.shape
    st = out.stride()
    assert st[1] == H * W and st[2] == W and st[3] == 1 and st[0] >= C * H * W and out.data_ptr() % 16 == 0, \
        "out= must be contiguous or a channel block of a contiguous NCHW tensor"
    return st[0]


def is_channel_block(out):
    if out.dim() != 4 or out.dtype != torch.float32:
        return False
    N, C, H, W = out.shape
    st = out.stride()
    return st[1] == H * W and st[2] == W and st[3] == 1 and st[0] >= C * H * W and out.data_ptr() % 16 == 0


POOL_MAX, POOL_AVG = 1, 2          # include/sstem_conv.h, SSTEM_POOL_*


def pooled_store_ok(x, conv, pool):
    """Can the launch of the 3x3 `conv` on x also store the 2 x 2 pooling `pool` (an nn.MaxPool2d(2) / nn.AvgPool2d(2) module) of its
    output (sstem_conv3x3_forward_scaled_strided_f32, pooled_output)?  Returns the SSTEM_POOL_* kind or 0.  Nothing recorded, the
    fp16 id through the scaled entry, whole tiles (H % 8 == 0, W % 32 == 0), never split over K."""
    kind = _pool_kind(pool) if (_NATIVE_POOL and _POOL_FUSION) else None
    if kind is None or not isinstance(conv, torch.nn.Conv2d) or tuple(conv.weight.shape[2:]) != (3, 3):
        return 0
    if not (x.is_cuda and x.dim() == 4 and x.dtype == torch.float32) or _recording(x, conv.weight, conv.bias):
        return 0
    N, Cin, H, W = x.shape
    Cout = conv.weight.shape[0]
    if H % 8 or W % 32 or Cout * H * W * 4 >= (1 << 32):
        return 0
    algo = _forced_algo
    if algo == ALGO_AUTO:
        algo = _inference_algo(N, Cin, H, W, Cout)
    if _layer_algo(N, Cin, H, W, Cout, algo) != ALGO_MFMA_F16X3:
        return 0
    if _q("sstem_conv3x3_forward_workspace_floats_algo", N, Cin, H, W, Cout, algo) != _q("sstem_conv3x3_packed_floats", Cin, Cout, algo):
        return 0
    return POOL_MAX if kind == "max" else POOL_AVG


_POOL_FUSION = os.environ.get("SSTEM_POOL_FUSION", "1") != "0"       # A/B knob: 0 keeps every 2 x 2 pooling a launch of its own


def strided_store_ok(x, conv, out):
    """Can the launch of the 3x3 `conv` (or the sub-pixel form of a ConvTranspose2d(k3,s2,p1,op1)) on x store straight into `out`, a
    channel block of a larger NCHW tensor (sstem_conv3x3_forward_scaled_strided_f32)?  Nothing recorded, a launch through the scaled
    entry, never split over K."""
    if isinstance(conv, torch.nn.ConvTranspose2d):       # only the sub-pixel form takes an `out` at all
        return x.is_cuda and x.dim() == 4 and is_channel_block(out) and not _recording(x, conv.weight, conv.bias) \
            and _convT_subpixel_ok(x, conv.weight, conv, False, None)
    if out.is_contiguous():
        return True
    if not (x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and is_channel_block(out)) or _recording(x, conv.weight, conv.bias):
        return False
    if tuple(conv.weight.shape[2:]) != (3, 3):
        return False
    N, Cin, H, W = x.shape
    Cout = conv.weight.shape[0]
    algo = _forced_algo
    if algo == ALGO_AUTO:
        algo = _inference_algo(N, Cin, H, W, Cout)
    algo = _layer_algo(N, Cin, H, W, Cout, algo)
    if not (algo == ALGO_MFMA_F16X3 or (algo in _SPLIT_ALGOS and _AUTO_F16)):
        return False
    return _q("sstem_conv3x3_forward_workspace_floats_algo", N, Cin, H, W, Cout, algo) == _q("sstem_conv3x3_packed_floats", Cin, Cout, algo)


def _raw_conv(x, w, b, scale, shift, act, slope, transposed=False, owner=None, prepacked_ws=None, residual=None, res_scale=1.0,
              bn_part=None, in_mask=None, out_mask=None, out=None, inference=None, out_blocked=False, convt_parity=False, pool_out=None,
              pool_kind=0, pool_only=False):
    """One native launch; w is [Cout,Cin,KH,KW], or [Cin,Cout,3,3] when transposed.  owner: the module that owns w, given only
    when no backward can follow this call (then the packed weights are cached on it).  residual: out = (act(..) + residual) *
    res_scale in the store; bn_part: a [Cout, P, 3] tensor the launch fills with train-mode BatchNorm statistics partials
    (include/sstem_conv.h, sstem_conv2d_forward_ex_f32)."""
    lib = sstem_native.load_library()
    if inference is None:
        inference = owner is not None
    N, Cin, H, W = x.shape
    if transposed:
        assert w.shape[0] == Cin and w.shape[2:] == (3, 3)
        Cout, KH, KW = w.shape[1], 3, 3
    else:
        assert w.shape[1] == Cin, "weight/in-channel mismatch %s vs %s" % (tuple(w.shape), tuple(x.shape))
        Cout, KH, KW = w.shape[0], w.shape[2], w.shape[3]
    if convt_parity:      # the sub-pixel form of a ConvTranspose2d(k3, s2, p1, op1): w is [4 C, Cin, 3, 3] (convT_subpixel), the store shuffles
        assert not transposed and (KH, KW) == (3, 3) and Cout % 128 == 0 and prepacked_ws is None and bn_part is None
        assert in_mask is None and out_mask is None and not out_blocked
        out_stride = 0
        if out is None:
            out = x.new_empty((N, Cout // 4, 2 * H, 2 * W))
        else:
            assert tuple(out.shape) == (N, Cout // 4, 2 * H, 2 * W) and out.dtype == torch.float32 and out.device == x.device
            out_stride = _channel_block_stride(out)
        ws_n = _q("sstem_conv3x3_forward_workspace_floats_algo", N, Cin, H, W, Cout, ALGO_MFMA_F16X3)
        if owner is not None:
            ws, prepacked = _cached_workspace(owner, w, ("convT-subpixel", N, Cin, H, W, Cout), ws_n, x)
        else:
            ws, prepacked = x.new_empty((max(ws_n, 1),)), False
        in_word = measured_amax_word(x)
        out_word = _new_amax_word(x.device)
        with _on(x.device):
            rc = lib.sstem_conv3x3_forward_scaled_strided_f32(
                x.data_ptr(), in_word.data_ptr(), w.data_ptr(), _ptr(b), _ptr(scale), _ptr(shift), _ptr(residual), float(res_scale),
                out.data_ptr(), out_word.data_ptr(), ws.data_ptr(), ws_n, N, Cin, H, W, Cout, 2 if prepacked else 0,
                act, float(slope), _stream(), ALGO_MFMA_F16X3, LAYOUT_CONVT_PARITY, out_stride, None, 0)
        sstem_native.check(rc, "sstem_conv3x3_forward_scaled_strided_f32 (sub-pixel ConvTranspose)")
        return tag_amax(out, out_word)
    if out_blocked:       # the caller has asked blocked_store_ok
        assert out is None and residual is None and not transposed and (KH, KW) == (3, 3)
        out = x.new_empty((N, H, (W + 63) // 64, Cout, 64))
    elif pool_only:       # the pooled copy alone (the caller has asked pooled_store_ok): no full-resolution tensor at all
        assert out is None and pool_out is not None and residual is None
    elif out is None:
        out = x.new_empty((N, Cout, H, W))
    else:           # the caller's tensor: contiguous, or a channel block of a larger NCHW tensor (hipnn.fused.run_fused(out=...): a producer
        #             storing into the tensor its consumer concatenates; the callers have asked strided_store_ok)
        assert tuple(out.shape) == (N, Cout, H, W) and out.dtype == torch.float32 and out.device == x.device
    out_stride = _channel_block_stride(out) if not (out_blocked or pool_only) else 0
    algo = _forced_algo
    # the fp16 two-piece id serves launches nothing is recorded for (`inference`: the caller says so by naming the owner, or by asking
    # for it) that need none of the training extras; everything else of a forced fp16 id runs under X6
    f16_ok = inference and not transposed and prepacked_ws is None and bn_part is None and in_mask is None and out_mask is None
    if algo == ALGO_MFMA_F16X3 and not f16_ok:
        algo = ALGO_MFMA_BF16X6
    if algo in (ALGO_MFMA, ALGO_MFMA_BF16, ALGO_MFMA_F16X3) + _SPLIT_ALGOS and (KH, KW) != (3, 3):
        algo = ALGO_DIRECT
    if (KH, KW) == (3, 3):
        if algo == ALGO_AUTO and prepacked_ws is None:
            algo = ALGO_MFMA if bn_part is not None else (_inference_algo(N, Cin, H, W, Cout) if f16_ok else _auto_algo(N, Cin, H, W, Cout))
        algo = _layer_algo(N, Cin, H, W, Cout, algo)
    ws = None
    ws_n = 0
    prepacked = False
    # decided BEFORE the workspace lookup: the streaming kernel reads the plain weights and packs nothing, so it must not register a
    # workspace under this call's key either (a later launch of the same module that takes the packed path -- a residual, a strided or
    # pooled store, a forced id -- would find the entry, believe it packed and multiply uninitialised memory: round-4 advisor finding)
    stream_small = (KH, KW) == (3, 3) and f16_ok and _forced_algo == ALGO_AUTO and residual is None and not out_blocked \
        and out_stride == 0 and pool_out is None and not pool_only and in_mask is None and out_mask is None \
        and _stream_small_ok(N, Cin, H, W, Cout)
    if prepacked_ws is not None:                 # (algo, workspace) whose head already holds this call's packed weights
        algo, ws = prepacked_ws
        ws_n, prepacked = ws.numel(), True
    elif stream_small:
        pass
    elif (KH, KW) == (3, 3) and algo != ALGO_DIRECT:
        ws_n = _q("sstem_conv3x3_forward_workspace_floats_algo", N, Cin, H, W, Cout, algo)   # packed weights + split-K slices
        if owner is not None:
            ws, prepacked = _cached_workspace(owner, w, (bool(transposed), algo, N, Cin, H, W, Cout), ws_n, x)
        else:
            ws = x.new_empty((max(ws_n, 1),))
    if transposed and algo == ALGO_DIRECT:      # the direct kernel wants [Cout,Cin,3,3]
        w = w.transpose(0, 1).flip(2, 3).contiguous()
        transposed = False
    if (KH, KW) == (3, 3) and algo == ALGO_MFMA_F16X3 and (in_mask is not None or out_mask is not None or not inference):
        # a recorded launch on fp16 pieces (the pair workspace holds the packed weights behind their bound): masks inside, bound left behind
        assert residual is None and bn_part is None and out_stride == 0 and not out_blocked and pool_out is None
        assert (in_mask is None and out_mask is None) or _mask_fusable(algo, W)
        in_word = measured_amax_word(x)
        out_word = _new_amax_word(x.device)
        with _on(x.device):
            rc = lib.sstem_conv3x3_forward_scaled_masked_f32(
                x.data_ptr(), in_word.data_ptr(), _ptr(in_mask), w.data_ptr(), _ptr(b), _ptr(scale), _ptr(shift), out.data_ptr(),
                out_word.data_ptr(), _ptr(out_mask), _ptr(ws), ws_n, N, Cin, H, W, Cout, (1 if transposed else 0) | (2 if prepacked else 0),
                act, float(slope), _stream())
        sstem_native.check(rc, "sstem_conv3x3_forward_scaled_masked_f32")
        return tag_amax(out, out_word)
    if in_mask is not None or out_mask is not None:          # the callers have checked (_mask_fusable): no residual / statistics
        assert _mask_fusable(algo, W) and residual is None and bn_part is None
        with _on(x.device):
            if algo == ALGO_MFMA_BF16:
                rc = lib.sstem_conv3x3_forward_bf16io_masked(
                    x.data_ptr(), 0, _ptr(in_mask), w.data_ptr(), _ptr(b), _ptr(scale), _ptr(shift), out.data_ptr(), 0, _ptr(out_mask),
                    _ptr(ws), ws_n, N, Cin, H, W, Cout, (1 if transposed else 0) | (2 if prepacked else 0), act, float(slope), _stream())
            else:
                rc = lib.sstem_conv3x3_forward_masked_f32(
                    x.data_ptr(), _ptr(in_mask), w.data_ptr(), _ptr(b), _ptr(scale), _ptr(shift), out.data_ptr(), _ptr(out_mask),
                    _ptr(ws), ws_n, N, Cin, H, W, Cout, (1 if transposed else 0) | (2 if prepacked else 0), act, float(slope), _stream(), algo)
        sstem_native.check(rc, "sstem_conv3x3_forward_masked_f32")
        return out
    if stream_small:
        out_word = _new_amax_word(x.device)
        with _on(x.device):
            rc = lib.sstem_conv3x3_forward_scaled_strided_f32(
                x.data_ptr(), None, w.data_ptr(), _ptr(b), _ptr(scale), _ptr(shift), None, 1.0, out.data_ptr(), out_word.data_ptr(), None, 0,
                N, Cin, H, W, Cout, 0, act, float(slope), _stream(), ALGO_DIRECT, 0, 0, None, 0)
        sstem_native.check(rc, "sstem_conv3x3_forward_scaled_strided_f32 (streaming kernel)")
        return tag_amax(out, out_word)
    if (KH, KW) == (3, 3) and inference and (algo == ALGO_MFMA_F16X3 or (algo in _SPLIT_ALGOS and _AUTO_F16 and bn_part is None)):
        # the scaled entry: the fp16 id needs the input's bound; every split launch of an inference chain leaves its output's bound
        # behind for the next layer (the largest value it stores), so only tensors from other producers are ever measured
        in_word = measured_amax_word(x) if algo == ALGO_MFMA_F16X3 else None
        out_word = _new_amax_word(x.device)
        with _on(x.device):
            rc = lib.sstem_conv3x3_forward_scaled_strided_f32(
                x.data_ptr(), _ptr(in_word), w.data_ptr(), _ptr(b), _ptr(scale), _ptr(shift), _ptr(residual), float(res_scale),
                _ptr(out), out_word.data_ptr(), _ptr(ws), ws_n, N, Cin, H, W, Cout, (1 if transposed else 0) | (2 if prepacked else 0),
                act, float(slope), _stream(), algo, 1 if out_blocked else 0, out_stride, _ptr(pool_out), int(pool_kind) if pool_out is not None else 0)
        sstem_native.check(rc, "sstem_conv3x3_forward_scaled_strided_f32")
        if pool_out is not None:
            tag_amax(pool_out, out_word)             # a maximum / an average of four stored values: the output's bound holds
        if pool_only:
            return pool_out
        return tag_amax(out, out_word)
    assert pool_out is None and not pool_only, "a pooled copy needs a launch through the scaled entry (pooled_store_ok)"
    assert not out_blocked, "a blocked store needs a launch through the scaled entry (blocked_store_ok)"
    assert out_stride == 0, "a strided store needs a launch through the scaled entry (strided_store_ok)"
    with _on(x.device):
        rc = lib.sstem_conv2d_forward_ex_f32(
            x.data_ptr(), w.data_ptr(), _ptr(b), _ptr(scale), _ptr(shift), _ptr(residual), float(res_scale), out.data_ptr(), _ptr(bn_part),
            _ptr(ws), ws_n, N, Cin, H, W, Cout, KH, KW, KH // 2, KW // 2, (1 if transposed else 0) | (2 if prepacked else 0),
            act, float(slope), _stream(), algo)
    sstem_native.check(rc, "sstem_conv2d_forward_ex_f32")
    return out


def _act_mask(ctx, out, act, *inputs):
    """What backward needs of a fused activation: the sign pattern of the output, as a bool tensor.  The output
    itself is NOT saved: callers modify it in place (``x += x512``, model_interp.py:74), and ReLU / LeakyReLU(0.2)
    keep the sign, so ``out > 0`` taken now is the mask of the pre-activation.
    Only when a backward can follow (``ctx.recording``, decided by the public wrappers where the grad mode is visible:
    inside ``Function.forward`` it is always off, and ``ctx.needs_input_grad`` reports ``requires_grad`` of the Parameters
    whatever the mode).  A first version went by ``requires_grad`` and launched a compare kernel after every fused
    conv+ReLU of every inference: 43 per IFNet forward, 2.6 ms of 67 at C2."""
    if act == ACT_NONE or not ctx.recording:
        return None
    return out > 0


_zero_scalars = {}


def _zero_like_scalar(g):
    """A cached 0-dim zero on g's device (torch.zeros(()) is a fill launch of its own: 43 per IFNet step)."""
    key = (g.device, g.dtype)
    z = _zero_scalars.get(key)
    if z is None:
        z = _zero_scalars[key] = torch.zeros((), dtype=g.dtype, device=g.device)
    return z


_GRAD_SINK = os.environ.get("SSTEM_GRAD_SINK", "1") != "0"      # developer knob (A/B runs)


def _grad_sink(p, wanted):
    """The buffer a parameter's gradient is accumulated INTO by the native launch (accumulate flag of the *_ex entry points), or
    None.  Only for parameters whose .grad lives in a dataparallel.FlatGradBucket (it marks them ``_sstem_grad_sink``): their
    gradients are consumed through ``backward()`` + the bucket, never through ``torch.autograd.grad``, so adding in place here and
    returning None to autograd is the same arithmetic as autograd's own AccumulateGrad (one add launch per parameter and step)."""
    if not (_GRAD_SINK and wanted) or p is None or not getattr(p, "_sstem_grad_sink", False):
        return None
    g = p.grad
    if g is None or g.dtype != torch.float32 or g.device != p.device or not g.is_contiguous() or g.shape != p.shape:
        return None
    return g


def _sink_done(*params):
    """The launch that adds into these parameters' gradient sinks has been issued (possibly on the side stream): tell whoever
    listens (dataparallel.OverlappedBuckets counts deliveries per bucket to start its all-reduce early).  Autograd's own
    AccumulateGrad has post-accumulate hooks for this; a sink bypasses it."""
    for p in params:
        if p is not None:
            cb = p.__dict__.get("_sstem_grad_notify")
            if cb is not None:
                cb(p)


# ---- one reduce launch per backward pass ---------------------------------------------------------------------------------------
# Every 3x3 / ConvTranspose weight gradient is two launches: partial slabs, then their fixed-order sum into the gradient.  In a
# training step whose gradients go into sinks the second launches (17 of the 288 launches of the 2-sample fusion step, 13 us each) are
# recorded instead (accumulate == 3: include/sstem_conv.h, "Grouped weight-gradient reduce") and run as ONE launch from a callback the
# autograd engine runs at the end of the backward pass -- the same sums in the same order, bit for bit.  The slabs are kept alive
# until then; the deliveries into the gradient buckets are reported at that point too (a layer whose bucket starts its all-reduce
# inside the backward pass -- dataparallel.OverlappedBuckets -- is not deferred).
# OPT-IN (SSTEM_WGRAD_GROUP_REDUCE=1).  Measured (profiles/r05/e2_*, same box): the 2-sample fusion step goes from 288 to 272 launches and
# takes the same 3.46-3.47 ms -- the reduces are bound by the 200+ MB of slabs they read (210 us as one launch, 223 us as 17), not by
# their launches -- while the eager 8-sample IFNet step LOSES 8 % (10.55 -> 11.46 ms) and the SP joint step 3 %: their large weight
# gradients run on the side stream, reduce included, beside the data-gradient chain; deferred to the end of the pass the reduces are
# exposed.  With the knob on, layers whose weight gradient goes to the side stream keep their own reduce launch.
_WGRAD_GROUP = os.environ.get("SSTEM_WGRAD_GROUP_REDUCE", "0") == "1"
_deferred_wgrad = {"keep": [], "params": [], "queued": False}


def _defer_wgrad_ok(*params, flop=None):
    """May this weight gradient leave its reduce to the grouped launch?  (queues the flush callback on first use in a backward pass)"""
    if not _WGRAD_GROUP:
        return False
    if flop is not None and _SIDE_WGRAD and flop >= _SIDE_WGRAD_MIN_FLOP:
        return False                                   # a side-stream weight gradient: its reduce overlaps the data-gradient chain there
    for p in params:
        if p is not None and p.__dict__.get("_sstem_grad_notify") is not None:
            return False
    if not _deferred_wgrad["queued"]:
        try:
            torch.autograd.Variable._execution_engine.queue_callback(flush_deferred_wgrad)
        except RuntimeError:           # not inside a backward pass (a backward function called directly)
            return False
        _deferred_wgrad["queued"] = True
    return True


def _deferred_wgrad_issued(ws, *params):
    _deferred_wgrad["keep"].append(ws)
    _deferred_wgrad["params"].append(params)


def flush_deferred_wgrad():
    """The grouped reduce of every deferred weight gradient, on the current stream (behind the side stream's slab launches); also
    called by FlatGradBucket.allreduce_mean / FlatAdam.step / GraphedCallable as a safety net (a no-op when nothing is pending)."""
    d = _deferred_wgrad
    d["queued"] = False
    keep, d["keep"] = d["keep"], []
    params, d["params"] = d["params"], []
    if not keep:
        return
    join_side_streams()
    lib = sstem_native.load_library()
    dev = keep[0].device
    with _on(dev):
        main = torch.cuda.current_stream(dev)
        rc = lib.sstem_wgrad_deferred_flush(main.cuda_stream)
    for t in keep:
        t.record_stream(main)          # (a slab allocated under the side stream: its memory must not be reused before this launch ran)
    sstem_native.check(rc, "sstem_wgrad_deferred_flush")
    for ps in params:
        _sink_done(*ps)


def drop_deferred_wgrad():
    """Forget recorded reduce jobs (a backward pass that raised before its flush)."""
    d = _deferred_wgrad
    d["queued"] = False
    d["keep"], d["params"] = [], []
    sstem_native.load_library().sstem_wgrad_deferred_drop()


# ---- weight gradients beside the data-gradient chain -----------------------------------------------------------------------
# The backward of a layer is  BatchNorm/activation backward -> { data gradient -> the previous layer ...,  weight gradient }.
# Only the data gradient is on the critical path; the weight (+ bias) gradient is needed when the step's all-reduce / optimiser
# runs.  At the per-GPU batch of a data-parallel step (2 samples at 8 GPUs, BASELINE config 3) both are grids of 64-512
# workgroups on a 256-CU chip: launched on a second stream, the weight-gradient launches fill the CUs the data-gradient chain
# leaves idle.  Only when the result goes into a gradient sink (then nothing reads it before the backward pass ends); the main
# stream waits for the side stream in a callback the autograd engine runs at the end of the backward pass, so callers see the
# usual semantics.  SSTEM_SIDE_WGRAD=0 turns it off (A/B runs).
_SIDE_WGRAD = os.environ.get("SSTEM_SIDE_WGRAD", "1") != "0"
# Measured on the SFF fusion step (one box, eager / HIP-graph replay): batch 16: 23.75 -> 22.98 / 23.81 -> 23.60 ms; batch 2: 5.38 ->
# 5.58 (the per-layer stream switching costs the host ~20 us, and the eager batch-2 step is at the edge of being host-bound) /
# 5.34 -> 5.27 ms.  So only layers whose weight gradient is at least this much work go to the side stream:
_SIDE_WGRAD_MIN_FLOP = float(os.environ.get("SSTEM_SIDE_WGRAD_MIN_GFLOP", "8")) * 1e9
_side_streams = {}
_side_pending = set()


def _side_stream_for(device):
    s = _side_streams.get(device)
    if s is None:
        s = _side_streams[device] = torch.cuda.Stream(device=device)
    return s


def join_side_streams():
    """Make the current stream of every device with weight-gradient work on its side stream wait for that work."""
    for dev in list(_side_pending):
        torch.cuda.current_stream(dev).wait_stream(_side_streams[dev])
    _side_pending.clear()


class _on_side_stream:
    """with-block: launches inside go to the device's side stream, ordered after everything already on the current stream; the
    tensors they read are kept from being recycled until the side stream is done with them."""

    def __init__(self, enabled, *tensors, flop=None):
        self.enabled = enabled and _SIDE_WGRAD and (flop is None or flop >= _SIDE_WGRAD_MIN_FLOP)
        self.tensors = [t for t in tensors if t is not None]

    def __enter__(self):
        if not self.enabled:
            return self
        dev = self.tensors[0].device
        self.side = _side_stream_for(dev)
        self.side.wait_stream(torch.cuda.current_stream(dev))
        self.ctx = torch.cuda.stream(self.side)
        self.ctx.__enter__()
        # every side-stream launch queues the join (idempotent; a handful per backward pass): a pass that raised never ran its
        # callbacks, so "this device is already pending" says nothing about whether THIS pass has one queued.  FlatGradBucket.zero /
        # allreduce_mean and FlatAdam.step join explicitly as well.
        _side_pending.add(dev)
        try:
            torch.autograd.Variable._execution_engine.queue_callback(join_side_streams)
        except RuntimeError:           # not inside a backward pass (a backward function called directly): the explicit joins cover it
            pass
        return self

    def __exit__(self, *exc):
        if self.enabled:
            self.ctx.__exit__(*exc)
            for t in self.tensors:
                t.record_stream(self.side)
        return False


def _mask_grad(g, mask, act, slope):
    # one select kernel per activation (a cast of the mask + a multiply were two, 23 us per fused ReLU of an IFNet step)
    if act == ACT_RELU:
        return torch.where(mask, g, _zero_like_scalar(g))
    if act == ACT_LEAKY:
        return torch.where(mask, g, g * slope)
    return g


# On by default (SSTEM_PACK_PAIR=0 turns it off).  Same-box A/B on the IFNet training step with a properly warmed benchmark: 7.31 /
# 7.53 -> 7.26 / 7.25 ms (bf16 id), 19.58 / 19.55 -> 19.44 / 19.35 ms (fp32).  (A first A/B had shown 10-70 % outliers and kept it
# off: those were steps timed before the GPU had reached its clocks, not an effect of the pairing.)
_PACK_PAIR = os.environ.get("SSTEM_PACK_PAIR", "1") != "0"


def _pack_pair(x, w, bn_part=None):
    """Training, 3x3, an MFMA id: ONE launch packs the weights for the forward and (transposed + flipped) for the data gradient.
    Returns (algo, forward workspace, data-gradient workspace) or None when the pairing does not apply."""
    if not _PACK_PAIR:
        return None
    N, Cin, H, W = x.shape
    Cout = w.shape[0]
    algo = _forced_algo
    if algo == ALGO_AUTO and N * ((max(Cout, Cin) + 31) // 32) < 65536 and tuple(w.shape[2:]) == (3, 3):
        if bn_part is not None:
            algo = ALGO_MFMA                      # statistics partials come from the fp32 MFMA kernel's store
        else:
            algo = _auto_algo(N, Cin, H, W, Cout)     # what AUTO resolves to for the forward ...
            if _auto_algo(N, Cout, H, W, Cin) != algo:
                return None                       # ... and the data gradient would run under another id: each packs for itself
    if algo == ALGO_MFMA_F16X3:
        algo = ALGO_MFMA_BF16X6                   # a forced fp16 id: X6 where its recorded kernels do not take the layer
    if algo not in (ALGO_MFMA, ALGO_MFMA_BF16) + _SPLIT_ALGOS or tuple(w.shape[2:]) != (3, 3):
        return None
    if _layer_algo(N, Cin, H, W, Cout, algo) != algo or _layer_algo(N, Cout, H, W, Cin, algo) != algo:
        return None                               # a layer the bf16 id cannot take: packed per call under the fp32 id
    if bn_part is None:
        algo = _train_f16(algo, N, Cin, H, W, Cout)
    lib = sstem_native.load_library()
    n_f = _q("sstem_conv3x3_forward_workspace_floats_algo", N, Cin, H, W, Cout, algo)
    n_t = _q("sstem_conv3x3_forward_workspace_floats_algo", N, Cout, H, W, Cin, algo)
    if n_f <= 0 or n_t <= 0:
        return None
    slot = _pack_slot(w, (algo, N, Cin, H, W, Cout), n_f, n_t)
    if slot is not None:
        sig = (w._version, w.data_ptr())
        if slot.sig == sig and not _pack_always:
            return algo, slot.ws_f, slot.ws_t         # packed by the group launch after the optimiser step (or by an earlier call)
        ws_f, ws_t = slot.ws_f, slot.ws_t
        slot.sig = sig
    else:
        ws_f = w.new_empty((n_f,)); ws_t = w.new_empty((n_t,))      # fp32 like the weights (x may be a bf16 tensor inside a conv chain)
    with _on(x.device):
        rc = lib.sstem_conv3x3_pack_weights_f32(w.data_ptr(), Cin, Cout, algo, ws_f.data_ptr(), ws_t.data_ptr(), _stream())
    sstem_native.check(rc, "sstem_conv3x3_pack_weights_f32")
    return algo, ws_f, ws_t


# ---- one pack launch per optimiser step ----------------------------------------------------------------------------------
# A training step packs every 3x3 layer's weights once (forward + data-gradient orientation): one small launch per layer, 19 of the
# 278 launches of the 2-sample fusion step and 46 of the IFNet step.  The weights change in exactly one place -- the optimiser
# launch -- so the pair workspaces are kept on the Parameter (one slot per algorithm and activation shape, at most _PACK_SLOTS),
# and train_utils.FlatAdam.step re-packs ALL of them with one launch (sstem_conv3x3_pack_weights_group_f32) right after its
# update.  A slot is trusted only while the parameter's version counter and address are the ones it was packed from: any other
# writer (another optimiser, load_state_dict, .data assignment + increment_version) sends the layer back to its own pack launch.
# A captured graph (train_utils.GraphedCallable) contains no pack launches: the callable checks the slots' signatures before every
# replay (refresh_stale_pack_slots).
_PACK_GROUP = os.environ.get("SSTEM_PACK_GROUP", "1") != "0"      # developer knob (A/B runs)
_PACK_SLOTS = 2
_pack_always = False           # developer switch: every layer launches its own pack whatever its slot says (A/B runs, tests)
_group_tables = {}             # (device, algo) -> (signature, device table, total blocks)


_pin_slots = False             # train_utils.GraphedCallable: slots touched while it warms up and captures are never evicted
                               # (the captured launches read their workspaces on every replay)


class _PackSlot(object):
    __slots__ = ("key", "ws_f", "ws_t", "sig", "entry", "blocks", "pinned", "__weakref__")


def _pack_slot(w, key, n_f, n_t):
    if not _PACK_GROUP or not isinstance(w, torch.nn.Parameter):
        return None
    slots = w.__dict__.get("_sstem_pack_slots")
    if slots is None:
        slots = w.__dict__["_sstem_pack_slots"] = {}
    s = slots.get(key)
    if s is not None and s.ws_f.device == w.device:
        s.pinned = s.pinned or _pin_slots
        return s
    if key not in slots and sum(1 for v in slots.values() if not v.pinned) >= _PACK_SLOTS:
        slots.pop(next(k for k, v in slots.items() if not v.pinned))      # the oldest slot no captured graph reads
    s = _PackSlot()
    s.pinned = _pin_slots
    s.key = key
    s.ws_f = w.new_empty((n_f,)); s.ws_t = w.new_empty((n_t,))
    s.sig = None
    entry = (ctypes.c_int64 * 16)()
    s.blocks = int(sstem_native.load_library().sstem_conv3x3_pack_group_entry(key[2], key[5], key[0], entry))
    s.entry = list(entry)
    slots[key] = s
    return s


def refresh_stale_pack_slots(params):
    """Re-pack, one launch per layer, the pair workspaces of ``params`` whose weights have changed since they were packed (version
    counter or address) -- what a replayed graph that contains no pack launches needs checked before it runs.  Returns the count."""
    n = 0
    lib = None
    for p in params:
        slots = p.__dict__.get("_sstem_pack_slots")
        if not slots:
            continue
        sig = (p._version, p.data_ptr())
        for s in slots.values():
            if s.sig != sig and s.ws_f.device == p.device:
                lib = lib or sstem_native.load_library()
                algo, _, Cin, _, _, Cout = s.key
                with _on(p.device):
                    rc = lib.sstem_conv3x3_pack_weights_f32(p.data_ptr(), Cin, Cout, algo, s.ws_f.data_ptr(), s.ws_t.data_ptr(), _stream())
                sstem_native.check(rc, "sstem_conv3x3_pack_weights_f32")
                s.sig = sig
                n += 1
    return n


def repack_after_update(params):
    """ONE launch re-packs every pair workspace of ``params`` (the Parameters an optimiser launch has just written and whose version
    counters it has already bumped), on the current stream.  Returns the number of layers packed."""
    if not _PACK_GROUP:
        return 0
    groups = {}
    for p in params:
        slots = p.__dict__.get("_sstem_pack_slots")
        if not slots:
            continue
        for s in slots.values():
            if s.blocks > 0 and s.ws_f.device == p.device:
                groups.setdefault((p.device, s.key[0]), []).append((p, s))
    lib = None
    done = 0
    for (dev, algo), items in groups.items():
        sig = tuple((p.data_ptr(), s.ws_f.data_ptr(), s.ws_t.data_ptr()) for p, s in items)
        cached = _group_tables.get((dev, algo))
        if cached is None or cached[0] != sig:
            rows, first = [], 0
            for (p, s), ptrs in zip(items, sig):
                e = list(s.entry)
                e[0], e[1], e[2] = ptrs
                e[13], e[15] = first, 0
                if algo != ALGO_MFMA_F16X3:
                    e[14] = 0
                first += s.blocks
                rows.append(e)
            bounds, first_b = None, 0
            if algo == ALGO_MFMA_F16X3:          # every layer packs under its own bound: one float each, raised by the bound launch
                bounds = torch.zeros((len(rows),), dtype=torch.float32, device=dev)
                for i, e in enumerate(rows):
                    nb = e[14]
                    e[14], e[15] = first_b, bounds.data_ptr() + 4 * i
                    first_b += nb
            cached = _group_tables[(dev, algo)] = (sig, torch.tensor(rows, dtype=torch.int64, device=dev), first, bounds, first_b)
        lib = lib or sstem_native.load_library()
        with _on(dev):
            if algo == ALGO_MFMA_F16X3:
                rc = lib.sstem_conv3x3_pack_weights_group_f16(cached[1].data_ptr(), len(items), cached[2], cached[4], cached[3].data_ptr(), _stream())
            else:
                rc = lib.sstem_conv3x3_pack_weights_group_f32(cached[1].data_ptr(), len(items), cached[2], algo, _stream())
        sstem_native.check(rc, "sstem_conv3x3_pack_weights_group_f32")
        for p, s in items:
            s.sig = (p._version, p.data_ptr())
        done += len(items)
    return done


class _Conv2dFused(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, scale, shift, act, slope, recording=True, owner=None, residual=None, res_scale=1.0, bn_part=None, out=None,
                out_blocked=False, pool_out=None, pool_kind=0):
        ctx.recording = recording
        ctx.params = (w, b)                      # the Parameter objects themselves (their .grad may be a gradient sink)
        if residual is not None and recording:
            raise NotImplementedError("a fused residual has no backward (hipnn.fused only fuses it when nothing is recorded)")
        if out is not None and recording:
            raise NotImplementedError("out= is for launches nothing is recorded for (hipnn.fused only passes it then)")
        x = _check(x, "input"); w = _check(w, "weight")
        b = _check(b, "bias") if b is not None else None
        scale = _check(scale, "scale") if scale is not None else None
        shift = _check(shift, "shift") if shift is not None else None
        if w.shape[2] != w.shape[3] or w.shape[2] % 2 != 1:
            raise NotImplementedError("only odd square kernels with 'same' padding")
        ctx.dgrad_ws = None
        pair = _pack_pair(x, w, bn_part) if (recording and x.requires_grad) else None
        out_mask = None
        if recording and act == ACT_RELU and _MASK_FUSION and tuple(w.shape[2:]) == (3, 3) and bn_part is None:
            N, Cin, H, W = x.shape
            if _mask_fusable(pair[0] if pair is not None else _resolved_algo(N, Cin, H, W, w.shape[0]), W) and x.data_ptr() % 16 == 0:
                out_mask = torch.empty((N, w.shape[0], H, W), dtype=torch.bool, device=x.device)     # written by the launch
        if pair is not None:                     # a data gradient will follow: both packings now, in one launch
            out = _raw_conv(x, w, b, scale, shift, act, slope, prepacked_ws=(pair[0], pair[1]), bn_part=bn_part, out_mask=out_mask)
            ctx.dgrad_ws = (pair[0], pair[2])
        else:
            out = _raw_conv(x, w, b, scale, shift, act, slope, owner=None if recording else owner, residual=residual, res_scale=res_scale,
                            bn_part=bn_part, out_mask=out_mask, out=out, inference=not recording, out_blocked=out_blocked,
                            pool_out=pool_out, pool_kind=pool_kind)
        ctx.act, ctx.slope = act, slope
        ctx.has_bias = b is not None
        ctx.folded = scale is not None or shift is not None
        ctx.x_word = amax_word_of(x) if recording else None      # (measured by an fp16 launch above; the weight gradient scales by it)
        ctx.save_for_backward(x, w, out_mask if out_mask is not None else _act_mask(ctx, out, act, x, w, b))
        return out

    @staticmethod
    def backward(ctx, g):
        x, w, mask = ctx.saved_tensors
        if ctx.folded:
            raise NotImplementedError("backward through a folded (eval-mode) BatchNorm affine is not supported")
        g = _check(g, "grad_output")
        lib = sstem_native.load_library()
        N, Cin, H, W = x.shape
        Cout, _, KH, KW = w.shape
        want_gb = ctx.has_bias and ctx.needs_input_grad[2]
        # the ReLU mask inside the gradient launches: when every launch that reads g is a split-bf16 one
        fuse = mask is not None and ctx.act == ACT_RELU and _MASK_FUSION and (KH, KW) == (3, 3) and H * W * 256 < (1 << 32)
        if fuse and ctx.needs_input_grad[0]:
            fuse = _mask_fusable(ctx.dgrad_ws[0] if ctx.dgrad_ws is not None else _resolved_algo(N, Cout, H, W, Cin), W)
        wg_algo = _wgrad_algo(N, Cin, H, W, Cout) if (KH, KW) == (3, 3) else ALGO_DIRECT
        # the weight gradient on fp16 pieces: wherever X6 would run and -- measured after the kernel's round-5 tuning: 1.7x the fp32 MFMA
        # kernels on EVERY narrow layer too (6 -> 6, 32 -> 2, 32 -> 32 at 256^2: tools/bench_wgrad_split.py) -- wherever ALGO_AUTO would
        # have taken the fp32 kernels; a small layer whose operands would have to be measured first keeps its bound-free kernel
        wg_f16 = (KH, KW) == (3, 3) and ctx.needs_input_grad[1] and (wg_algo == ALGO_MFMA_BF16X6 or (wg_algo == ALGO_AUTO and _F16_WGRAD_NARROW)) \
            and _train_f16(ALGO_MFMA_BF16X6, N, Cin, H, W, Cout, both=False) == ALGO_MFMA_F16X3
        if wg_f16 and N * H * W < _F16_WGRAD_MIN_PIXELS and (ctx.x_word is None and amax_word_of(x) is None or amax_word_of(g) is None):
            wg_f16 = False
        if fuse and ctx.needs_input_grad[1]:
            fuse = wg_f16 or _mask_fusable(wg_algo, W)       # (the fp16 weight gradient applies the mask on both of its staging paths, as X6's)
        if fuse and (g.data_ptr() % 16 != 0 or x.data_ptr() % 16 != 0):
            fuse = False
        if fuse and want_gb and not ctx.needs_input_grad[1]:
            fuse = False                         # a bias gradient on its own is a torch reduction of the masked tensor
        if not fuse:
            g = _mask_grad(g, mask, ctx.act, ctx.slope)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            if (KH, KW) == (3, 3):
                gx = _raw_conv(g, w, None, None, None, ACT_NONE, 0.0, transposed=True, prepacked_ws=ctx.dgrad_ws,
                               in_mask=mask if fuse else None)
            else:   # generic odd kernel: correlate with the flipped, transposed weights
                gx = _raw_conv(g, w.transpose(0, 1).flip(2, 3).contiguous(), None, None, None, ACT_NONE, 0.0)
        if ctx.needs_input_grad[1]:
            algo = ALGO_MFMA_F16X3 if wg_f16 else wg_algo
            fused_gb = want_gb and (KH, KW) == (3, 3) and algo != ALGO_DIRECT     # the bias gradient rides along with the 3x3 MFMA weight gradient
            x_word = g_word = None
            if wg_f16:                               # bounds of both operands, measured on THIS stream where no producer left one
                x_word = ctx.x_word if (ctx.x_word is not None and amax_word_of(x) is None) else measured_amax_word(x)
                g_word = measured_amax_word(g)
            # gradient sinks: the launch adds into the parameters' .grad buffers (both or neither: one accumulate flag)
            sink_w = _grad_sink(ctx.params[0], True)
            sink_b = _grad_sink(ctx.params[1], fused_gb) if sink_w is not None else None
            if fused_gb and sink_b is None:
                sink_w = None
            gw = sink_w if sink_w is not None else torch.empty_like(w)
            if fused_gb:
                gb = sink_b if sink_b is not None else g.new_empty((Cout,))
            defer = sink_w is not None and (KH, KW) == (3, 3) and algo != ALGO_DIRECT and \
                _defer_wgrad_ok(ctx.params[0], ctx.params[1] if fused_gb else None, flop=2.0 * N * H * W * Cin * Cout * KH * KW)
            acc = 3 if defer else (1 if sink_w is not None else 0)
            with _on_side_stream(sink_w is not None, x, g, mask if fuse else None, x_word, g_word, flop=2.0 * N * H * W * Cin * Cout * KH * KW):
                ws, ws_n = None, 0
                if (KH, KW) == (3, 3) and algo != ALGO_DIRECT:
                    ws_n = _q("sstem_conv3x3_wgrad_workspace_floats_algo", N, Cin, H, W, Cout, algo)
                    ws = x.new_empty((max(ws_n, 1),))
                with _on(x.device):
                    if wg_f16:
                        rc = lib.sstem_conv3x3_backward_weight_scaled_masked_f32(
                            x.data_ptr(), x_word.data_ptr(), g.data_ptr(), g_word.data_ptr(), mask.data_ptr() if fuse else None, gw.data_ptr(),
                            _ptr(gb), _ptr(ws), ws_n, N, Cin, H, W, Cout, acc, _stream())
                    elif fuse and algo == ALGO_MFMA_BF16:
                        rc = lib.sstem_conv3x3_backward_weight_bf16_masked(x.data_ptr(), 0, g.data_ptr(), mask.data_ptr(), gw.data_ptr(), _ptr(gb),
                                                                           _ptr(ws), ws_n, N, Cin, H, W, Cout, acc, _stream())
                    elif fuse:
                        rc = lib.sstem_conv3x3_backward_weight_masked_f32(x.data_ptr(), g.data_ptr(), mask.data_ptr(), gw.data_ptr(), _ptr(gb),
                                                                          _ptr(ws), ws_n, N, Cin, H, W, Cout, acc, _stream(), algo)
                    else:
                        rc = lib.sstem_conv2d_backward_weight_bias_ex_f32(x.data_ptr(), g.data_ptr(), gw.data_ptr(), _ptr(gb), _ptr(ws), ws_n,
                                                                          N, Cin, H, W, Cout, KH, KW, KH // 2, KW // 2, acc, _stream(), algo)
                sstem_native.check(rc, "sstem_conv2d_backward_weight_bias_ex_f32")
            if sink_w is not None:
                gw = None
                if defer:
                    _deferred_wgrad_issued(ws, ctx.params[0], ctx.params[1] if fused_gb else None)
                else:
                    _sink_done(ctx.params[0], ctx.params[1] if fused_gb else None)
                if fused_gb:
                    gb = None
                    want_gb = False
        if want_gb and gb is None:
            gb = g.sum((0, 2, 3))
        return gx, gw, gb, None, None, None, None, None, None, None, None, None, None, None, None, None


_bf16_wgrad = True


def set_bf16_weight_gradient(on):
    """Under ALGO_MFMA_BF16: True (default) = the bf16-operand weight-gradient kernel; False = the fp32 MFMA kernel on the
    fp32 tensors (forward and data gradient stay bf16)."""
    global _bf16_wgrad
    _bf16_wgrad = bool(on)


def _wgrad_algo(N=0, Cin=0, H=0, W=0, Cout=0):
    """Algorithm id for the weight-gradient entry of a 3x3 layer.  Under ALGO_AUTO the split-bf16 X6 kernel (64 x 64 channel blocks)
    where it wins over the fp32 MFMA kernels (tools/bench_wgrad_split.py: 1.8-1.9x with full blocks -- also 51 -> 51, 63 % of a block --,
    1.15-1.2x with 32 x 64 channels (half a block), 0.8x at 32 x 32 and below, where the fp32 kernels have their narrow-side shapes)."""
    if _forced_algo == ALGO_MFMA_BF16 and not _bf16_wgrad:
        return ALGO_AUTO
    if _forced_algo == ALGO_MFMA_F16X3:
        return ALGO_MFMA_BF16X6                   # the recorded id of a forced fp16 id (the caller moves it to fp16 pieces where they apply)
    if _forced_algo == ALGO_AUTO and _AUTO_SPLIT and N * H * W >= _AUTO_SPLIT_WGRAD_MIN_PIXELS:
        cin_p, cout_p = (Cin + 63) // 64 * 64, (Cout + 63) // 64 * 64
        if 2 * Cin * Cout >= cin_p * cout_p:      # at least half of the 64 x 64 channel blocks is real channels
            return ALGO_MFMA_BF16X6
    return _forced_algo


def _zero_insert(x):
    """[N,C,H,W] -> [N,C,2H,2W] with x at the even positions (the stride-2 transposed convolution is a
    stride-1 3x3 convolution of this tensor with the transposed, flipped weights)."""
    N, C, H, W = x.shape
    z = x.new_zeros((N, C, 2 * H, 2 * W))
    z[:, :, ::2, ::2] = x
    return z


def _wgrad3x3(lib, x, g, Cout, want_bias=False):
    """Native 3x3 weight gradient [Cout,Cin,3,3] of a stride-1 'same' convolution (MFMA path); with want_bias also the
    bias gradient sum(g) over batch and pixels, from the same launches.  Returns (gw, gb or None)."""
    N, Cin, H, W = x.shape
    gw = x.new_empty((Cout, Cin, 3, 3))
    algo = _wgrad_algo(N, Cin, H, W, Cout)
    ws, ws_n, gb = None, 0, None
    if algo != ALGO_DIRECT:
        ws_n = _q("sstem_conv3x3_wgrad_workspace_floats_algo", N, Cin, H, W, Cout, algo)
        ws = x.new_empty((max(ws_n, 1),))
        if want_bias:
            gb = g.new_empty((Cout,))
    with _on(x.device):
        rc = lib.sstem_conv2d_backward_weight_bias_f32(x.data_ptr(), g.data_ptr(), gw.data_ptr(), _ptr(gb), _ptr(ws), ws_n,
                                                       N, Cin, H, W, Cout, 3, 3, 1, 1, _stream(), algo)
    sstem_native.check(rc, "sstem_conv2d_backward_weight_bias_f32")
    return gw, gb


# SSTEM_CONVT_ZERO_INSERT=1: the round-1 route of the fp32 ids (zero-inserted input + the 3x3 kernel at the output resolution), kept
# for A/B runs; the bf16 id still takes it (its ConvTranspose has no native kernel yet)
_CONVT_ZERO_INSERT = os.environ.get("SSTEM_CONVT_ZERO_INSERT", "0") == "1"


def _convT_route():
    if _forced_algo == ALGO_DIRECT:
        return "direct"
    if _forced_algo == ALGO_MFMA_BF16 or _CONVT_ZERO_INSERT:
        return "zero_insert"
    return "native"


_CT_TAP = {(0, 0): 1, (1, 0): 2, (1, 1): 0}       # (output parity, window offset) -> tap index of the transposed convolution


def convT_subpixel_weight(w):
    """[Cin, C, 3, 3] weights of nn.ConvTranspose2d(k3, s2, p1, op1) -> [4 C, Cin, 3, 3] weights of its sub-pixel form (include/
    sstem_conv.h, SSTEM_LAYOUT_CONVT_PARITY): output pixel (2y + py, 2x + px) = a 2 x 2 window over in[y .. y+1][x .. x+1], channel
    (2 py + px) C + co, the window in taps (1 + dy, 1 + dx) of a 3 x 3 kernel."""
    Cin, C = w.shape[:2]
    out = w.new_zeros((4, C, Cin, 3, 3))
    for py in (0, 1):
        for px in (0, 1):
            for dy in (0, 1):
                for dx in (0, 1):
                    if (py, dy) in _CT_TAP and (px, dx) in _CT_TAP:
                        out[2 * py + px, :, :, 1 + dy, 1 + dx] = w[:, :, _CT_TAP[(py, dy)], _CT_TAP[(px, dx)]].transpose(0, 1)
    return out.reshape(4 * C, Cin, 3, 3)


_CONVT_SUBPIXEL = os.environ.get("SSTEM_CONVT_SUBPIXEL", "1") != "0"        # A/B knob: 0 keeps every ConvTranspose on the fp32 MFMA kernels
_CONVT_SUBPIXEL_MIN_TILES = 256


def _convT_subpixel_ok(x, w, owner, recording, bn_part):
    """May this ConvTranspose launch run as its sub-pixel form on the fp16 two-piece id?  An inference launch of a module (packed
    weights cached on it) under AUTO with the fp16 id allowed (or that id forced), C % 32 == 0, Cin % 16 == 0, W % 4 == 0, enough
    tiles to fill the chip (the store is never split over K)."""
    if not _CONVT_SUBPIXEL or recording or owner is None or bn_part is not None:
        return False
    if _convT_route() != "native":       # the A/B routes (zero-inserted, direct) allocate their own result: no sub-pixel form, no out=
        return False
    if not (_forced_algo == ALGO_MFMA_F16X3 or (_forced_algo == ALGO_AUTO and _AUTO_SPLIT and _AUTO_F16)):
        return False
    N, Cin, H, W = x.shape
    C = w.shape[1]
    if C % 32 or Cin % 16 or W % 4 or W <= 16 or x.dtype != torch.float32 or not x.is_contiguous():      # (W <= 16: the split kernel's
        return False                                                                                    #  16 x 16 tiles have no such instance)
    if ((W + 31) // 32) * ((H + 7) // 8) * N * (4 * C // 64) < _CONVT_SUBPIXEL_MIN_TILES and _forced_algo != ALGO_MFMA_F16X3:
        return False
    return bool(_q("sstem_conv3x3_algo_supported", N, Cin, H, W, 4 * C, ALGO_MFMA_F16X3)) and C * 4 * H * W * 4 < (1 << 32)


def _rep4_cached(store, key, t):
    """t.repeat(4) kept beside the sub-pixel weights (per-channel parameters: the real channel's value for each of its four parities)."""
    if t is None:
        return None
    ent = store.get(key)
    if ent is None or ent[0] is not t or ent[1] != t._version:
        ent = store[key] = (t, t._version, t.detach().repeat(4).contiguous())
    return ent[2]


def _convT_subpixel(x, w, b, scale, shift, act, slope, owner, residual, res_scale, out=None):
    st = owner.__dict__.setdefault("_sstem_ct", {})
    ent = st.get("w")
    if ent is None or ent[0] != (w._version, w.data_ptr()) or ent[1].device != x.device:
        sub = torch.nn.Module()                       # owns the sub-pixel weights' packed image (hipnn's per-module pack cache)
        ent = st["w"] = ((w._version, w.data_ptr()), convT_subpixel_weight(w.detach()), sub)
    if _touch_log is not None:                        # a captured graph must notice a change of the ORIGINAL weights
        _touch_log.append((w,))
    return _raw_conv(x, ent[1], _rep4_cached(st, "b", b), _rep4_cached(st, "scale", scale), _rep4_cached(st, "shift", shift), act, slope,
                     owner=ent[2], residual=residual, res_scale=res_scale, inference=True, convt_parity=True, out=out)


class _ConvT3x3s2Fused(torch.autograd.Function):
    """ConvTranspose2d(k3,s2,p1,op1) [+affine][+act][+residual].  Default route: the native output-parity kernels
    (csrc/convt_kernels.hip: forward, data gradient and weight + bias gradient on the fp32 matrix cores, no zero-inserted
    tensor).  ALGO_DIRECT uses the gather kernels of the library instead (the cross-check); the bf16 id zero-inserts the input
    and runs its 3x3 kernel (4x redundant flops)."""

    @staticmethod
    def forward(ctx, x, w, b, scale, shift, act, slope, recording=True, owner=None, residual=None, res_scale=1.0, bn_part=None, out=None):
        ctx.recording = recording
        ctx.params = (w, b)
        if residual is not None and recording:
            raise NotImplementedError("a fused residual has no backward (hipnn.fused only fuses it when nothing is recorded)")
        if out is not None and (recording or not _convT_subpixel_ok(x, w, owner, recording, bn_part)):
            raise NotImplementedError("out= is for sub-pixel ConvTranspose launches nothing is recorded for (hipnn.fused asks strided_store_ok)")
        x = _check(x, "input"); w = _check(w, "weight")
        b = _check(b, "bias") if b is not None else None
        scale = _check(scale, "scale") if scale is not None else None
        shift = _check(shift, "shift") if shift is not None else None
        lib = sstem_native.load_library()
        N, Cin, H, W = x.shape
        assert w.shape[0] == Cin and tuple(w.shape[2:]) == (3, 3)
        Cout = w.shape[1]
        route = _convT_route()
        if route != "native" and (residual is not None or bn_part is not None):
            raise NotImplementedError("residual / bn_part need the native ConvTranspose kernel")
        if route == "direct":
            out = x.new_empty((N, Cout, 2 * H, 2 * W))
            with _on(x.device):
                rc = lib.sstem_conv_transpose3x3s2_forward_f32(x.data_ptr(), w.data_ptr(), _ptr(b), _ptr(scale), _ptr(shift),
                                                               out.data_ptr(), N, Cin, H, W, Cout, act, float(slope), _stream())
            sstem_native.check(rc, "sstem_conv_transpose3x3s2_forward_f32")
        elif route == "zero_insert":
            out = _raw_conv(_zero_insert(x), w, b, scale, shift, act, slope, transposed=True, owner=None if recording else owner)
        elif _convT_subpixel_ok(x, w, owner, recording, bn_part):
            out = _convT_subpixel(x, w, b, scale, shift, act, slope, owner, residual, res_scale, out)
        else:
            out = x.new_empty((N, Cout, 2 * H, 2 * W))
            ws_n = _q("sstem_conv_transpose3x3s2_workspace_floats", N, Cin, H, W, Cout, 0)
            prepacked = False
            if owner is not None and not recording:
                ws, prepacked = _cached_workspace(owner, w, ("convT", N, Cin, H, W, Cout), ws_n, x)
            else:
                ws = x.new_empty((max(ws_n, 1),))
            with _on(x.device):
                rc = lib.sstem_conv_transpose3x3s2_forward_ex_f32(x.data_ptr(), w.data_ptr(), _ptr(b), _ptr(scale), _ptr(shift), _ptr(residual),
                                                                  float(res_scale), out.data_ptr(), _ptr(bn_part), ws.data_ptr(), ws_n,
                                                                  N, Cin, H, W, Cout, 2 if prepacked else 0, act, float(slope), _stream())
            sstem_native.check(rc, "sstem_conv_transpose3x3s2_forward_ex_f32")
        ctx.act, ctx.slope = act, slope
        ctx.has_bias = b is not None
        ctx.folded = scale is not None or shift is not None
        ctx.route = route
        ctx.save_for_backward(x, w, _act_mask(ctx, out, act, x, w, b))
        return out

    @staticmethod
    def backward(ctx, g):
        x, w, mask = ctx.saved_tensors
        if ctx.folded:
            raise NotImplementedError("backward through a folded (eval-mode) BatchNorm affine is not supported")
        g = _mask_grad(_check(g, "grad_output"), mask, ctx.act, ctx.slope)
        lib = sstem_native.load_library()
        N, Cin, H, W = x.shape
        Cout = w.shape[1]
        gx = gw = gb = None
        want_gb = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.route == "direct":
            gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
            gw = torch.empty_like(w) if ctx.needs_input_grad[1] else None
            with _on(x.device):
                rc = lib.sstem_conv_transpose3x3s2_backward_f32(x.data_ptr(), w.data_ptr(), g.data_ptr(), _ptr(gx), _ptr(gw),
                                                                N, Cin, H, W, Cout, _stream())
            sstem_native.check(rc, "sstem_conv_transpose3x3s2_backward_f32")
        elif ctx.route == "zero_insert":
            if ctx.needs_input_grad[0]:
                # grad_in[y,x] = conv3x3(g, W as [out=Cin][in=Cout])[2y,2x]
                gx = _raw_conv(g, w, None, None, None, ACT_NONE, 0.0)[:, :, ::2, ::2].contiguous()
            if ctx.needs_input_grad[1]:
                # grad_W[ci,co,ky,kx] = wgrad3x3(zero_insert(x), g)[co,ci,2-ky,2-kx]
                gw, gb = _wgrad3x3(lib, _zero_insert(x), g, Cout, want_gb)      # the bias gradient rides along
                gw = gw.transpose(0, 1).flip(2, 3).contiguous()
        else:
            want_gw = ctx.needs_input_grad[1]
            gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
            fused_gb = want_gb and want_gw
            sink_w = _grad_sink(ctx.params[0], want_gw)
            sink_b = _grad_sink(ctx.params[1], fused_gb) if sink_w is not None else None
            if fused_gb and sink_b is None:
                sink_w = None
            if want_gw:
                gw = sink_w if sink_w is not None else torch.empty_like(w)
                if fused_gb:
                    gb = sink_b if sink_b is not None else g.new_empty((Cout,))
            if gx is not None:          # data gradient: on the critical path, this stream
                ws_n = _q("sstem_conv_transpose3x3s2_workspace_floats", N, Cin, H, W, Cout, 1)
                ws = x.new_empty((max(ws_n, 1),))
                with _on(x.device):
                    rc = lib.sstem_conv_transpose3x3s2_backward_ex_f32(x.data_ptr(), w.data_ptr(), g.data_ptr(), gx.data_ptr(), None, None,
                                                                       ws.data_ptr(), ws_n, N, Cin, H, W, Cout, 0, _stream())
                sstem_native.check(rc, "sstem_conv_transpose3x3s2_backward_ex_f32")
            defer = False
            if want_gw:                 # weight (+ bias) gradient: beside it when it goes into a gradient sink
                defer = sink_w is not None and _defer_wgrad_ok(ctx.params[0], ctx.params[1] if fused_gb else None, flop=18.0 * N * H * W * Cin * Cout)
                with _on_side_stream(sink_w is not None, x, g, flop=18.0 * N * H * W * Cin * Cout):
                    ws_n = _q("sstem_conv_transpose3x3s2_workspace_floats", N, Cin, H, W, Cout, 2)
                    ws = x.new_empty((max(ws_n, 1),))
                    with _on(x.device):
                        rc = lib.sstem_conv_transpose3x3s2_backward_ex_f32(x.data_ptr(), w.data_ptr(), g.data_ptr(), None, gw.data_ptr(), _ptr(gb),
                                                                           ws.data_ptr(), ws_n, N, Cin, H, W, Cout,
                                                                           3 if defer else (1 if sink_w is not None else 0), _stream())
                    sstem_native.check(rc, "sstem_conv_transpose3x3s2_backward_ex_f32")
                if defer:
                    _deferred_wgrad_issued(ws, ctx.params[0], ctx.params[1] if fused_gb else None)
            if sink_w is not None:
                gw = None
                if not defer:
                    _sink_done(ctx.params[0], ctx.params[1] if fused_gb else None)
                if fused_gb:
                    gb = None
                    want_gb = False
        if want_gb and gb is None:
            gb = g.sum((0, 2, 3))
        return gx, gw, gb, None, None, None, None, None, None, None, None, None, None


class _ConvChain(torch.autograd.Function):
    """K consecutive Conv2d(3x3, "same") [+ ReLU | LeakyReLU] groups of one FusedSequential as ONE autograd function, under
    ALGO_MFMA_BF16: the tensors between the convolutions are bf16 (stored by the producer's epilogue, read as they are by the
    consumer's staging and by its weight gradient) -- half the traffic and half the saved-activation memory, same numbers: every
    consumer of such a tensor rounds it to bf16 anyway.  One function, because autograd casts a gradient to the dtype of the tensor
    it belongs to: as separate functions the data gradients between the convolutions would be rounded to bf16.
    forward(x, spec, w0, b0, w1, b1, ...), spec = ((act, slope), ...)."""

    @staticmethod
    def forward(ctx, x, spec, *wb):
        K = len(spec)
        cur = _check(x, "input")
        saved, has_mask, dgrad_ws = [], [], []
        for i, (act, slope) in enumerate(spec):
            w = _check(wb[2 * i], "weight")
            b = _check(wb[2 * i + 1], "bias") if wb[2 * i + 1] is not None else None
            pair = _pack_pair(cur, w) if (i > 0 or x.requires_grad) else None      # both packings in one launch when a data gradient follows
            om = None
            if act == ACT_RELU and _MASK_FUSION and cur.data_ptr() % 16 == 0:        # the launch writes the mask (chains have W % 4 == 0)
                om = torch.empty((cur.shape[0], w.shape[0], cur.shape[2], cur.shape[3]), dtype=torch.bool, device=cur.device)
            y = conv3x3_bf16io(cur, w, b, None, None, act, slope, out_bf16=(i < K - 1), prepacked_ws=pair[1] if pair else None, out_mask=om)
            dgrad_ws.append(pair[2] if pair else None)
            saved += [cur, w]
            has_mask.append(act != ACT_NONE)
            if act != ACT_NONE:
                saved.append(om if om is not None else y > 0)
            cur = y
        ctx.spec, ctx.has_mask, ctx.dgrad_ws = spec, has_mask, dgrad_ws
        ctx.params = wb
        ctx.has_bias = [wb[2 * i + 1] is not None for i in range(K)]
        ctx.save_for_backward(*saved)
        return cur

    @staticmethod
    def backward(ctx, g):
        lib = sstem_native.load_library()
        saved = list(ctx.saved_tensors)
        K = len(ctx.spec)
        per = []
        pos = 0
        for i in range(K):
            xin, w = saved[pos], saved[pos + 1]
            pos += 2
            mask = None
            if ctx.has_mask[i]:
                mask = saved[pos]
                pos += 1
            per.append((xin, w, mask))
        g = _check(g, "grad_output")
        grads = [None] * (2 * K)
        for i in reversed(range(K)):
            xin, w, mask = per[i]
            act, slope = ctx.spec[i]
            N, Cin, H, W = xin.shape
            Cout = w.shape[0]
            # the ReLU mask inside the weight- and data-gradient launches (both read g through the 16-byte staging path)
            fuse = mask is not None and act == ACT_RELU and _MASK_FUSION and g.data_ptr() % 16 == 0 and xin.data_ptr() % 16 == 0 \
                and not (ctx.has_bias[i] and ctx.needs_input_grad[3 + 2 * i] and not ctx.needs_input_grad[2 + 2 * i])
            if fuse and (i > 0 or ctx.needs_input_grad[0]) and ctx.dgrad_ws[i] is None:
                fuse = _mask_fusable(_resolved_algo(N, Cout, H, W, Cin), W)      # the id the data gradient resolves to by itself
            if not fuse:
                g = _mask_grad(g, mask, act, slope)
            if ctx.needs_input_grad[2 + 2 * i]:
                want_gb = ctx.has_bias[i] and ctx.needs_input_grad[3 + 2 * i]
                sink_w = _grad_sink(ctx.params[2 * i], True)
                sink_b = _grad_sink(ctx.params[2 * i + 1], want_gb) if sink_w is not None else None
                if want_gb and sink_b is None:
                    sink_w = None
                gw = sink_w if sink_w is not None else torch.empty_like(w)
                gb = (sink_b if sink_b is not None else g.new_empty((Cout,))) if want_gb else None
                defer = sink_w is not None and _defer_wgrad_ok(ctx.params[2 * i], ctx.params[2 * i + 1] if want_gb else None,
                                                               flop=18.0 * N * H * W * Cin * Cout)
                acc = 3 if defer else (1 if sink_w is not None else 0)
                with _on_side_stream(sink_w is not None, xin, g, mask if fuse else None, flop=18.0 * N * H * W * Cin * Cout):
                    ws_n = _q("sstem_conv3x3_wgrad_workspace_floats_algo", N, Cin, H, W, Cout, ALGO_MFMA_BF16)
                    ws = g.new_empty((max(ws_n, 1),))
                    with _on(g.device):
                        if fuse:
                            rc = lib.sstem_conv3x3_backward_weight_bf16_masked(xin.data_ptr(), 1 if xin.dtype == torch.bfloat16 else 0, g.data_ptr(),
                                                                               mask.data_ptr(), gw.data_ptr(), _ptr(gb), ws.data_ptr(), ws_n,
                                                                               N, Cin, H, W, Cout, acc, _stream())
                        elif xin.dtype == torch.bfloat16:
                            rc = lib.sstem_conv3x3_backward_weight_bf16in_ex(xin.data_ptr(), g.data_ptr(), gw.data_ptr(), _ptr(gb), ws.data_ptr(), ws_n,
                                                                             N, Cin, H, W, Cout, acc, _stream())
                        else:
                            rc = lib.sstem_conv2d_backward_weight_bias_ex_f32(xin.data_ptr(), g.data_ptr(), gw.data_ptr(), _ptr(gb), ws.data_ptr(), ws_n,
                                                                              N, Cin, H, W, Cout, 3, 3, 1, 1, acc, _stream(), ALGO_MFMA_BF16)
                    sstem_native.check(rc, "conv chain weight gradient")
                if sink_w is None:
                    grads[2 * i], grads[2 * i + 1] = gw, gb
                elif defer:
                    _deferred_wgrad_issued(ws, ctx.params[2 * i], ctx.params[2 * i + 1] if want_gb else None)
                else:
                    _sink_done(ctx.params[2 * i], ctx.params[2 * i + 1] if want_gb else None)
            elif ctx.has_bias[i] and ctx.needs_input_grad[3 + 2 * i]:
                grads[2 * i + 1] = g.sum((0, 2, 3))
            if i > 0 or ctx.needs_input_grad[0]:
                pre = (ALGO_MFMA_BF16, ctx.dgrad_ws[i]) if ctx.dgrad_ws[i] is not None else None
                g = _raw_conv(g, w, None, None, None, ACT_NONE, 0.0, transposed=True, prepacked_ws=pre,     # fp32 tensors, bf16 operands
                              in_mask=mask if fuse else None)
            else:
                g = None
        return (g, None) + tuple(grads)


def conv_chain_ok(x, convs):
    """Can _ConvChain run these Conv2d modules on x?  (bf16 id with its weight-gradient kernel, a backward is being recorded,
    every layer within what the bf16-tensor kernels take)"""
    if _forced_algo != ALGO_MFMA_BF16 or not _bf16_wgrad or not _BF16_IO or len(convs) < 2:
        return False
    if not (x.is_cuda and x.dim() == 4 and x.dtype == torch.float32):
        return False
    if not _recording(x, *[p for c in convs for p in (c.weight, c.bias)]):
        return False
    lib = sstem_native.load_library()
    N, _, H, W = x.shape
    cin = x.shape[1]
    for k, c in enumerate(convs):
        if c.weight.shape[1] != cin or not lib.sstem_conv3x3_bf16io_supported(N, cin, H, W, c.weight.shape[0], 1 if k < len(convs) - 1 else 0):
            return False
        cin = c.weight.shape[0]
    return True


def conv_chain(x, convs, spec):
    args = []
    for c in convs:
        args += [c.weight, c.bias]
    return _ConvChain.apply(x, tuple(spec), *args)


def first_layer_u8_ok(frames, conv):
    """May ``first_layer_u8`` run this layer?  uint8 frames [N,2,H,W] on the GPU, a Conv2d(6 -> 6, 3x3, padding 1), nothing recorded."""
    return (isinstance(frames, torch.Tensor) and frames.is_cuda and frames.dtype == torch.uint8 and frames.dim() == 4 and frames.shape[1] == 2
            and isinstance(conv, torch.nn.Conv2d) and tuple(conv.weight.shape) == (6, 6, 3, 3) and conv.padding == (1, 1)
            and conv.stride == (1, 1) and conv.weight.dtype == torch.float32 and not _recording(conv.weight, conv.bias)
            and bool(_q("sstem_conv3x3_first_layer_u8_supported", frames.shape[0], frames.shape[2], frames.shape[3], 6)))


def first_layer_u8(frames, conv, act=ACT_RELU, slope=0.0):
    """The IFNet's first convolution straight from the two uint8 frames (include/sstem_conv.h, sstem_conv3x3_first_layer_u8; the
    reference builds [1,6,H,W] float32 = each frame / 255 replicated x3: inference_singleImage.py:55-66).  Returns (the layer's output
    [N,6,H,W] with its bound, the normalised float32 planes [2,N,1,H,W], frame-major) -- the bits of the fp32 launch on the
    materialised input."""
    frames = frames.contiguous()
    N, _, H, W = frames.shape
    out = torch.empty((N, 6, H, W), dtype=torch.float32, device=frames.device)
    planes = torch.empty((2, N, 1, H, W), dtype=torch.float32, device=frames.device)
    word = _new_amax_word(frames.device)
    lib = sstem_native.load_library()
    with _on(frames.device):
        rc = lib.sstem_conv3x3_first_layer_u8(frames.data_ptr(), conv.weight.data_ptr(), _ptr(conv.bias), out.data_ptr(), planes.data_ptr(),
                                              word.data_ptr(), N, H, W, 6, act, float(slope), _stream())
    sstem_native.check(rc, "sstem_conv3x3_first_layer_u8")
    return tag_amax(out, word), planes


def _recording(*tensors):
    """Can a backward follow this call?  (grad mode on and something to differentiate)"""
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


def conv2d_fused(x, w, b=None, scale=None, shift=None, act=ACT_NONE, slope=0.0, owner=None, residual=None, res_scale=1.0, bn_part=None,
                 out=None, out_blocked=False, pool_out=None, pool_kind=0, pool_only=False):
    """owner: the nn.Module that owns w (FusedSequential passes it): when no backward can follow, the packed weights of the 3x3
    MFMA launch are kept on it and the next call skips its packing launch.  residual / res_scale: out = (act(..) + residual) *
    res_scale in the store (only when nothing is recorded).  bn_part: see bn_partials_for.  out: a contiguous fp32 tensor of the
    result's shape to store into (only when nothing is recorded)."""
    if pool_only:         # (pooled_store_ok has said yes: nothing is recorded) the pooled copy is the launch's only result
        assert pool_out is not None and out is None and not _recording(x, w, b)
        # the checks _Conv2dFused.forward makes (this branch bypasses it): device, dtype, and a contiguous view of everything the launch
        # reads through data_ptr()
        x = _check(x, "input"); w = _check(w, "weight")
        b = _check(b, "bias") if b is not None else None
        scale = _check(scale, "scale") if scale is not None else None
        shift = _check(shift, "shift") if shift is not None else None
        return _raw_conv(x, w, b, scale, shift, act, slope, owner=owner, pool_out=pool_out, pool_kind=pool_kind, pool_only=True)
    res = _Conv2dFused.apply(x, w, b, scale, shift, act, slope, _recording(x, w, b), owner, residual, res_scale, bn_part, out, out_blocked,
                             pool_out, pool_kind)
    if out is not None and res is not out:       # autograd hands back an alias of a tensor that came in as an argument: the bound rides on
        word = amax_word_of(out)                 # the object the launch tagged
        if word is not None:
            tag_amax(res, word)
    return res


def can_store_into(x, w, b=None):
    """May conv2d_fused(x, w, b, ..., out=...) be used?  (a GPU fp32 launch nothing is recorded for)"""
    return x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and not _recording(x, w, b)


def conv_transpose3x3s2_fused(x, w, b=None, scale=None, shift=None, act=ACT_NONE, slope=0.0, owner=None, residual=None, res_scale=1.0,
                              bn_part=None, out=None):
    res = _ConvT3x3s2Fused.apply(x, w, b, scale, shift, act, slope, _recording(x, w, b), owner, residual, res_scale, bn_part, out)
    if out is not None and res is not out:
        word = amax_word_of(out)
        if word is not None:
            tag_amax(res, word)
    return res


_RESIDUAL_FUSION = os.environ.get("SSTEM_RESIDUAL_FUSION", "1") != "0"      # developer knob (A/B runs)


def residual_fusable(x, conv, residual):
    """Can the launch of `conv` on x add `residual` in its store?  (fp32 3x3 MFMA convolution or the native ConvTranspose, nothing
    recorded for a backward, a contiguous fp32 residual of the output's shape)"""
    if not _RESIDUAL_FUSION or torch.is_grad_enabled() and (x.requires_grad or residual.requires_grad or conv.weight.requires_grad):
        return False
    if not (x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and residual.dtype == torch.float32 and residual.is_contiguous()):
        return False
    transposed = isinstance(conv, torch.nn.ConvTranspose2d)
    if tuple(conv.weight.shape[2:]) != (3, 3) or _forced_algo not in (ALGO_AUTO, ALGO_MFMA) + _SPLIT_ALGOS or (transposed and _convT_route() != "native"):
        return False
    N, _, H, W = x.shape
    Cout = conv.weight.shape[1] if transposed else conv.weight.shape[0]
    if _forced_algo == ALGO_AUTO and not transposed and N * ((Cout + 31) // 32) >= 65536:
        return False
    return tuple(residual.shape) == ((N, Cout, 2 * H, 2 * W) if transposed else (N, Cout, H, W))


# Statistics partials from the convolution's own store (bn_partials_for): OFF by default.  It removes the BatchNorm's statistics
# pass (one read of the tensor, one launch per layer), but the in-register reduction it adds to every workgroup's epilogue (two
# passes of wave shuffles over 16 channel values per lane, an LDS round, four barriers) costs slightly more than that pass on this
# chip: same-box A/B on the SFF fusion step (profiles/r02/i_feature_ablation.txt) 4.28 vs 4.32 ms at 2 samples, 20.64 vs 20.79 ms at
# 16 with it OFF vs ON.  SSTEM_BN_FUSED_STATS=1 turns it on; the tests run both.
_BN_FUSED_STATS = os.environ.get("SSTEM_BN_FUSED_STATS", "0") == "1"


def bn_partials_for(x, conv):
    """A [Cout, P, 3] tensor for the train-mode BatchNorm statistics partials the launch of `conv` (nn.Conv2d 3x3 "same" or the
    up-sampling nn.ConvTranspose2d) on x can write while it stores its output, or None when that launch writes none (other
    kernel sizes, the direct / bf16 ids, a ConvTranspose launch split over K)."""
    if not _BN_FUSED_STATS or not x.is_cuda or x.dim() != 4 or x.dtype != torch.float32:
        return None
    transposed = isinstance(conv, torch.nn.ConvTranspose2d)
    if _forced_algo not in (ALGO_AUTO, ALGO_MFMA) or (transposed and _convT_route() != "native"):
        return None
    N, Cin, H, W = x.shape
    Cout = conv.weight.shape[1] if transposed else conv.weight.shape[0]
    KH, KW = conv.weight.shape[2:]
    P = _q("sstem_conv_bn_partials", N, Cin, H, W, Cout, KH, KW, 1 if transposed else 0, _forced_algo)
    if P <= 0:
        return None
    return x.new_empty((Cout, P, 3))


class _UpsampleBilinear2x(torch.autograd.Function):
    """Native forward and backward: torch's forward kernel collapses on many small planes (0.3-1.3 ms per call on the trunk
    planes of a 256x256 training step, 8 calls = 2.7 ms of a 28 ms IFNet step); its backward scatters with atomic adds (0.08 ms
    per call, 7 % of the bf16 IFNet step) -- the native one is a deterministic gather with the forward kernel's own weights."""

    @staticmethod
    def forward(ctx, x):
        ctx.in_size = tuple(x.shape)
        return upsample_bilinear2x(x)

    @staticmethod
    def backward(ctx, g):
        N, C, H, W = ctx.in_size
        g = _check(g, "grad_output")
        gin = g.new_empty((N, C, H, W))
        lib = sstem_native.load_library()
        with _on(g.device):
            rc = lib.sstem_upsample_bilinear2x_backward_f32(g.data_ptr(), gin.data_ptr(), N * C, H, W, _stream())
        sstem_native.check(rc, "sstem_upsample_bilinear2x_backward_f32")
        # a source pixel gathers at most 3 x 3 weights of at most 1 (align_corners, scale (H - 1) / (2H - 1) >= 1/3): 16 x g's bound holds
        w = amax_word_of(g)
        if w is not None:
            tag_amax(gin, w * 16.0)
        return gin


# Measured on MI355X (tools/bench_upsample.py): torch's forward kernel collapses on many small planes (8x512x32x32: 1.31 ms =
# 64 GB/s against 0.03 ms natively) and is 20-30 % slower at 256x256 planes; on the 512x512 planes of the kernel heads at C2 it was 5-10 %
# faster than the first native kernel (0.76-0.80 against 0.85 ms) and is 9 % slower than the current one (0.79 against 0.72 ms: one
# 16-byte load per source row and the next plane's values in flight).  A 1.7 GB fill takes 0.25 ms: neither kernel is near the bound.
NATIVE_UPSAMPLE_MAX_PIXELS = 1024 * 1024      # (round 2: with 16-byte source loads the native kernel is ahead on 512 x 512 planes too: 0.716 vs 0.788 ms)


def upsample_bilinear2x_module(m, x):
    """Run an nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True) module `m` on x: native forward where it wins
    (with the native gather backward when a gradient is being recorded), the module itself otherwise."""
    if x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and x.shape[3] % 2 == 0 \
            and x.shape[2] * x.shape[3] <= NATIVE_UPSAMPLE_MAX_PIXELS:
        if torch.is_grad_enabled() and x.requires_grad:
            return hand_on_amax(x, _UpsampleBilinear2x.apply(x))
        return hand_on_amax(x, upsample_bilinear2x(x))      # interpolation weights are convex: x's bound holds
    return m(x)


def is_bilinear2x(m):
    if not isinstance(m, torch.nn.Upsample) or m.mode != "bilinear" or not m.align_corners or m.size is not None:
        return False
    sf = m.scale_factor
    return (sf == 2 or sf == 2.0) if not isinstance(sf, (tuple, list)) else (len(sf) == 2 and sf[0] == 2 and sf[1] == 2)


def upsample_bilinear2x(x, out=None):
    """nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True) forward as one native launch (no autograd:
    see upsample_bilinear2x_module for the differentiable entry).  out: a contiguous [N,C,2H,2W] fp32 tensor to write into."""
    x = _check(x, "input")
    N, C, H, W = x.shape
    if out is None:
        out = x.new_empty((N, C, 2 * H, 2 * W))
    else:
        assert out.shape == (N, C, 2 * H, 2 * W) and out.dtype == torch.float32 and out.device == x.device and out.is_contiguous()
    lib = sstem_native.load_library()
    with _on(x.device):
        rc = lib.sstem_upsample_bilinear2x_f32(x.data_ptr(), out.data_ptr(), N * C, H, W, _stream())
    sstem_native.check(rc, "sstem_upsample_bilinear2x_f32")
    return out


def skip_cat_upsample2x(m, skip, x, cat=None):
    """torch.cat([skip, up(x)], 1) for the U-Nets' decoder levels (networks.py Up.forward), `m` the nn.Upsample module.  cat: the
    concatenated tensor when `skip` already IS its first channel block (networks.UNet places its encoder outputs there): only the
    up-sampled half is written then.  When no
    gradient is being recorded, the sizes match and the native up-sampling applies, the up-sampled planes are written straight into their
    half of the concatenated tensor, image by image (the planes of one image are one contiguous run there): the concatenation copies the
    skip only.  Otherwise: the module's own path and torch.cat (a zero-width F.pad is left out: torch returns a clone for it)."""
    native = is_bilinear2x(m) and x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and x.shape[3] % 2 == 0 \
        and x.shape[2] * x.shape[3] <= NATIVE_UPSAMPLE_MAX_PIXELS
    recording = torch.is_grad_enabled() and (x.requires_grad or skip.requires_grad)
    N, C, H, W = x.shape
    if native and not recording and skip.dtype == torch.float32 and skip.shape[0] == N and tuple(skip.shape[2:]) == (2 * H, 2 * W):
        x = _check(x, "input")
        Cs = skip.shape[1]
        if cat is not None and cat.shape == (N, Cs + C, 2 * H, 2 * W) and cat.data_ptr() == skip.data_ptr():
            out = cat
        else:
            out = x.new_empty((N, Cs + C, 2 * H, 2 * W))
            out[:, :Cs].copy_(skip)
        for n in range(N):
            upsample_bilinear2x(x[n:n + 1], out=out[n:n + 1, Cs:])
        ws, wx = amax_word_of(skip), amax_word_of(x)
        if ws is not None and wx is not None:               # the concatenation's bound: slot-wise maximum of the two halves' words
            tag_amax(out, torch.maximum(ws, wx))
        return out
    up = upsample_bilinear2x_module(m, x) if is_bilinear2x(m) else m(x)
    dy, dx = skip.size(2) - up.size(2), skip.size(3) - up.size(3)
    if dy or dx:
        up = torch.nn.functional.pad(up, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
        return torch.cat([skip, up], dim=1)
    return tag_concat_amax(torch.cat([skip, up], dim=1), skip, up)


# ---- 2 x 2 / stride 2 pooling (include/sstem_resize.h) --------------------------------------------------------------------
# aten's max-pool backward takes 132 us on a [16,32,256,256] gradient (64-bit flat indices, one thread per input element searching its
# window) and its forward 31 us on the output side; the native pair streams: one argmax BYTE per output, 40 us / 20 us.  Same values
# as torch bit for bit (the average adds in torch's order; the maximum takes the first largest element, NaN propagates).
_NATIVE_POOL = os.environ.get("SSTEM_NATIVE_POOL", "1") != "0"      # developer knob (A/B runs)


def _pool_kind(m):
    """'max' / 'avg' when m is the 2 x 2 / stride-2 pooling the networks use (no padding, floor mode), else None."""
    def two(v):
        return v == 2 or v == (2, 2) or v == [2, 2]
    if isinstance(m, torch.nn.MaxPool2d):
        ok = two(m.kernel_size) and (m.stride is None or two(m.stride)) and m.padding in (0, (0, 0)) and m.dilation in (1, (1, 1)) \
            and not m.ceil_mode and not m.return_indices
        return "max" if ok else None
    if isinstance(m, torch.nn.AvgPool2d):
        ok = two(m.kernel_size) and (m.stride is None or two(m.stride)) and m.padding in (0, (0, 0)) and not m.ceil_mode \
            and m.divisor_override is None
        return "avg" if ok else None
    return None


class _Pool2x2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, is_max, recording):
        lib = sstem_native.load_library()
        if not recording and x.is_cuda and not x.is_contiguous() and is_channel_block(x):
            # a channel block of a larger tensor (an encoder output stored inside the tensor its decoder concatenates): the planes of one
            # image are contiguous -- one launch per image instead of a copy of the whole tensor
            N, C, H, W = x.shape
            out = x.new_empty((N, C, H // 2, W // 2))
            with _on(x.device):
                for n in range(N):
                    rc = lib.sstem_pool2x2_forward_f32(x[n].data_ptr(), out[n].data_ptr(), None, C, H, W, 1 if is_max else 0, _stream())
                    sstem_native.check(rc, "sstem_pool2x2_forward_f32")
            ctx.is_max, ctx.in_shape = is_max, (N, C, H, W)
            return out
        x = _check(x, "input")
        N, C, H, W = x.shape
        out = x.new_empty((N, C, H // 2, W // 2))
        idx = torch.empty((N, C, H // 2, W // 2), dtype=torch.uint8, device=x.device) if (is_max and recording) else None
        with _on(x.device):
            rc = lib.sstem_pool2x2_forward_f32(x.data_ptr(), out.data_ptr(), _ptr(idx), N * C, H, W, 1 if is_max else 0, _stream())
        sstem_native.check(rc, "sstem_pool2x2_forward_f32")
        ctx.is_max, ctx.in_shape = is_max, (N, C, H, W)
        if idx is not None:
            ctx.save_for_backward(idx)
        return out

    @staticmethod
    def backward(ctx, g):
        N, C, H, W = ctx.in_shape
        g = _check(g, "grad_output")
        idx = ctx.saved_tensors[0] if ctx.is_max else None
        gin = g.new_empty((N, C, H, W))
        lib = sstem_native.load_library()
        with _on(g.device):
            rc = lib.sstem_pool2x2_backward_f32(g.data_ptr(), _ptr(idx), gin.data_ptr(), N * C, H, W, 1 if ctx.is_max else 0, _stream())
        sstem_native.check(rc, "sstem_pool2x2_backward_f32")
        return hand_on_amax(g, gin), None, None              # g's elements (or quarters of them) and zeros: g's bound holds


def pool_module(m, x):
    """Run a pooling module `m` on x: the native 2 x 2 kernels for the nn.MaxPool2d(2) / nn.AvgPool2d((2,2),(2,2)) of the reference's
    networks on fp32 GPU tensors, the module itself otherwise."""
    kind = _pool_kind(m) if _NATIVE_POOL else None
    if kind is None or not (x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and x.shape[2] >= 2 and x.shape[3] >= 2):
        return m(x)
    recording = torch.is_grad_enabled() and x.requires_grad
    out = _Pool2x2.apply(x, kind == "max", recording)
    return hand_on_amax(x, out)                              # a maximum / an average of four elements: x's bound holds


class _BatchNormTrainAct(torch.autograd.Function):
    """Train-mode BatchNorm2d (+ ReLU / LeakyReLU) as two native streaming passes forward and two backward
    (include/sstem_norm.h).  Saves x, the affine parameters and the two per-channel statistics only: the activation mask is
    recomputed from x in the backward.  running_mean / running_var are updated in place by the forward launch."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum, eps, act, slope, partials=None, nbt=None):
        x = _check(x, "input")
        N, C, H, W = x.shape
        lib = sstem_native.load_library()
        y = torch.empty_like(x)
        save_mean = x.new_empty((C,)); save_invstd = x.new_empty((C,))
        ws, ws_n = None, 0
        if partials is None:             # the statistics launch runs first
            ws_n = _q("sstem_batchnorm_workspace_floats", N, C, H * W)
            ws = x.new_empty((max(ws_n, 1),))
        y_word = _new_amax_word(x.device) if _bounds_wanted() else None     # the next convolution's fp16 launch scales by it
        with _on(x.device):
            rc = lib.sstem_batchnorm_train_forward_amax_f32(x.data_ptr(), _ptr(weight), _ptr(bias), _ptr(running_mean), _ptr(running_var),
                                                            _ptr(nbt), y.data_ptr(), _ptr(y_word), save_mean.data_ptr(), save_invstd.data_ptr(),
                                                            _ptr(partials), partials.shape[1] if partials is not None else 0, _ptr(ws), ws_n,
                                                            N, C, H * W, float(momentum), float(eps), act, float(slope), _stream())
        sstem_native.check(rc, "sstem_batchnorm_train_forward_amax_f32")
        if y_word is not None:
            tag_amax(y, y_word)
        ctx.act, ctx.slope = act, slope
        ctx.params = (weight, bias)
        ctx.has_affine = weight is not None
        ctx.save_for_backward(x, weight, bias, save_mean, save_invstd)
        return y

    @staticmethod
    def backward(ctx, g):
        x, weight, bias, save_mean, save_invstd = ctx.saved_tensors
        g = _check(g, "grad_output")
        N, C, H, W = x.shape
        lib = sstem_native.load_library()
        dx = torch.empty_like(x)
        sink_w = _grad_sink(ctx.params[0], ctx.has_affine and ctx.needs_input_grad[1])
        sink_b = _grad_sink(ctx.params[1], ctx.has_affine and ctx.needs_input_grad[2]) if sink_w is not None else None
        sunk = sink_w is not None and sink_b is not None
        dw = sink_w if sunk else (x.new_empty((C,)) if ctx.has_affine else None)
        db = sink_b if sunk else (x.new_empty((C,)) if ctx.has_affine else None)
        ws_n = _q("sstem_batchnorm_workspace_floats", N, C, H * W)
        ws = x.new_empty((max(ws_n, 1),))
        dx_word = _new_amax_word(x.device) if _bounds_wanted() else None    # the convolution below scales its gradient launches by it
        with _on(x.device):
            rc = lib.sstem_batchnorm_train_backward_amax_f32(g.data_ptr(), x.data_ptr(), _ptr(weight), _ptr(bias), save_mean.data_ptr(),
                                                             save_invstd.data_ptr(), dx.data_ptr(), _ptr(dx_word), _ptr(dw), _ptr(db), ws.data_ptr(),
                                                             ws_n, N, C, H * W, ctx.act, float(ctx.slope), 1 if sunk else 0, _stream())
        sstem_native.check(rc, "sstem_batchnorm_train_backward_amax_f32")
        if dx_word is not None:
            tag_amax(dx, dx_word)
        if sunk:
            dw = db = None
            _sink_done(ctx.params[0], ctx.params[1])
        return dx, dw, db, None, None, None, None, None, None, None, None


def batchnorm_train_act(bn, x, act=ACT_NONE, slope=0.0, partials=None):
    """Train-mode forward of the nn.BatchNorm2d module `bn` (+ activation) on x, with torch's bookkeeping: the momentum /
    cumulative-average factor and num_batches_tracked (torch/nn/modules/batchnorm.py), running statistics updated in place.
    partials: the [C, P, 3] statistics partials the producing convolution wrote (bn_partials_for) -- no statistics pass then."""
    if x.numel() // x.shape[1] <= 1:       # torch.nn.functional.batch_norm refuses this in training mode, with this message
        raise ValueError("Expected more than 1 value per channel when training, got input size %s" % (x.size(),))
    factor = 0.0 if bn.momentum is None else bn.momentum
    nbt = None
    if bn.track_running_stats and bn.num_batches_tracked is not None:
        if bn.momentum is None:            # cumulative average: the factor needs the counter's value on the host
            bn.num_batches_tracked.add_(1)
            factor = 1.0 / float(bn.num_batches_tracked)
        elif bn.num_batches_tracked.is_cuda and bn.num_batches_tracked.dtype == torch.int64:
            nbt = bn.num_batches_tracked   # incremented by the launch itself (a launch of its own per BatchNorm layer before)
        else:
            bn.num_batches_tracked.add_(1)
    rm = bn.running_mean if bn.track_running_stats else None
    rv = bn.running_var if bn.track_running_stats else None
    out = _BatchNormTrainAct.apply(x, bn.weight if bn.affine else None, bn.bias if bn.affine else None, rm, rv, factor, bn.eps, act, slope,
                                   partials, nbt)
    if rm is not None:      # the launch updated them through raw pointers: tell autograd (the eval-mode fold cache is keyed on the versions)
        torch.autograd.graph.increment_version(rm)
        torch.autograd.graph.increment_version(rv)
    if nbt is not None:
        torch.autograd.graph.increment_version(nbt)
    return out
