"""GPU parity of the bf16-COEFFICIENT entry points (include/sstem_sepconv.h: sstem_sepconv_forward_bf16coef, ..._backward_bf16coef,
..._interp_apply_gray_bf16coef; BASELINE config 5 "bf16 activations with fp32 sepconv accumulate").

The rounding point is ONE: the coefficient tensors are bfloat16 (round-to-nearest-even, done by the producer -- here torch's
``.bfloat16()``); everything else is the fp32 arithmetic of the _f32 entry points.  So the tests say:
  * against the CPU oracle on the ROUNDED coefficients (``k.bfloat16().float()``): the fp32 tests' tolerance, 2e-5 of the largest value;
  * against the _f32 entry points on the rounded coefficients: bit for bit on x3-replicated grayscale frames (the streaming kernels:
    the same instruction sequence on the same values -- a bf16 is widened exactly), and bit for bit against the one-lane-per-element
    kernels on anything else;
  * the byte model: the two 51 H W terms halved.
"""
import numpy as np
import pytest
import torch

import libs.sepconv._ext.cunnex as cunnex
import sstem_native
from libs.sepconv.SeparableConvolution import SeparableConvolution
from libs.sepconv.fused import interp_apply_gray, interp_apply_gray_bf16coef, interp_apply_gray_bf16coef_supported
from oracle import sepconv_c
from sepconv_cases import make_case

pytestmark = pytest.mark.gpu
REL = 2e-5


@pytest.fixture(autouse=True)
def _reset_algo():
    yield
    cunnex.set_algorithm(cunnex.ALGO_AUTO)


def _close(a, ref, rel=REL):
    scale = float(np.abs(ref).max()) + 1e-12
    err = float(np.abs(a - ref).max())
    assert err <= rel * scale, "max err %.3e vs scale %.3e" % (err, scale)


def _case(seed, B, H, W, gray):
    inp, ver, hor, grad = make_case(seed, B, 3, H, W)
    if gray:
        inp = np.repeat(inp[:, :1], 3, axis=1).copy()
    ti, tg = torch.from_numpy(inp).cuda(), torch.from_numpy(grad).cuda()
    v16, h16 = torch.from_numpy(ver).cuda().bfloat16(), torch.from_numpy(hor).cuda().bfloat16()
    return ti, tg, v16, h16


# tile-aligned, ragged, smaller than a tile, odd width, several images; with and without identical channels
SHAPES = [(2, 32, 64), (1, 37, 70), (1, 5, 9), (1, 70, 131), (3, 33, 65)]


@pytest.mark.parametrize("gray", [True, False])
@pytest.mark.parametrize("shape", SHAPES)
def test_forward_bf16_coefficients(shape, gray):
    B, H, W = shape
    ti, _, v16, h16 = _case(31, B, H, W, gray)
    out = SeparableConvolution.apply(ti, v16, h16)
    assert out.dtype == torch.float32
    vr, hr = v16.float(), h16.float()
    _close(out.cpu().numpy(), sepconv_c.forward(ti.cpu().numpy(), vr.cpu().numpy(), hr.cpu().numpy()))
    if not gray:
        cunnex.set_algorithm(cunnex.ALGO_DIRECT)
    same = SeparableConvolution.apply(ti, vr, hr)          # the _f32 entry on the rounded values: the same bits
    assert torch.equal(out, same)


@pytest.mark.parametrize("gray", [True, False])
@pytest.mark.parametrize("shape", SHAPES)
def test_backward_bf16_coefficients(shape, gray):
    B, H, W = shape
    ti, tg, v16, h16 = _case(32, B, H, W, gray)
    v16.requires_grad_(); h16.requires_grad_()
    out = SeparableConvolution.apply(ti, v16, h16)
    out.backward(tg)
    assert v16.grad.dtype == torch.bfloat16 and h16.grad.dtype == torch.bfloat16          # autograd's contract: the input's dtype
    # the fp32 gradients the entry point itself returns
    vr, hr = v16.detach().float(), h16.detach().float()
    gv, gh = torch.empty_like(vr), torch.empty_like(hr)
    cunnex.SeparableConvolution_cuda_backward(tg, ti, v16.detach(), h16.detach(), None, gv, gh)
    _, rv, rh = sepconv_c.backward(tg.cpu().numpy(), ti.cpu().numpy(), vr.cpu().numpy(), hr.cpu().numpy())
    _close(gv.cpu().numpy(), rv)
    _close(gh.cpu().numpy(), rh)
    assert torch.equal(v16.grad, gv.bfloat16()) and torch.equal(h16.grad, gh.bfloat16())
    if not gray:
        cunnex.set_algorithm(cunnex.ALGO_DIRECT)
    gv2, gh2 = torch.empty_like(vr), torch.empty_like(hr)
    cunnex.SeparableConvolution_cuda_backward(tg, ti, vr, hr, None, gv2, gh2)             # the _f32 entry on the rounded values
    assert torch.equal(gv, gv2) and torch.equal(gh, gh2)


@pytest.mark.parametrize("shape", [(2, 32, 64), (1, 37, 70), (1, 100, 130), (2, 7, 200), (8, 256, 256)])
def test_fused_apply_bf16_coefficients(shape):
    B, H, W = shape
    assert interp_apply_gray_bf16coef_supported(B, H, W)
    g = torch.Generator().manual_seed(33)
    g1, g2 = torch.rand(B, 1, H, W, generator=g).cuda(), torch.rand(B, 1, H, W, generator=g).cuda()
    ks = [torch.softmax(torch.randn(B, 51, H, W, generator=g), 1).cuda().bfloat16() for _ in range(4)]
    out = interp_apply_gray_bf16coef(g1, g2, *ks)
    same = interp_apply_gray(g1, g2, *(k.float() for k in ks))
    assert torch.equal(out, same)
    if H * W <= 130 * 100:          # the oracle on the rounded coefficients (model_interp.py:90-97)
        pad = ((0, 0), (0, 0), (25, 25), (25, 25))
        r1, r2 = (np.pad(np.repeat(t.cpu().numpy(), 3, 1), pad, mode="edge") for t in (g1, g2))
        kk = [k.float().cpu().numpy() for k in ks]
        ref = (sepconv_c.forward(r2, kk[2], kk[3]) + sepconv_c.forward(r1, kk[0], kk[1])).mean(axis=1, keepdims=True)
        assert np.abs(out.cpu().numpy() - ref).max() <= 1e-4          # [0,1] images, normalised kernels: north_star's bound


def test_bf16_coefficient_entry_points_refuse_what_they_cannot_do():
    ti, tg, v16, h16 = _case(34, 1, 8, 16, True)
    with pytest.raises(TypeError):
        SeparableConvolution.apply(ti, v16, h16.float())                                  # mixed coefficient dtypes
    with pytest.raises(TypeError):
        SeparableConvolution.apply(ti.bfloat16(), v16, h16)                               # the frames stay float32
    with pytest.raises(NotImplementedError):
        SeparableConvolution.apply(ti.cpu(), v16.cpu(), h16.cpu())                        # no CPU path, as the reference


def test_byte_model_halves_the_coefficient_terms():
    lib = sstem_native.load_library()
    B, C, H, W = 8, 3, 1024, 1024
    coef = B * 51 * H * W
    assert lib.sstem_sepconv_forward_bytes(B, C, H, W) - lib.sstem_sepconv_forward_bytes_bf16coef(B, C, H, W) == 2 * 2 * coef
    assert lib.sstem_sepconv_backward_bytes(B, C, H, W) - lib.sstem_sepconv_backward_bytes_bf16coef(B, C, H, W) == 2 * 2 * coef
    assert lib.sstem_sepconv_interp_apply_bytes(B, H, W, 1) - lib.sstem_sepconv_interp_apply_bytes_bf16coef(B, H, W, 1) == 2 * 4 * coef
