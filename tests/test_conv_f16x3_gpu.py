"""GPU parity tests of the fp16 two-piece convolution id (ALGO_MFMA_F16X3; csrc/conv_split_kernels.hip, include/sstem_conv.h): fp32
operands as two fp16 pieces under per-tensor power-of-two scales, three exact products per term summed in fp32 -- 2^-22 per
product.  Reference: float64 PyTorch on the CPU, identical inputs.  Tolerance: the fp32 kernels' own bound, max|a-ref| <= 2e-5 *
max|ref| (+1e-6), unchanged (tests/test_conv_gpu.py, tests/test_conv_split_gpu.py).  NOT required of this id, and stated in the
header: a one-hot weight does not copy its input bit for bit (22 of 24 bits survive the split); that is pinned here as a bound."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

import hipnn.functional as HF
import sstem_native
from hipnn import FusedSequential
from test_conv_split_gpu import SHAPES, _act_ref, _close, _err

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _reset_algo():
    yield
    HF.set_algorithm(HF.ALGO_AUTO)


# + layers whose launch is split over K (the bound then comes from the slice-sum launch) and a 64-channel-block layer with many tiles
F16_SHAPES = SHAPES + [(2, 256, 32, 32, 256), (2, 512, 16, 16, 256), (1, 64, 64, 96, 64)]


@pytest.mark.parametrize("shape", F16_SHAPES)
def test_f16x3_conv3x3_forward_fused(shape):
    HF.set_algorithm(HF.ALGO_MFMA_F16X3)
    N, Cin, H, W, Cout = shape
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, Cin, H, W, generator=g); w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.2
    b = torch.randn(Cout, generator=g); sc = torch.rand(Cout, generator=g) + 0.5; sh = torch.randn(Cout, generator=g)
    for act, slope in ((HF.ACT_NONE, 0.0), (HF.ACT_RELU, 0.0), (HF.ACT_LEAKY, 0.2)):
        out = HF.conv2d_fused(x.cuda(), w.cuda(), b.cuda(), sc.cuda(), sh.cuda(), act, slope)
        ref = F.conv2d(x.double(), w.double(), b.double(), padding=1) * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)
        ref = _act_ref(ref, act, slope)
        _close(out, ref)
        # the bound the launch leaves behind is the largest magnitude it stored, exactly
        word = HF.amax_word_of(out)
        assert word is not None and word.max().item() == out.abs().max().item()
    _close(HF.conv2d_fused(x.cuda(), w.cuda()), F.conv2d(x.double(), w.double(), padding=1))


@pytest.mark.parametrize("xs,ws", [(1e-6, 1.0), (3e5, 1e-3), (1.0, 4e4), (7e-12, 9e-9), (1e18, 1e-18)])
def test_f16x3_is_scale_invariant(xs, ws):
    """fp16 has five exponent bits; the per-tensor scales must make the result independent of the tensors' magnitudes: inputs and
    weights multiplied by arbitrary factors give the same relative error as at unit scale (gradient-sized and huge tensors alike)."""
    HF.set_algorithm(HF.ALGO_MFMA_F16X3)
    g = torch.Generator().manual_seed(11)
    N, Cin, H, W, Cout = 2, 48, 20, 40, 70
    x = torch.randn(N, Cin, H, W, generator=g) * xs; w = torch.randn(Cout, Cin, 3, 3, generator=g) * ws
    out = HF.conv2d_fused(x.cuda(), w.cuda())
    assert torch.isfinite(out).all()
    _close_rel(out, F.conv2d(x.double(), w.double(), padding=1), 2e-6)


def _close_rel(a, ref, rel):
    err, scale = _err(a, ref)
    assert err <= rel * scale, "max err %.3e vs scale %.3e" % (err, scale)


def test_f16x3_wide_dynamic_range_inside_one_tensor():
    """Values 2^-30 .. 1 of the bound in ONE tensor: elements more than 18 binades below the bound fade out with an ABSOLUTE error
    of 2^-25 of the bound (documented), so the result stays within the fp32 tolerance of max|ref| -- nothing overflows, nothing
    turns into garbage."""
    HF.set_algorithm(HF.ALGO_MFMA_F16X3)
    g = torch.Generator().manual_seed(12)
    N, Cin, H, W, Cout = 1, 32, 24, 64, 32
    x = torch.randn(N, Cin, H, W, generator=g) * torch.exp2(-30.0 * torch.rand(N, Cin, H, W, generator=g))
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * torch.exp2(-30.0 * torch.rand(Cout, Cin, 3, 3, generator=g))
    _close(HF.conv2d_fused(x.cuda(), w.cuda()), F.conv2d(x.double(), w.double(), padding=1))


def test_f16x3_long_same_sign_sums_and_one_hot_bound():
    """K = 9 * 256 same-sign products (every systematic error would add up): within 2^-20 of max|ref| and within 3 x the fp32 MFMA
    kernel's own distance from float64.  One-hot weights: the shifted input to 2^-21 (NOT bit for bit: that is the X6 id's property)."""
    torch.manual_seed(8)
    N, Cin, H, W, Cout = 1, 256, 24, 64, 64
    x = torch.rand(N, Cin, H, W, device="cuda") + 0.5
    w = torch.rand(Cout, Cin, 3, 3, device="cuda") + 0.1
    ref = F.conv2d(x.double().cpu(), w.double().cpu(), padding=1)
    res = {}
    for algo in (HF.ALGO_MFMA, HF.ALGO_MFMA_F16X3):
        HF.set_algorithm(algo)
        res[algo] = (HF.conv2d_fused(x, w).cpu().double() - ref).abs().max().item() / ref.abs().max().item()
    print("same-sign sums: fp32 MFMA %.2e, f16x3 %.2e of max|ref|" % (res[HF.ALGO_MFMA], res[HF.ALGO_MFMA_F16X3]))
    assert res[HF.ALGO_MFMA_F16X3] <= 2.0 ** -20 and res[HF.ALGO_MFMA_F16X3] <= 3.0 * res[HF.ALGO_MFMA] + 1e-7
    HF.set_algorithm(HF.ALGO_MFMA_F16X3)
    x = torch.randn(2, 20, 19, 40, device="cuda") * 3
    w = torch.zeros(35, 20, 3, 3, device="cuda")
    for co in range(35):
        w[co, (co * 7) % 20, co % 3, (co // 3) % 3] = 1.0
    out = HF.conv2d_fused(x, w)
    ref = F.conv2d(x.double().cpu(), w.double().cpu(), padding=1)
    assert (out.cpu().double() - ref).abs().max().item() <= 2.0 ** -21 * ref.abs().max().item()


def _block(cin, cmid, cout):
    return FusedSequential(nn.Conv2d(cin, cmid, 3, padding=1), nn.ReLU(), nn.Conv2d(cmid, cmid, 3, padding=1), nn.BatchNorm2d(cmid),
                           nn.LeakyReLU(0.2), nn.MaxPool2d(2), nn.Conv2d(cmid, cout, 3, padding=1), nn.ReLU(),
                           nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True), nn.Conv2d(cout, cout, 3, padding=1))


def _ref_of(net):
    """The same modules as a plain nn.Sequential in float64 on the CPU (FusedSequential has no CPU path)."""
    import copy
    return nn.Sequential(*copy.deepcopy(list(net))).double().cpu().eval()


def test_bounds_travel_with_the_tensors_through_an_inference_chain():
    """Every 3x3 launch of a Conv / BatchNorm(eval) / activation / pooling / up-sampling chain under the fp16 id leaves its output's
    bound for the next one (pooling and up-sampling hand the word on): only the network input is ever measured, and the chain
    matches float64 torch.  Recording a gradient never uses the id (its launches have no backward bookkeeping)."""
    torch.manual_seed(21)
    net = _block(32, 64, 48).cuda().eval()
    with torch.no_grad():
        net[3].running_mean.uniform_(-0.2, 0.2); net[3].running_var.uniform_(0.5, 2.0)
    x = torch.randn(2, 32, 32, 96, device="cuda")
    ref = _ref_of(net)(x.double().cpu())
    calls = []
    real = HF.measured_amax_word

    def counting(t):
        calls.append(HF.amax_word_of(t) is None)
        return real(t)
    HF.set_algorithm(HF.ALGO_MFMA_F16X3)
    HF.measured_amax_word = counting
    try:
        with torch.no_grad():
            out = net(x)
    finally:
        HF.measured_amax_word = real
    assert calls == [True, False, False, False]               # four fp16 launches; only the network input was measured
    _close(out, ref)
    assert HF.amax_word_of(out) is not None and HF.amax_word_of(out).max().item() == out.abs().max().item()
    y = net(x.clone().requires_grad_())                       # forced id, recording: fp16 launches as well since round 5 (bounds left behind) ...
    assert HF.amax_word_of(y) is not None
    _close(y, ref)
    HF._AUTO_F16_TRAIN = False                                # ... X6 launches, no bounds, with the knob off (the rounds 3-4 behaviour)
    try:
        y = net(x.clone().requires_grad_())
    finally:
        HF._AUTO_F16_TRAIN = True
    assert HF.amax_word_of(y) is None
    _close(y, ref)


def test_auto_runs_inference_layers_under_f16x3_where_x6_would_run():
    """hipnn's ALGO_AUTO: a layer large enough for the split kernels runs under the fp16 id when nothing is recorded (bit-equal to
    the forced id) and, since round 5, when a gradient is recorded too; under X6 when SSTEM_CONV_AUTO_F16X3 (or, for recorded launches,
    SSTEM_CONV_AUTO_F16_TRAIN) is off (bit-equal to forced X6)."""
    torch.manual_seed(24)
    x = torch.randn(4, 64, 128, 128, device="cuda"); w = torch.randn(64, 64, 3, 3, device="cuda") * 0.1; b = torch.randn(64, device="cuda")
    assert HF._AUTO_F16 and HF._auto_algo(4, 64, 128, 128, 64) == HF.ALGO_MFMA_BF16X6
    forced = {}
    for algo in (HF.ALGO_MFMA_F16X3, HF.ALGO_MFMA_BF16X6):
        HF.set_algorithm(algo)
        forced[algo] = HF.conv2d_fused(x, w, b, None, None, HF.ACT_RELU, 0.0)
    HF.set_algorithm(HF.ALGO_AUTO)
    assert not torch.equal(forced[HF.ALGO_MFMA_F16X3], forced[HF.ALGO_MFMA_BF16X6])
    assert torch.equal(HF.conv2d_fused(x, w, b, None, None, HF.ACT_RELU, 0.0), forced[HF.ALGO_MFMA_F16X3])
    # a recorded launch: the fp16 id too since round 5 (the mask it writes changes no bit), X6 with SSTEM_CONV_AUTO_F16_TRAIN off
    assert torch.equal(HF.conv2d_fused(x.clone().requires_grad_(), w, b, None, None, HF.ACT_RELU, 0.0).detach(), forced[HF.ALGO_MFMA_F16X3])
    HF._AUTO_F16_TRAIN = False
    try:
        assert torch.equal(HF.conv2d_fused(x.clone().requires_grad_(), w, b, None, None, HF.ACT_RELU, 0.0).detach(), forced[HF.ALGO_MFMA_BF16X6])
    finally:
        HF._AUTO_F16_TRAIN = True
    HF._AUTO_F16 = False
    try:
        assert torch.equal(HF.conv2d_fused(x, w, b, None, None, HF.ACT_RELU, 0.0), forced[HF.ALGO_MFMA_BF16X6])
    finally:
        HF._AUTO_F16 = True


def test_a_tensor_edited_in_place_is_measured_again():
    """The bound rides on the tensor together with its version counter: after an in-place edit the next layer must not trust it
    (x * 1000 with the old bound would overflow fp16)."""
    HF.set_algorithm(HF.ALGO_MFMA_F16X3)
    torch.manual_seed(22)
    w1 = torch.randn(32, 16, 3, 3, device="cuda") * 0.2; w2 = torch.randn(32, 32, 3, 3, device="cuda") * 0.2
    x = torch.randn(1, 16, 16, 32, device="cuda")
    y = HF.conv2d_fused(x, w1)
    assert HF.amax_word_of(y) is not None
    y.mul_(1000.0)
    assert HF.amax_word_of(y) is None
    out = HF.conv2d_fused(y, w2)
    assert torch.isfinite(out).all()
    _close(out, F.conv2d(y.double().cpu(), w2.double().cpu(), padding=1))


def test_f16x3_in_a_replayed_graph_starts_from_fresh_bounds():
    """A captured inference body: the amax words its launches use are allocated (and zeroed) inside the capture, so every replay
    starts from zero bounds -- a replay on SMALLER inputs than the previous one must give the eager result of those inputs bit for bit
    (a bound left over from the previous replay would pick another scale)."""
    import train_utils
    torch.manual_seed(23)
    net = _block(32, 64, 48).cuda().eval()
    xbuf = torch.randn(2, 32, 32, 64, device="cuda") * 50.0
    holder = {}

    def body():
        with torch.no_grad():
            holder["out"] = net(xbuf)
    g = train_utils.GraphedCallable(body, modules=[net])
    g(); torch.cuda.synchronize()
    big = holder["out"].clone()
    with torch.no_grad():
        assert torch.equal(big, net(xbuf.clone()))
    xbuf.mul_(1e-4)                                           # the graph reads the same buffer: much smaller values now
    g(); torch.cuda.synchronize()
    with torch.no_grad():
        assert torch.equal(holder["out"], net(xbuf.clone()))


@pytest.mark.parametrize("algo", [HF.ALGO_MFMA_F16X3, HF.ALGO_MFMA_BF16X6, HF.ALGO_AUTO])
@pytest.mark.parametrize("shape", [(2, 51, 40, 96, 51), (1, 64, 16, 64, 51), (1, 20, 9, 33, 7), (3, 32, 24, 16, 40), (1, 51, 64, 256, 51), (8, 51, 128, 160, 51),
                                   (64, 32, 128, 16, 40), (8, 20, 131, 200, 7)])
def test_row_segment_output_layout_is_the_same_values_reordered(shape, algo):
    """SSTEM_LAYOUT_ROW_SEGMENTS (what an IFNet kernel head stores for the fused sepconv apply): [N, H, ceil(W/64), Cout, 64] holds
    exactly the NCHW result of the same launch -- whole and ragged tiles, widths that end inside a 64-column segment, the 16-wide
    tile form, ragged channel blocks; the bound left behind is the same."""
    HF.set_algorithm(algo)
    N, Cin, H, W, Cout = shape
    g = torch.Generator().manual_seed(31)
    x = torch.randn(N, Cin, H, W, generator=g).cuda(); w = (torch.randn(Cout, Cin, 3, 3, generator=g) * 0.2).cuda()
    b = torch.randn(Cout, generator=g).cuda()
    conv = nn.Conv2d(Cin, Cout, 3, padding=1).cuda().requires_grad_(False)       # an inference layer
    with torch.no_grad():
        conv.weight.copy_(w); conv.bias.copy_(b)
    if not HF.blocked_store_ok(x, conv):                      # small grids: AUTO keeps them on the fp32 kernel, the split ids split them over K
        assert N * H * W < 64 * 128 * 16
        return
    ref = HF.conv2d_fused(x, w, b, None, None, HF.ACT_RELU, 0.0)
    blk = HF.conv2d_fused(x, w, b, None, None, HF.ACT_RELU, 0.0, out_blocked=True)
    TX = (W + 63) // 64
    assert tuple(blk.shape) == (N, H, TX, Cout, 64)
    back = blk.permute(0, 3, 1, 2, 4).reshape(N, Cout, H, TX * 64)[..., :W]
    seq = FusedSequential(nn.Conv2d(Cin, Cin, 3, padding=1), nn.ReLU(), conv).cuda().eval()
    with torch.no_grad():
        a = seq(x, out_blocked=True); c = seq(x)
    assert a.dim() == 5
    a = a.permute(0, 3, 1, 2, 4).reshape(N, Cout, H, TX * 64)[..., :W]
    assert torch.equal(back, ref) and torch.equal(a, c)     # (blocked_store_ok only grants launches the NCHW form would not split over K)
    assert torch.equal(HF.amax_word_of(blk).max(), HF.amax_word_of(ref).max())


def test_sp_unet_inference_keeps_its_bounds_through_in_place_skips_and_concatenations():
    """networks.UNet at inference stores every encoder output inside the tensor its decoder level concatenates (out=) and the
    up-sampled half beside it: the bounds must survive both (autograd hands back an ALIAS of an `out=` argument -- the tag is carried
    over; the concatenation's bound is the slot-wise maximum of its halves).  Only the output of the first layer (1 -> 64 channels,
    fp32 MFMA kernel: no bound) is ever measured."""
    import networks
    torch.manual_seed(0)
    net = networks.UNet(1, 1).cuda().eval()
    x = torch.rand(1, 1, 512, 512, device="cuda")
    measured = []
    real = HF.measured_amax_word

    def counting(t):
        if HF.amax_word_of(t) is None:
            measured.append(tuple(t.shape))
        return real(t)
    HF.measured_amax_word = counting
    try:
        with torch.no_grad():
            net(x)
    finally:
        HF.measured_amax_word = real
    assert measured == [(1, 64, 512, 512)], measured


# ---- round 4: the tile-walking stream (conv3x3_split_mfma<..., DEEP>) ---------------------------------------------------------------
@pytest.mark.parametrize("shape", [(1, 32, 64, 64, 32), (2, 32, 40, 96, 32), (1, 64, 72, 64, 32), (1, 32, 64, 32, 20), (2, 128, 24, 64, 8),
                                   (1, 48, 64, 64, 32), (2, 6, 48, 64, 32), (1, 6, 72, 32, 6), (1, 16, 40, 64, 24)])      # odd chunk counts
@pytest.mark.parametrize("walk", [2, 4, 3])
def test_f16x3_tile_walking_stream_equals_the_per_tile_kernel_bit_for_bit(shape, walk):
    """The 32-output-channel block instance walks several tiles per workgroup with its loads two stream steps ahead (DEEP): the same
    MFMAs on the same fragments in the same order per tile as the per-tile kernel => the same bits, whatever the number of tiles walked
    (ragged last group included), with the epilogue's variants (residual, row-segment store) and the output's bound."""
    import os
    N, Cin, H, W, Cout = shape
    torch.manual_seed(7)
    x = torch.randn(N, Cin, H, W, device="cuda"); w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05; b = torch.randn(Cout, device="cuda")
    res = torch.randn(N, Cout, H, W, device="cuda")
    keep = {k: os.environ.get(k) for k in ("SSTEM_SPLIT_WALK", "SSTEM_SPLIT_WALK_MIN_WGS")}

    def run(walk_value):
        os.environ["SSTEM_SPLIT_WALK"] = str(walk_value); os.environ["SSTEM_SPLIT_WALK_MIN_WGS"] = "1"
        with HF.algorithm(HF.ALGO_MFMA_F16X3), torch.no_grad():
            a = HF.conv2d_fused(x, w, b, None, None, HF.ACT_RELU, 0.0)
            c = HF.conv2d_fused(x, w, b, None, None, HF.ACT_LEAKY, 0.2, residual=res, res_scale=0.5)
            d = HF.conv2d_fused(x, w, b, None, None, HF.ACT_RELU, 0.0, out_blocked=True)       # the row-segment store
            bound = float(HF.amax_word_of(a).max())
        return a, (c, d), bound
    try:
        a0, (c0, d0), m0 = run(0)
        a1, (c1, d1), m1 = run(walk)
    finally:
        for k, v in keep.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    assert torch.equal(a0, a1) and torch.equal(c0, c1)
    TX = (W + 63) // 64                                      # (columns behind W in the last segment are padding: never written)
    v0, v1 = (d.permute(0, 3, 1, 2, 4).reshape(N, Cout, H, TX * 64)[..., :W] for d in (d0, d1))
    assert torch.equal(v0, v1)                               # (a blocked store is never split over K, the NCHW launch of a small grid is:
    assert (v1 - a1).abs().max().item() <= 2e-6 * a1.abs().max().item()       #  same values up to the order of the slice sums)
    assert m0 == m1 == float(a0.abs().max())
    ref = torch.relu(torch.nn.functional.conv2d(x.double(), w.double(), b.double(), padding=1))
    assert (a1.double() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()


# ---- round 4: ConvTranspose2d(k3, s2, p1, op1) as its sub-pixel form on the fp16 two-piece id ------------------------------------------
@pytest.mark.parametrize("shape", [(2, 64, 16, 32, 32), (1, 128, 24, 64, 64), (2, 32, 40, 36, 32), (1, 256, 8, 32, 128), (1, 16, 9, 20, 96)])
def test_conv_transpose_subpixel_form_matches_float64_torch(shape):
    """SSTEM_LAYOUT_CONVT_PARITY (include/sstem_conv.h): the transposed convolution of the decoder blocks (model_fusionnet.py:21-27,
    model_unet.py:32,70) as a 2 x 2 convolution with 4 C parity-major channels and a pixel-shuffle store, with folded BatchNorm,
    activation and the additive skip `(deconv + skip) / 2` (model_fusionnet.py:129-138) in the store -- against float64 torch at the
    fp32 ids' tolerance, and next to the exact-fp32 native kernel; ragged sizes take the generic store path."""
    N, Cin, H, W, C = shape
    torch.manual_seed(11)
    x = torch.randn(N, Cin, H, W, device="cuda")
    m = nn.ConvTranspose2d(Cin, C, 3, stride=2, padding=1, output_padding=1).cuda().requires_grad_(False)
    scale = torch.rand(C, device="cuda") + 0.5; shift = torch.randn(C, device="cuda") * 0.1
    res = torch.randn(N, C, 2 * H, 2 * W, device="cuda")
    ref = F.conv_transpose2d(x.double(), m.weight.double(), m.bias.double(), stride=2, padding=1, output_padding=1)
    ref = F.leaky_relu(ref * scale.double()[None, :, None, None] + shift.double()[None, :, None, None], 0.2)
    ref = (ref + res.double()) * 0.5
    with HF.algorithm(HF.ALGO_MFMA_F16X3), torch.no_grad():
        assert HF._convT_subpixel_ok(x, m.weight, m, False, None)
        got = HF.conv_transpose3x3s2_fused(x, m.weight, m.bias, scale, shift, HF.ACT_LEAKY, 0.2, owner=m, residual=res, res_scale=0.5)
        again = HF.conv_transpose3x3s2_fused(x, m.weight, m.bias, scale, shift, HF.ACT_LEAKY, 0.2, owner=m, residual=res, res_scale=0.5)
        plain = HF.conv_transpose3x3s2_fused(x, m.weight, m.bias, None, None, HF.ACT_NONE, 0.0, owner=m)
    with HF.algorithm(HF.ALGO_MFMA), torch.no_grad():
        native = HF.conv_transpose3x3s2_fused(x, m.weight, m.bias, scale, shift, HF.ACT_LEAKY, 0.2, owner=m, residual=res, res_scale=0.5)
    assert got.shape == ref.shape and torch.equal(got, again)
    tol = 2e-5 * ref.abs().max().item()
    assert (got.double() - ref).abs().max().item() <= tol and (native.double() - ref).abs().max().item() <= tol
    ref0 = F.conv_transpose2d(x.double(), m.weight.double(), m.bias.double(), stride=2, padding=1, output_padding=1)
    assert (plain.double() - ref0).abs().max().item() <= 2e-5 * ref0.abs().max().item()
    assert float(HF.amax_word_of(got).max()) == float(got.abs().max())          # the bound the next layer scales by
    # new weights are picked up (the sub-pixel image is rebuilt from the module's weights)
    with torch.no_grad():
        m.weight.mul_(0.5)
    with HF.algorithm(HF.ALGO_MFMA_F16X3), torch.no_grad():
        half = HF.conv_transpose3x3s2_fused(x, m.weight, None, None, None, HF.ACT_NONE, 0.0, owner=m)
    ref_h = F.conv_transpose2d(x.double(), m.weight.double(), None, stride=2, padding=1, output_padding=1)
    assert (half.double() - ref_h).abs().max().item() <= 2e-5 * ref_h.abs().max().item()


# ---- round 4: the 2 x 2 pooling behind a layer stored by the layer's own launch ---------------------------------------------------------
@pytest.mark.parametrize("shape", [(2, 32, 64, 64, 32), (1, 6, 48, 96, 32), (2, 64, 40, 64, 64), (1, 128, 16, 32, 128), (1, 48, 24, 64, 20),
                                   (4, 32, 128, 128, 64), (2, 48, 256, 96, 128), (8, 6, 64, 128, 64)])      # four MFMA rows per wave, granted
@pytest.mark.parametrize("kind", ["max", "avg"])
def test_f16x3_pooled_copy_equals_the_pooling_kernel_bit_for_bit(shape, kind):
    """sstem_conv3x3_forward_scaled_strided_f32(pooled_output): the nn.MaxPool2d(2) / nn.AvgPool2d(2) that follows a block in the
    reference's networks (model_interp.py:60-70, model_fusionnet.py:116-123, model_unet.py:78-84) is stored by the block's last launch
    -- same output, and a pooled tensor equal, bit for bit, to what the pooling kernel (and torch) make of that output; also together
    with a store into a channel block of a larger tensor, on the tile-walking and the per-tile instances."""
    N, Cin, H, W, Cout = shape
    torch.manual_seed(13)
    x = torch.randn(N, Cin, H, W, device="cuda")
    conv = nn.Conv2d(Cin, Cout, 3, padding=1).cuda().requires_grad_(False)
    pool = nn.MaxPool2d(2) if kind == "max" else nn.AvgPool2d((2, 2), (2, 2))
    seq = FusedSequential(conv, nn.LeakyReLU(0.2)).cuda().eval()
    with HF.algorithm(HF.ALGO_MFMA_F16X3), torch.no_grad():
        # (granted unless the plain launch would be split over K -- the small grids here with 64 and more input channels: a pooled
        #  store never is, and the two spellings must give the same bits; then the pooling kernel answers the request)
        granted = HF.pooled_store_ok(x, conv, pool)
        assert granted == ((HF.POOL_MAX if kind == "max" else HF.POOL_AVG) if Cin < 64 else 0)
        plain = seq(x)
        y, p = seq(x, pool=pool)
        big = torch.zeros(N, Cout + 5, H, W, device="cuda")
        y2, p2 = seq(x, out=big[:, 5:], pool=pool)
        standalone = HF.pool_module(pool, plain)
    assert torch.equal(y, plain) and torch.equal(y2, plain) and torch.equal(big[:, 5:], plain) and float(big[:, :5].abs().max()) == 0.0
    assert torch.equal(p, standalone) and torch.equal(p2, standalone) and torch.equal(p, pool(plain))
    assert float(HF.amax_word_of(p).max()) == float(plain.abs().max())
    # the pooled copy alone (the IFNets' first block: nobody reads the full-resolution result): no full-resolution tensor, same pooled bits
    with HF.algorithm(HF.ALGO_MFMA_F16X3), torch.no_grad():
        y3, p3 = seq(x, pool=pool, pool_only=True)
    assert (y3 is None) == bool(granted) and torch.equal(p3, standalone)
    assert (not granted) or float(HF.amax_word_of(p3).max()) == float(plain.abs().max())
    # ragged sizes: the request is answered by the pooling kernel (same values)
    xr = torch.randn(1, Cin, 20, 44, device="cuda")
    with HF.algorithm(HF.ALGO_MFMA_F16X3), torch.no_grad():
        assert HF.pooled_store_ok(xr, conv, pool) == 0
        yr, pr = seq(xr, pool=pool)
    assert torch.equal(pr, pool(yr))
