"""GPU parity tests of the split-bf16 convolution ids (csrc/conv_split_kernels.hip): fp32 operands split into three
(ALGO_MFMA_BF16X6) or two (ALGO_MFMA_BF16X3) bf16 pieces whose exact products are summed in fp32 on the bf16 matrix cores.
Reference: float64 PyTorch of the same op on the CPU, identical inputs.  Tolerance: the fp32 kernels' own bound,
max|a-ref| <= 2e-5 * max|ref| (+1e-6) (tests/test_conv_gpu.py) -- unchanged for both ids; on top of that the X6 id must be as
close to float64 as the fp32 MFMA kernel is (its products are fp32 products to 2^-26)."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

import hipnn.functional as HF
import sstem_native
from hipnn import FusedSequential

pytestmark = pytest.mark.gpu
SPLIT = [HF.ALGO_MFMA_BF16X6, HF.ALGO_MFMA_BF16X3]


@pytest.fixture(autouse=True)
def _reset_algo():
    yield
    HF.set_algorithm(HF.ALGO_AUTO)


def _err(a, ref):
    a = a.detach().cpu().double(); ref = ref.detach().cpu().double()
    return (a - ref).abs().max().item(), ref.abs().max().item() + 1e-12


def _close(a, ref, rel=2e-5):
    err, scale = _err(a, ref)
    assert err <= rel * scale + 1e-6, "max err %.3e vs scale %.3e" % (err, scale)


def _act_ref(y, act, slope):
    if act == HF.ACT_RELU:
        return F.relu(y)
    if act == HF.ACT_LEAKY:
        return F.leaky_relu(y, slope)
    return y


# (N, Cin, H, W, Cout): the fp32 tests' shapes (ragged sizes, 6 / 51 / 1 / 2 channels, several chunks and channel blocks, images
# smaller than a tile, W % 4 != 0 -> dword staging) + two with more than one 16-channel chunk per K slice and 64-channel blocks
SHAPES = [(1, 8, 8, 32, 32), (2, 6, 13, 37, 6), (1, 51, 9, 40, 51), (1, 64, 16, 33, 128), (2, 3, 5, 7, 1),
          (1, 130, 4, 4, 70), (1, 1, 1, 1, 2), (2, 40, 24, 64, 70), (1, 96, 16, 36, 64),
          # maps up to 16 pixels wide with W % 4 == 0: the 16 x 16 tiles (two image rows per MFMA row), whole and ragged, both channel blocks
          (2, 64, 16, 16, 96), (1, 32, 24, 12, 32), (3, 48, 8, 8, 40), (1, 20, 33, 16, 70),
          # a ragged last chunk of 1..4 channels behind whole ones: packed and staged by tap row under the three-piece id (16-byte and dword
          # staging, both tile widths, both channel blocks; 51 -> 51 are the IFNet's kernel heads)
          (1, 51, 20, 64, 51), (2, 19, 16, 16, 40), (1, 36, 9, 37, 20), (1, 33, 12, 64, 70), (2, 115, 8, 32, 24)]


@pytest.mark.parametrize("algo", SPLIT)
@pytest.mark.parametrize("shape", SHAPES)
def test_split_conv3x3_forward_fused(shape, algo):
    HF.set_algorithm(algo)
    N, Cin, H, W, Cout = shape
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, Cin, H, W, generator=g); w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.2
    b = torch.randn(Cout, generator=g); sc = torch.rand(Cout, generator=g) + 0.5; sh = torch.randn(Cout, generator=g)
    for act, slope in ((HF.ACT_NONE, 0.0), (HF.ACT_RELU, 0.0), (HF.ACT_LEAKY, 0.2)):
        out = HF.conv2d_fused(x.cuda(), w.cuda(), b.cuda(), sc.cuda(), sh.cuda(), act, slope)
        ref = F.conv2d(x.double(), w.double(), b.double(), padding=1) * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)
        _close(out, _act_ref(ref, act, slope))
    _close(HF.conv2d_fused(x.cuda(), w.cuda()), F.conv2d(x.double(), w.double(), padding=1))


@pytest.mark.parametrize("algo", SPLIT)
@pytest.mark.parametrize("shape", [(2, 6, 13, 37, 10), (1, 40, 8, 8, 33), (1, 3, 3, 3, 3), (2, 70, 16, 32, 64), (3, 130, 9, 64, 70), (2, 64, 40, 36, 64),
                                   (2, 96, 16, 16, 64), (1, 40, 20, 12, 24)])
def test_split_conv3x3_backward(shape, algo):
    """Data gradient (transposed + flipped packing of the forward kernel) and weight + bias gradient (conv3x3_wgrad_split_mfma)
    under the split ids."""
    HF.set_algorithm(algo)
    N, Cin, H, W, Cout = shape
    g = torch.Generator().manual_seed(4)
    x = torch.randn(N, Cin, H, W, generator=g); w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.3
    b = torch.randn(Cout, generator=g); go = torch.randn(N, Cout, H, W, generator=g)
    for act, slope in ((HF.ACT_LEAKY, 0.2), (HF.ACT_RELU, 0.0), (HF.ACT_NONE, 0.0)):
        xg, wg, bg = x.cuda().requires_grad_(), w.cuda().requires_grad_(), b.cuda().requires_grad_()
        HF.conv2d_fused(xg, wg, bg, None, None, act, slope).backward(go.cuda())
        xr, wr, br = x.double().requires_grad_(), w.double().requires_grad_(), b.double().requires_grad_()
        _act_ref(F.conv2d(xr, wr, br, padding=1), act, slope).backward(go.double())
        _close(xg.grad, xr.grad); _close(wg.grad, wr.grad); _close(bg.grad, br.grad)


@pytest.mark.parametrize("algo", SPLIT)
def test_split_one_hot_weights_copy_the_input_bit_for_bit(algo):
    """A weight tensor that selects one input channel and one tap: h + m (+ l) of every input value meet a weight of exactly 1, so
    the X6 id must return the shifted input bit for bit (pins indexing, padding, packing and the exactness of the split); the X3 id
    returns h + m = the input to 2^-17."""
    HF.set_algorithm(algo)
    torch.manual_seed(7)
    N, Cin, H, W, Cout = 2, 20, 19, 40, 35
    x = torch.randn(N, Cin, H, W, device="cuda") * 3
    w = torch.zeros(Cout, Cin, 3, 3, device="cuda")
    for co in range(Cout):
        w[co, (co * 7) % Cin, co % 3, (co // 3) % 3] = 1.0
    out = HF.conv2d_fused(x, w)
    ref = F.conv2d(x.double().cpu(), w.double().cpu(), padding=1)
    if algo == HF.ALGO_MFMA_BF16X6:
        assert torch.equal(out.cpu().double(), ref)
    else:
        assert (out.cpu().double() - ref).abs().max().item() <= 2.0 ** -16 * ref.abs().max().item()


def test_split_x6_is_as_close_to_float64_as_the_fp32_mfma_kernel():
    """Long sums (K = 9 * 256) of same-sign products, where every dropped term would add up: max and rms error of the X6 id against
    float64 within 1.25 x the fp32 MFMA kernel's own (both are 'exact products, fp32 accumulation in some order')."""
    torch.manual_seed(8)
    N, Cin, H, W, Cout = 1, 256, 24, 64, 64
    x = torch.rand(N, Cin, H, W, device="cuda") + 0.5
    w = torch.rand(Cout, Cin, 3, 3, device="cuda") + 0.1
    ref = F.conv2d(x.double().cpu(), w.double().cpu(), padding=1)
    res = {}
    for algo in (HF.ALGO_MFMA, HF.ALGO_MFMA_BF16X6, HF.ALGO_MFMA_BF16X3):
        HF.set_algorithm(algo)
        d = HF.conv2d_fused(x, w).cpu().double() - ref
        res[algo] = (d.abs().max().item(), d.pow(2).mean().sqrt().item())
    scale = ref.abs().max().item()
    print("errors / max|ref| (max, rms): fp32 MFMA %.2e %.2e   X6 %.2e %.2e   X3 %.2e %.2e" % tuple(
        v / scale for algo in (HF.ALGO_MFMA, HF.ALGO_MFMA_BF16X6, HF.ALGO_MFMA_BF16X3) for v in res[algo]))
    assert res[HF.ALGO_MFMA_BF16X6][0] <= 1.25 * res[HF.ALGO_MFMA][0] + 1e-7 * scale
    assert res[HF.ALGO_MFMA_BF16X6][1] <= 1.25 * res[HF.ALGO_MFMA][1] + 1e-8 * scale
    assert res[HF.ALGO_MFMA_BF16X3][0] <= 2e-5 * scale


# split over K on small grids: (N, Cin, H, W, Cout)
@pytest.mark.parametrize("algo", SPLIT)
@pytest.mark.parametrize("shape", [(2, 512, 16, 16, 512), (2, 128, 32, 32, 256), (1, 64, 20, 37, 32), (2, 256, 32, 64, 64), (2, 512, 8, 8, 256),
                                   (2, 115, 16, 16, 64), (1, 131, 32, 32, 32)])       # the last K slice ends in a tap-row chunk
def test_split_conv3x3_split_k_matches_unsplit_and_fp64(shape, algo):
    lib = sstem_native.load_library()
    N, Cin, H, W, Cout = shape
    torch.manual_seed(9)
    x = torch.randn(N, Cin, H, W, device="cuda"); w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05
    b = torch.randn(Cout, device="cuda")
    full = lib.sstem_conv3x3_forward_workspace_floats_algo(N, Cin, H, W, Cout, algo)
    minimum = lib.sstem_conv3x3_packed_floats(Cin, Cout, algo)
    outs = []
    for n_ws in (full, minimum):
        ws = torch.empty(n_ws, device="cuda"); out = torch.empty(N, Cout, H, W, device="cuda")
        rc = lib.sstem_conv2d_forward_f32(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, None, out.data_ptr(), ws.data_ptr(), n_ws,
                                          N, Cin, H, W, Cout, 3, 3, 1, 1, 0, HF.ACT_LEAKY, 0.2, None, algo)
        assert rc == 0
        outs.append(out)
    ref = F.leaky_relu(F.conv2d(x.double().cpu(), w.double().cpu(), b.double().cpu(), padding=1), 0.2)
    _close(outs[0], ref); _close(outs[1], ref)
    if full > minimum:
        _close(outs[0], outs[1], 1e-5)          # another summation order (K slices), same arithmetic


@pytest.mark.parametrize("algo", SPLIT)
def test_split_residual_in_the_store_and_fused_sequential(algo):
    HF.set_algorithm(algo)
    torch.manual_seed(10)
    x = torch.randn(2, 24, 16, 32, device="cuda"); res = torch.randn(2, 40, 16, 32, device="cuda")
    conv = nn.Conv2d(24, 40, 3, padding=1).cuda()
    with torch.no_grad():
        assert HF.residual_fusable(x, conv, res)
        out = HF.conv2d_fused(x, conv.weight, conv.bias, None, None, HF.ACT_RELU, 0.0, residual=res, res_scale=0.5)
        ref = (F.relu(F.conv2d(x.double().cpu(), conv.weight.double().cpu(), conv.bias.double().cpu(), padding=1)) + res.double().cpu()) * 0.5
    _close(out, ref)
    # a Conv + BN(eval) + LeakyReLU + Conv + ReLU block through FusedSequential (folded affine, cached packed weights)
    mods = [nn.Conv2d(5, 12, 3, padding=1), nn.BatchNorm2d(12), nn.LeakyReLU(0.2), nn.Conv2d(12, 7, 3, padding=1), nn.ReLU()]
    mods[1].running_mean.uniform_(-0.3, 0.3); mods[1].running_var.uniform_(0.5, 1.5)
    ref_net = nn.Sequential(*mods).eval()
    xi = torch.randn(3, 5, 10, 12)
    with torch.no_grad():
        want = ref_net.double()(xi.double())
        ref_net.float()
        import copy
        fused = FusedSequential(*copy.deepcopy(mods)).eval().cuda()
        got1 = fused(xi.cuda()); got2 = fused(xi.cuda())
    _close(got1, want, 1e-4)
    assert torch.equal(got1, got2)


@pytest.mark.parametrize("algo", SPLIT)
def test_split_weight_gradient_long_sums_and_accumulate(algo):
    """Many pixel tiles per workgroup and several K slices (N x H x W = 4 x 64 x 96), same-sign data: the X6 weight gradient within
    1.25 x the fp32 MFMA kernel's distance from float64, both ids inside the fp32 tolerance; accumulate adds onto what is there."""
    lib = sstem_native.load_library()
    torch.manual_seed(12)
    N, Cin, H, W, Cout = 4, 48, 64, 96, 80
    x = torch.rand(N, Cin, H, W, device="cuda") + 0.5; g = torch.rand(N, Cout, H, W, device="cuda") + 0.1
    wref = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double().cpu(), wref, padding=1).backward(g.double().cpu())
    ref = wref.grad; bref = g.double().cpu().sum((0, 2, 3))

    def run(a, accumulate=0, gw=None, gb=None):
        n = lib.sstem_conv3x3_wgrad_workspace_floats_algo(N, Cin, H, W, Cout, a)
        ws = torch.empty(max(n, 1), device="cuda")
        gw = torch.empty(Cout, Cin, 3, 3, device="cuda") if gw is None else gw
        gb = torch.empty(Cout, device="cuda") if gb is None else gb
        rc = lib.sstem_conv2d_backward_weight_bias_ex_f32(x.data_ptr(), g.data_ptr(), gw.data_ptr(), gb.data_ptr(), ws.data_ptr(), n,
                                                          N, Cin, H, W, Cout, 3, 3, 1, 1, accumulate, None, a)
        assert rc == 0
        torch.cuda.synchronize()
        return gw, gb
    gw32, _ = run(HF.ALGO_MFMA)
    gw, gb = run(algo)
    _close(gw, ref); _close(gb, bref)
    e32, scale = _err(gw32, ref); e, _ = _err(gw, ref)
    print("wgrad errors / max|ref|: fp32 MFMA %.2e, id %d %.2e" % (e32 / scale, algo, e / scale))
    if algo == HF.ALGO_MFMA_BF16X6:
        assert e <= 1.25 * e32 + 1e-7 * scale
    gw2, gb2 = run(algo, 1, gw.clone(), gb.clone())
    _close(gw2, 2 * ref); _close(gb2, 2 * bref)
    again, _ = run(algo)
    assert torch.equal(again, gw)                       # fixed-order sums: bit-reproducible


# ---- the ReLU mask inside the launches (sstem_conv3x3_forward_masked_f32 / sstem_conv3x3_backward_weight_masked_f32) ----------------
# (N, Cin, H, W, Cout): 16-byte staging, dword staging (W % 4 != 0), a ragged channel count, a launch split over K
MASK_SHAPES = [(2, 40, 24, 64, 70), (1, 24, 9, 37, 33), (2, 128, 32, 32, 256), (3, 64, 13, 36, 64), (2, 64, 16, 16, 96), (1, 32, 24, 12, 32),
               (2, 51, 24, 64, 51), (1, 35, 16, 16, 20)]      # tap-row last chunk, forward (Cin) and data gradient (Cout)


@pytest.mark.parametrize("algo", SPLIT)
@pytest.mark.parametrize("shape", MASK_SHAPES)
def test_masked_launches_equal_the_separate_passes_bit_for_bit(shape, algo):
    lib = sstem_native.load_library()
    N, Cin, H, W, Cout = shape
    torch.manual_seed(21)
    x = torch.randn(N, Cin, H, W, device="cuda"); w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.1
    b = torch.randn(Cout, device="cuda"); g = torch.randn(N, Cout, H, W, device="cuda")

    def fwd(inp, weight, bias, cout, flags, act, in_mask=None, want_mask=False):
        n, cin = inp.shape[0], inp.shape[1]
        ws_n = lib.sstem_conv3x3_forward_workspace_floats_algo(n, cin, H, W, cout, algo)
        ws = torch.empty(ws_n, device="cuda"); out = torch.empty(n, cout, H, W, device="cuda")
        om = torch.zeros(n, cout, H, W, dtype=torch.bool, device="cuda") if want_mask else None
        if in_mask is None and not want_mask:
            rc = lib.sstem_conv2d_forward_f32(inp.data_ptr(), weight.data_ptr(), bias.data_ptr() if bias is not None else None, None, None,
                                              out.data_ptr(), ws.data_ptr(), ws_n, n, cin, H, W, cout, 3, 3, 1, 1, flags, act, 0.0, None, algo)
        else:
            rc = lib.sstem_conv3x3_forward_masked_f32(inp.data_ptr(), in_mask.data_ptr() if in_mask is not None else None, weight.data_ptr(),
                                                      bias.data_ptr() if bias is not None else None, None, None, out.data_ptr(),
                                                      om.data_ptr() if om is not None else None, ws.data_ptr(), ws_n, n, cin, H, W, cout,
                                                      flags, act, 0.0, None, algo)
        assert rc == 0
        torch.cuda.synchronize()
        return out, om
    # forward: same output, mask = (output > 0)
    plain, _ = fwd(x, w, b, Cout, 0, HF.ACT_RELU)
    out, mask = fwd(x, w, b, Cout, 0, HF.ACT_RELU, want_mask=True)
    assert torch.equal(out, plain) and torch.equal(mask, plain > 0)
    # data gradient: the mask applied while staging == the select pass first
    gm = torch.where(mask, g, torch.zeros((), device="cuda"))
    ref_gx, _ = fwd(gm, w, None, Cin, 1, HF.ACT_NONE)
    gx, _ = fwd(g, w, None, Cin, 1, HF.ACT_NONE, in_mask=mask)
    assert torch.equal(gx, ref_gx)
    # weight + bias gradient
    ws_n = lib.sstem_conv3x3_wgrad_workspace_floats_algo(N, Cin, H, W, Cout, algo); ws = torch.empty(ws_n, device="cuda")
    gw0 = torch.empty(Cout, Cin, 3, 3, device="cuda"); gb0 = torch.empty(Cout, device="cuda")
    gw1 = torch.empty_like(gw0); gb1 = torch.empty_like(gb0)
    assert lib.sstem_conv2d_backward_weight_bias_f32(x.data_ptr(), gm.data_ptr(), gw0.data_ptr(), gb0.data_ptr(), ws.data_ptr(), ws_n,
                                                     N, Cin, H, W, Cout, 3, 3, 1, 1, None, algo) == 0
    assert lib.sstem_conv3x3_backward_weight_masked_f32(x.data_ptr(), g.data_ptr(), mask.data_ptr(), gw1.data_ptr(), gb1.data_ptr(),
                                                        ws.data_ptr(), ws_n, N, Cin, H, W, Cout, 0, None, algo) == 0
    torch.cuda.synchronize()
    assert torch.equal(gw1, gw0) and torch.equal(gb1, gb0)


@pytest.mark.parametrize("algo", SPLIT)
def test_mask_fusion_in_a_training_step_changes_nothing(algo, monkeypatch):
    """Conv+ReLU x3 with a skip add in between (two consumers of one activation): loss and every gradient bit-identical with the mask
    inside the launches and as separate compare / select passes."""
    HF.set_algorithm(algo)
    res = []
    for fusion in (True, False):
        monkeypatch.setattr(HF, "_MASK_FUSION", fusion)
        torch.manual_seed(22)
        convs = [nn.Conv2d(16, 48, 3, padding=1), nn.Conv2d(48, 48, 3, padding=1), nn.Conv2d(48, 20, 3, padding=1)]
        net = [FusedSequential(c, nn.ReLU()).cuda() for c in convs]
        x = torch.randn(2, 16, 20, 32).cuda().requires_grad_(True)
        a = net[0](x); bb = net[1](a); out = net[2](a + bb)
        out.square().mean().backward()
        res.append([out.detach(), x.grad] + [p.grad for m in net for p in m.parameters()])
    for u, v in zip(res[0], res[1]):
        assert torch.equal(u, v)
    # and against float64 torch
    torch.manual_seed(22)
    convs = [nn.Conv2d(16, 48, 3, padding=1), nn.Conv2d(48, 48, 3, padding=1), nn.Conv2d(48, 20, 3, padding=1)]
    x = torch.randn(2, 16, 20, 32).double().requires_grad_(True)
    for c in convs:
        c.double()
    a = F.relu(convs[0](x)); bb = F.relu(convs[1](a)); out = F.relu(convs[2](a + bb))
    out.square().mean().backward()
    _close(res[0][0], out); _close(res[0][1], x.grad)
    for got, p in zip(res[0][2:], [p for c in convs for p in c.parameters()]):
        _close(got, p.grad)


@pytest.mark.parametrize("algo", SPLIT)
def test_split_range_and_non_finite_inputs(algo):
    """The split keeps fp32's exponent range (values of 1e30 and 1e-30 convolve like any others: no scaling is involved), and the
    documented deviation is pinned: an infinite input poisons exactly the 3 x 3 neighbourhood that reads it (NaN or inf there --
    the fp32 kernels give inf --, every other output untouched and finite)."""
    HF.set_algorithm(algo)
    torch.manual_seed(41)
    N, Cin, H, W, Cout = 1, 32, 24, 64, 40
    x = torch.randn(N, Cin, H, W, device="cuda"); w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.1
    for scale in (1e30, 1e-30):
        out = HF.conv2d_fused(x * scale, w)
        ref = F.conv2d((x * scale).double().cpu(), w.double().cpu(), padding=1)
        _close(out, ref, 2e-5)
    clean = HF.conv2d_fused(x, w)
    xi = x.clone(); xi[0, 5, 10, 20] = float("inf")
    out = HF.conv2d_fused(xi, w)
    bad = ~torch.isfinite(out)
    hit = torch.zeros_like(bad); hit[:, :, 9:12, 19:22] = True
    assert bad[hit].all() and not bad[~hit].any()
    assert torch.equal(out[~hit], clean[~hit])
    # the same in a channel of a tap-row last chunk (51 = 3 x 16 + 3; a pixel's K vector there spans its right-hand neighbours)
    x51 = torch.randn(1, 51, H, W, device="cuda"); w51 = torch.randn(Cout, 51, 3, 3, device="cuda") * 0.1
    clean = HF.conv2d_fused(x51, w51)
    for ch, col in ((49, 20), (50, 31), (48, 32), (50, 63), (48, 0)):
        xi = x51.clone(); xi[0, ch, 10, col] = float("inf")
        out = HF.conv2d_fused(xi, w51)
        bad = ~torch.isfinite(out)
        hit = torch.zeros_like(bad); hit[:, :, 9:12, max(col - 1, 0):col + 2] = True
        assert bad[hit].all() and not bad[~hit].any(), (ch, col)
        assert torch.equal(out[~hit], clean[~hit])


def _random_shapes(seed, n):
    import random
    rnd = random.Random(seed)
    shapes = []
    for _ in range(n):
        W = rnd.choice([1, 3, 4, 7, 8, 12, 13, 16, 17, 20, 31, 32, 33, 36, 64, 70])
        shapes.append((rnd.randint(1, 3), rnd.randint(1, 80), rnd.randint(1, 40), W, rnd.randint(1, 80)))
    return shapes


@pytest.mark.parametrize("algo", SPLIT)
def test_split_forty_random_shapes_forward_and_gradients(algo):
    """Seeded random shapes over every staging path (16-byte / dword, 32- and 16-wide tiles, ragged channels and edges): output,
    data gradient, weight and bias gradient against float64 torch (LeakyReLU: its mask path is the select pass; the in-launch ReLU
    masks have their own bit-exact tests)."""
    HF.set_algorithm(algo)
    for i, (N, Cin, H, W, Cout) in enumerate(_random_shapes(2026, 40)):
        g = torch.Generator().manual_seed(100 + i)
        x = torch.randn(N, Cin, H, W, generator=g); w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.2
        b = torch.randn(Cout, generator=g); go = torch.randn(N, Cout, H, W, generator=g)
        xg, wg, bg = x.cuda().requires_grad_(), w.cuda().requires_grad_(), b.cuda().requires_grad_()
        out = HF.conv2d_fused(xg, wg, bg)
        out.backward(go.cuda())
        xr, wr, br = x.double().requires_grad_(), w.double().requires_grad_(), b.double().requires_grad_()
        ref = F.conv2d(xr, wr, br, padding=1)
        ref.backward(go.double())
        for name, a, r_ in (("out", out, ref), ("gx", xg.grad, xr.grad), ("gw", wg.grad, wr.grad), ("gb", bg.grad, br.grad)):
            err, scale = _err(a, r_)
            assert err <= 2e-5 * scale + 1e-6, "shape %s %s: max err %.3e vs scale %.3e" % ((N, Cin, H, W, Cout), name, err, scale)
