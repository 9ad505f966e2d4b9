"""GPU tests of the opt-in bf16-operand convolution id (include/sstem_conv.h SSTEM_CONV_MFMA_BF16, BASELINE config 5).

Two references, both plain PyTorch on the CPU in fp64:
  * EXACT-OPERAND reference: the same op on inputs and weights rounded to bf16 first (torch's round-to-nearest-even).  Products
    of two bf16 values are exact in fp32, so the kernel may differ from this only by its fp32 summation order:
    tolerance 2e-5 * max|ref| (the fp32 tests' bound).  This pins indexing, padding, packing and the rounding mode.
  * the un-rounded fp64 op: the bf16 id is allowed 2^-8 relative per operand; for sums of n products of random sign the
    observed error is ~2^-9 * sqrt(n) * typical product, tested as 1.5e-2 * max|ref| -- a sanity bound, not a parity claim.
"""
import pytest
import torch
import torch.nn.functional as F

import hipnn.functional as HF

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _reset_algo():
    yield
    HF.set_algorithm(HF.ALGO_AUTO)


def _r(t):
    return t.bfloat16().double()


def _close(a, ref, rel):
    a = a.detach().cpu().double(); ref = ref.detach().cpu().double()
    scale = ref.abs().max().item() + 1e-12
    err = (a - ref).abs().max().item()
    assert err <= rel * scale + 1e-6, "max err %.3e vs scale %.3e (rel %.1e)" % (err, scale, rel)


# (N, Cin, H, W, Cout): ragged channel counts of the real layers (6, 51, 3, 1, 2), both workgroup shapes (Cout <= 32 / > 32),
# several 16-channel chunks and 64-channel blocks, images smaller than a tile and not multiples of it
SHAPES = [(1, 16, 8, 32, 32), (2, 6, 13, 37, 6), (1, 51, 9, 40, 51), (1, 64, 16, 33, 128), (2, 3, 5, 7, 1),
          (1, 130, 4, 4, 70), (1, 1, 1, 1, 2), (1, 32, 19, 70, 64), (3, 17, 8, 8, 33), (2, 48, 20, 96, 40), (1, 24, 9, 68, 16)]


@pytest.mark.parametrize("shape", SHAPES)
def test_bf16_forward_matches_fp64_of_rounded_operands(shape):
    N, Cin, H, W, Cout = shape
    g = torch.Generator().manual_seed(11)
    x = torch.randn(N, Cin, H, W, generator=g); w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.2
    b = torch.randn(Cout, generator=g); sc = torch.rand(Cout, generator=g) + 0.5; sh = torch.randn(Cout, generator=g)
    HF.set_algorithm(HF.ALGO_MFMA_BF16)
    for act, slope in ((HF.ACT_NONE, 0.0), (HF.ACT_RELU, 0.0), (HF.ACT_LEAKY, 0.2)):
        out = HF.conv2d_fused(x.cuda(), w.cuda(), b.cuda(), sc.cuda(), sh.cuda(), act, slope)
        ref = F.conv2d(_r(x), _r(w), b.double(), padding=1) * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)
        if act == HF.ACT_RELU:
            ref = F.relu(ref)
        elif act == HF.ACT_LEAKY:
            ref = F.leaky_relu(ref, slope)
        _close(out, ref, 2e-5)
    out = HF.conv2d_fused(x.cuda(), w.cuda())
    _close(out, F.conv2d(_r(x), _r(w), padding=1), 2e-5)
    _close(out, F.conv2d(x.double(), w.double(), padding=1), 1.5e-2)


@pytest.mark.parametrize("bf16_wgrad", [True, False])
@pytest.mark.parametrize("shape", [(2, 6, 13, 37, 6), (1, 51, 9, 40, 51), (1, 64, 16, 33, 128), (1, 40, 8, 8, 20), (3, 70, 33, 65, 130),
                                   (2, 3, 1, 1, 2), (2, 70, 10, 64, 130), (1, 16, 7, 36, 16), (2, 32, 5, 4, 8)])
def test_bf16_backward(shape, bf16_wgrad):
    """Backward under the bf16 id: data gradient = the bf16 kernel on (grad_output, W transposed + flipped); weight gradient = the
    bf16 weight-gradient kernel on (input, grad_output), or the fp32 one when switched off; bias gradient always from fp32 values."""
    N, Cin, H, W, Cout = shape
    g = torch.Generator().manual_seed(12)
    x = torch.randn(N, Cin, H, W, generator=g); w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.2
    b = torch.randn(Cout, generator=g); go = torch.randn(N, Cout, H, W, generator=g)
    HF.set_algorithm(HF.ALGO_MFMA_BF16)
    HF.set_bf16_weight_gradient(bf16_wgrad)
    try:
        xc = x.cuda().requires_grad_(True); wc = w.cuda().requires_grad_(True); bc = b.cuda().requires_grad_(True)
        out = HF.conv2d_fused(xc, wc, bc, None, None, HF.ACT_NONE, 0.0)
        out.backward(go.cuda())
    finally:
        HF.set_bf16_weight_gradient(True)
    _close(xc.grad, F.conv_transpose2d(_r(go), _r(w), padding=1), 2e-5)
    xd = x.double().requires_grad_(True); wd = w.double().requires_grad_(True); bd = b.double().requires_grad_(True)
    F.conv2d(xd, wd, bd, padding=1).backward(go.double())
    _close(bc.grad, bd.grad, 2e-5)
    _close(xc.grad, xd.grad, 1.5e-2)
    if bf16_wgrad:
        xr = _r(x).requires_grad_(False); wr = w.double().requires_grad_(True)
        F.conv2d(xr, wr, None, padding=1).backward(_r(go))
        _close(wc.grad, wr.grad, 2e-5)
        _close(wc.grad, wd.grad, 1.5e-2)
    else:
        _close(wc.grad, wd.grad, 2e-5)


@pytest.mark.parametrize("shape", [(8, 64, 128, 128, 64), (4, 128, 64, 64, 128), (8, 64, 128, 128, 51), (2, 32, 256, 256, 32)])
def test_bf16_layer_sized_shapes_against_the_fp32_kernels(shape):
    """Layer-sized problems (many pixel tiles per workgroup, every CU busy) -- the sizes at which a staging race would show: the
    bf16 id against the fp32 MFMA id on the same tensors, forward, data gradient, weight and bias gradient."""
    N, Cin, H, W, Cout = shape
    g = torch.Generator().manual_seed(21)
    x = torch.randn(N, Cin, H, W, generator=g).cuda(); w = (torch.randn(Cout, Cin, 3, 3, generator=g) * 0.1).cuda()
    b = torch.randn(Cout, generator=g).cuda(); go = torch.randn(N, Cout, H, W, generator=g).cuda()
    res = {}
    for algo in (HF.ALGO_MFMA, HF.ALGO_MFMA_BF16):
        HF.set_algorithm(algo)
        xc = x.clone().requires_grad_(True); wc = w.clone().requires_grad_(True); bc = b.clone().requires_grad_(True)
        out = HF.conv2d_fused(xc, wc, bc, None, None, HF.ACT_NONE, 0.0)
        out.backward(go)
        res[algo] = (out.detach(), xc.grad, wc.grad, bc.grad)
    for a, r, what in zip(res[HF.ALGO_MFMA_BF16], res[HF.ALGO_MFMA], ("output", "grad_input", "grad_weight", "grad_bias")):
        assert torch.isfinite(a).all(), what
        _close(a, r, 1.5e-2 if what != "grad_bias" else 2e-5)


def test_bf16_split_k_layer():
    """A deep layer at small batch (grid below two workgroups per CU): K slices + the separate epilogue launch."""
    import sstem_native
    lib = sstem_native.load_library()
    N, Cin, H, W, Cout = 1, 256, 16, 16, 256
    full = int(lib.sstem_conv3x3_forward_workspace_floats_algo(N, Cin, H, W, Cout, HF.ALGO_MFMA_BF16))
    packed = (Cout // 64) * (Cin // 16) * 9 * 64 * 16 // 2
    assert full > packed, "this shape is expected to be split (workspace %d vs packed weights %d floats)" % (full, packed)
    g = torch.Generator().manual_seed(13)
    x = torch.randn(N, Cin, H, W, generator=g); w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05; b = torch.randn(Cout, generator=g)
    HF.set_algorithm(HF.ALGO_MFMA_BF16)
    out = HF.conv2d_fused(x.cuda(), w.cuda(), b.cuda(), None, None, HF.ACT_RELU, 0.0)
    _close(out, F.relu(F.conv2d(_r(x), _r(w), b.double(), padding=1)), 2e-5)


def test_bf16_conv_transpose_block():
    g = torch.Generator().manual_seed(14)
    x = torch.randn(2, 40, 9, 17, generator=g); w = torch.randn(40, 33, 3, 3, generator=g) * 0.2; b = torch.randn(33, generator=g)
    HF.set_algorithm(HF.ALGO_MFMA_BF16)
    out = HF.conv_transpose3x3s2_fused(x.cuda(), w.cuda(), b.cuda(), None, None, HF.ACT_RELU, 0.0)
    ref = F.relu(F.conv_transpose2d(_r(x), _r(w), b.double(), stride=2, padding=1, output_padding=1))
    _close(out, ref, 2e-5)


def test_bf16_is_opt_in_and_scoped():
    assert HF.get_algorithm() == HF.ALGO_AUTO
    with HF.algorithm(HF.ALGO_MFMA_BF16):
        assert HF.get_algorithm() == HF.ALGO_MFMA_BF16
    assert HF.get_algorithm() == HF.ALGO_AUTO
    # AUTO stays bit-identical to the fp32 MFMA id
    g = torch.Generator().manual_seed(15)
    x = torch.randn(1, 20, 12, 40, generator=g).cuda(); w = (torch.randn(24, 20, 3, 3, generator=g) * 0.2).cuda()
    a = HF.conv2d_fused(x, w)
    HF.set_algorithm(HF.ALGO_MFMA)
    assert torch.equal(a, HF.conv2d_fused(x, w))


def test_bf16_whole_ifnet_forward_against_the_fp32_ids():
    """What the opt-in id costs at the output of the whole interpolation network (47 convolutions deep, random orthogonal init,
    two grayscale frames): recorded here as a floor, not as parity -- the fp32 ids remain the parity path (1e-4)."""
    import math
    from model.model_interp import IFNet
    torch.manual_seed(555)
    net = IFNet(51).eval().cuda()
    f = torch.rand(2, 2, 128, 128, device="cuda")
    x = torch.cat((f[:, :1].expand(2, 3, 128, 128), f[:, 1:].expand(2, 3, 128, 128)), 1).contiguous()
    with torch.no_grad():
        ref = net(x)
        with HF.algorithm(HF.ALGO_MFMA_BF16):
            got = net(x)
    assert torch.isfinite(got).all()
    mse = float(((got - ref).double() ** 2).mean())
    peak = float(ref.abs().max())
    psnr = 10.0 * math.log10(peak * peak / max(mse, 1e-30))
    print("IFNet forward, bf16 conv operands vs fp32 ids: PSNR %.1f dB (peak %.3f, max |diff| %.2e)" % (psnr, peak, float((got - ref).abs().max())))
    assert psnr >= 35.0, psnr


@pytest.mark.parametrize("shape", [(2, 24, 19, 40), (1, 24, 64, 96), (3, 6, 8, 32), (1, 24, 5, 4)])
@pytest.mark.parametrize("with_bn", [False, True])
def test_bf16_tensors_between_the_convolutions_of_a_block_change_nothing(shape, with_bn, monkeypatch):
    """Under the bf16 id and no_grad the convolutions inside one FusedSequential hand each other bf16 tensors
    (sstem_conv3x3_forward_bf16io).  The consumer would round the same fp32 values with the same instruction, so the block's
    output must be BIT-IDENTICAL to the fp32-tensor spelling -- on edge tiles, partial channel blocks, folded BatchNorm."""
    from hipnn import FusedSequential
    import torch.nn as nn
    N, Cin, H, W = shape
    torch.manual_seed(61)
    layers = []
    for ci, co in ((Cin, 40), (40, 64), (64, 51)):
        layers.append(nn.Conv2d(ci, co, 3, padding=1))
        if with_bn:
            layers.append(nn.BatchNorm2d(co))
        layers.append(nn.ReLU())
    seq = FusedSequential(*layers).cuda().eval()
    if with_bn:
        for m in seq:
            if isinstance(m, nn.BatchNorm2d):
                m.running_mean.normal_(); m.running_var.uniform_(0.5, 2.0); m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_()
    x = torch.randn(N, Cin, H, W, device="cuda")
    HF.set_algorithm(HF.ALGO_MFMA_BF16)
    with torch.no_grad():
        monkeypatch.setattr(HF, "_BF16_IO", False)
        ref = seq(x)
        monkeypatch.setattr(HF, "_BF16_IO", True)
        calls = []
        orig = HF.conv3x3_bf16io
        monkeypatch.setattr(HF, "conv3x3_bf16io", lambda *a, **k: (calls.append(k.get("out_bf16")), orig(*a, **k))[1])
        got = seq(x)
    assert got.dtype == torch.float32 and torch.equal(got, ref)
    assert calls == [True, True, False], calls          # the two inner hand-overs were bf16, the block's output fp32
    # recording a backward: without BatchNorm the block runs as one autograd function with the same bf16 hand-overs (next test);
    # with a (folded, eval-mode) BatchNorm in it the per-layer path, fp32 tensors everywhere
    calls.clear()
    seq[0].weight.requires_grad_(True)
    y = seq(x)
    assert y.dtype == torch.float32 and calls == ([] if with_bn else [True, True, False]), calls


@pytest.mark.parametrize("shape", [(2, 24, 19, 40), (1, 16, 32, 64), (2, 6, 9, 8)])
def test_conv_chain_with_bf16_inner_tensors_matches_the_per_layer_path(shape, monkeypatch):
    """Training under the bf16 id: a Conv-ReLU-Conv-ReLU-Conv block of a FusedSequential runs as one autograd function with bf16
    tensors between its convolutions.  Every consumer of those tensors rounds them to bf16 anyway, so output and all gradients must
    be BIT-IDENTICAL to the per-layer path (fp32 tensors everywhere)."""
    from hipnn import FusedSequential
    import torch.nn as nn
    N, Cin, H, W = shape
    torch.manual_seed(71)
    seq = FusedSequential(nn.Conv2d(Cin, 40, 3, padding=1), nn.ReLU(), nn.Conv2d(40, 64, 3, padding=1), nn.LeakyReLU(0.2),
                          nn.Conv2d(64, 51, 3, padding=1)).cuda()
    x = torch.randn(N, Cin, H, W, device="cuda")
    go = torch.randn(N, 51, H, W, device="cuda")
    HF.set_algorithm(HF.ALGO_MFMA_BF16)
    res = []
    for io in (False, True):
        monkeypatch.setattr(HF, "_BF16_IO", io)
        for p in seq.parameters():
            p.grad = None
        xc = x.clone().requires_grad_(True)
        y = seq(xc)
        y.backward(go)
        res.append([y.detach(), xc.grad] + [p.grad.clone() for p in seq.parameters()])
        if io:
            assert type(y.grad_fn).__name__ == "_ConvChainBackward", type(y.grad_fn).__name__
    for a, r in zip(res[0], res[1]):
        assert torch.equal(a, r)


def test_whole_ifnet_training_step_is_unchanged_by_the_conv_chains(monkeypatch):
    """End to end: loss and every parameter gradient of one IFNet training step under the bf16 id, with the blocks run as conv
    chains (bf16 inner tensors) and as separate layers -- bit-identical."""
    from model.model_interp import IFNet
    torch.manual_seed(555)
    net = IFNet(51).train().cuda()
    f = torch.rand(2, 2, 64, 64, device="cuda")
    x = torch.cat((f[:, :1].expand(2, 3, 64, 64), f[:, 1:].expand(2, 3, 64, 64)), 1).contiguous()
    target = torch.rand(2, 1, 64, 64, device="cuda")
    HF.set_algorithm(HF.ALGO_MFMA_BF16)
    res = []
    for io in (False, True):
        monkeypatch.setattr(HF, "_BF16_IO", io)
        net.zero_grad(set_to_none=True)
        loss = torch.nn.functional.l1_loss(net(x), target)
        loss.backward()
        res.append([loss.detach().clone()] + [None if p.grad is None else p.grad.clone() for p in net.parameters()])
    assert all(t is None or torch.isfinite(t).all() for t in res[1])
    assert sum(t is not None for t in res[1]) > 90
    bad = [i for i, (a, b) in enumerate(zip(res[0], res[1])) if (a is None) != (b is None) or (a is not None and not torch.equal(a, b))]
    assert not bad, "tensors that differ: %s of %d" % (bad[:8], len(res[0]))


# ---- the ReLU mask inside the bf16 launches (sstem_conv3x3_forward_bf16io_masked / sstem_conv3x3_backward_weight_bf16_masked) -------
@pytest.mark.parametrize("shape", [(2, 40, 24, 64, 70), (2, 128, 32, 32, 256), (3, 64, 12, 36, 32)])      # incl. a launch split over K
def test_bf16_masked_launches_equal_the_separate_passes_bit_for_bit(shape):
    import sstem_native
    lib = sstem_native.load_library()
    N, Cin, H, W, Cout = shape
    BF = HF.ALGO_MFMA_BF16
    torch.manual_seed(31)
    x = torch.randn(N, Cin, H, W, device="cuda"); w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.1
    b = torch.randn(Cout, device="cuda"); g = torch.randn(N, Cout, H, W, device="cuda")

    def fwd(inp, weight, bias, cout, flags, act, out_bf16=False, in_mask=None, want_mask=False, masked_entry=False):
        n, cin = inp.shape[0], inp.shape[1]
        ws_n = lib.sstem_conv3x3_forward_workspace_floats_algo(n, cin, H, W, cout, BF)
        if out_bf16:
            ws_n = lib.sstem_conv3x3_packed_floats(cin, cout, BF)           # a bf16 output: unsplit launch
        ws = torch.empty(ws_n, device="cuda")
        out = torch.empty(n, cout, H, W, device="cuda", dtype=torch.bfloat16 if out_bf16 else torch.float32)
        om = torch.zeros(n, cout, H, W, dtype=torch.bool, device="cuda") if want_mask else None
        p = lambda t: t.data_ptr() if t is not None else None      # noqa: E731
        if masked_entry:
            rc = lib.sstem_conv3x3_forward_bf16io_masked(inp.data_ptr(), 1 if inp.dtype == torch.bfloat16 else 0, p(in_mask), weight.data_ptr(),
                                                         p(bias), None, None, out.data_ptr(), 1 if out_bf16 else 0, p(om), ws.data_ptr(), ws_n,
                                                         n, cin, H, W, cout, flags, act, 0.0, None)
        else:
            rc = lib.sstem_conv3x3_forward_bf16io(inp.data_ptr(), 1 if inp.dtype == torch.bfloat16 else 0, weight.data_ptr(), p(bias), None, None,
                                                  out.data_ptr(), 1 if out_bf16 else 0, ws.data_ptr(), ws_n, n, cin, H, W, cout, flags, act,
                                                  0.0, None)
        assert rc == 0
        torch.cuda.synchronize()
        return out, om
    for out_bf16 in (False, True):
        if out_bf16 and not lib.sstem_conv3x3_bf16io_supported(N, Cin, H, W, Cout, 1):
            continue
        plain, _ = fwd(x, w, b, Cout, 0, HF.ACT_RELU, out_bf16)
        out, mask = fwd(x, w, b, Cout, 0, HF.ACT_RELU, out_bf16, want_mask=True, masked_entry=True)
        assert torch.equal(out, plain) and torch.equal(mask, plain > 0)
        xb = x.bfloat16()                                                    # a bf16 input tensor (inside a chain): mask out only
        plain_b, _ = fwd(xb, w, b, Cout, 0, HF.ACT_RELU, out_bf16)
        out_b, mask_b = fwd(xb, w, b, Cout, 0, HF.ACT_RELU, out_bf16, want_mask=True, masked_entry=True)
        assert torch.equal(out_b, plain_b) and torch.equal(mask_b, plain_b > 0)
    plain, _ = fwd(x, w, b, Cout, 0, HF.ACT_RELU)
    mask = plain > 0
    gm = torch.where(mask, g, torch.zeros((), device="cuda"))
    ref_gx, _ = fwd(gm, w, None, Cin, 1, HF.ACT_NONE)
    gx, _ = fwd(g, w, None, Cin, 1, HF.ACT_NONE, in_mask=mask, masked_entry=True)
    assert torch.equal(gx, ref_gx)
    ws_n = lib.sstem_conv3x3_wgrad_workspace_floats_algo(N, Cin, H, W, Cout, BF); ws = torch.empty(ws_n, device="cuda")
    for xin in (x, x.bfloat16()):
        gw0 = torch.empty(Cout, Cin, 3, 3, device="cuda"); gb0 = torch.empty(Cout, device="cuda")
        gw1 = torch.empty_like(gw0); gb1 = torch.empty_like(gb0)
        ib = 1 if xin.dtype == torch.bfloat16 else 0
        assert lib.sstem_conv3x3_backward_weight_bf16_masked(xin.data_ptr(), ib, gm.data_ptr(), None, gw0.data_ptr(), gb0.data_ptr(), ws.data_ptr(),
                                                             ws_n, N, Cin, H, W, Cout, 0, None) == 0
        assert lib.sstem_conv3x3_backward_weight_bf16_masked(xin.data_ptr(), ib, g.data_ptr(), mask.data_ptr(), gw1.data_ptr(), gb1.data_ptr(),
                                                             ws.data_ptr(), ws_n, N, Cin, H, W, Cout, 0, None) == 0
        torch.cuda.synchronize()
        assert torch.equal(gw1, gw0) and torch.equal(gb1, gb0)


def test_bf16_mask_fusion_in_conv_chains_changes_nothing(monkeypatch):
    """A Conv-ReLU-Conv-ReLU-Conv-ReLU block (one _ConvChain under the bf16 id) and a single Conv+ReLU layer: loss and every gradient
    bit-identical with the masks inside the launches and as separate compare / select passes."""
    import torch.nn as nn
    from hipnn import FusedSequential
    HF.set_algorithm(HF.ALGO_MFMA_BF16)
    res = []
    for fusion in (True, False):
        monkeypatch.setattr(HF, "_MASK_FUSION", fusion)
        torch.manual_seed(32)
        chain = FusedSequential(nn.Conv2d(16, 48, 3, padding=1), nn.ReLU(), nn.Conv2d(48, 40, 3, padding=1), nn.ReLU(),
                                nn.Conv2d(40, 24, 3, padding=1), nn.ReLU()).cuda()
        single = FusedSequential(nn.Conv2d(24, 20, 3, padding=1), nn.ReLU()).cuda()
        x = torch.randn(2, 16, 20, 32).cuda().requires_grad_(True)
        out = single(chain(x))
        out.square().mean().backward()
        res.append([out.detach(), x.grad] + [p.grad for m in (chain, single) for p in m.parameters()])
    for u, v in zip(res[0], res[1]):
        assert torch.equal(u, v)
