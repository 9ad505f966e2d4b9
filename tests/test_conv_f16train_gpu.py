"""GPU parity of the RECORDED (training) launches on the fp16 two-piece id (include/sstem_conv.h: sstem_conv3x3_forward_scaled_masked_f32,
sstem_conv3x3_backward_weight_scaled_masked_f32; csrc/conv_split_kernels.hip): F16X3's arithmetic (tests/test_conv_f16x3_gpu.py) with the
ReLU bookkeeping of the X6 launches (tests/test_conv_split_gpu.py).  Reference: float64 PyTorch on the CPU, identical inputs; tolerance:
the fp32 kernels' own bound, max|a - ref| <= 2e-5 max|ref| + 1e-6.  The masks change no bit: a launch that applies / writes a mask
equals the launch on pre-masked data."""
import pytest
import torch
import torch.nn.functional as F

import hipnn.functional as HF
import sstem_native
from test_conv_split_gpu import _close

pytestmark = pytest.mark.gpu

# (N, Cin, H, W, Cout), W % 4 == 0: plain, ragged channel counts, a launch split over K, 16-wide maps, tap-row last chunks (51, 35)
SHAPES = [(2, 40, 24, 64, 70), (2, 128, 32, 32, 256), (3, 64, 13, 36, 64), (2, 64, 16, 16, 96), (1, 32, 24, 12, 32), (2, 51, 24, 64, 51),
          (1, 35, 16, 16, 20), (2, 256, 8, 8, 512), (1, 16, 40, 96, 32)]


def _word(lib, t):
    w = torch.zeros(1024, device="cuda")
    assert lib.sstem_amax_f32(t.data_ptr(), t.numel(), w.data_ptr(), None) == 0
    return w


def _fwd(lib, x, w, b, cout, flags, act, in_mask=None, want_mask=False, masked_entry=True, x_word=None):
    n, cin, H, W = x.shape
    ws_n = lib.sstem_conv3x3_forward_workspace_floats_algo(n, cin, H, W, cout, HF.ALGO_MFMA_F16X3)
    ws = torch.empty(ws_n, device="cuda"); out = torch.empty(n, cout, H, W, device="cuda")
    xw = x_word if x_word is not None else _word(lib, x)
    ow = torch.zeros(1024, device="cuda")
    om = torch.zeros(n, cout, H, W, dtype=torch.bool, device="cuda") if want_mask else None
    bp = b.data_ptr() if b is not None else None
    if masked_entry:
        rc = lib.sstem_conv3x3_forward_scaled_masked_f32(x.data_ptr(), xw.data_ptr(), in_mask.data_ptr() if in_mask is not None else None,
                                                         w.data_ptr(), bp, None, None, out.data_ptr(), ow.data_ptr(),
                                                         om.data_ptr() if om is not None else None, ws.data_ptr(), ws_n, n, cin, H, W, cout,
                                                         flags, act, 0.0, None)
    else:
        rc = lib.sstem_conv3x3_forward_scaled_strided_f32(x.data_ptr(), xw.data_ptr(), w.data_ptr(), bp, None, None, None, 1.0, out.data_ptr(),
                                                          ow.data_ptr(), ws.data_ptr(), ws_n, n, cin, H, W, cout, flags, act, 0.0, None,
                                                          HF.ALGO_MFMA_F16X3, 0, 0, None, 0)
    assert rc == 0, sstem_native.last_error() if hasattr(sstem_native, "last_error") else rc
    torch.cuda.synchronize()
    return out, om, ow


@pytest.mark.parametrize("shape", SHAPES)
def test_recorded_forward_and_data_gradient_on_fp16_pieces(shape):
    lib = sstem_native.load_library()
    N, Cin, H, W, Cout = shape
    torch.manual_seed(51)
    x = torch.randn(N, Cin, H, W, device="cuda"); w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.1
    b = torch.randn(Cout, device="cuda"); g = torch.randn(N, Cout, H, W, device="cuda") * 1e-3          # gradient-sized
    # forward: the inference entry's bits, mask = (output > 0), bound = the largest stored magnitude
    plain, _, _ = _fwd(lib, x, w, b, Cout, 0, HF.ACT_RELU, masked_entry=False)
    out, mask, ow = _fwd(lib, x, w, b, Cout, 0, HF.ACT_RELU, want_mask=True)
    assert torch.equal(out, plain) and torch.equal(mask, plain > 0)
    assert ow.max().item() == out.abs().max().item()
    _close(out, F.relu(F.conv2d(x.double().cpu(), w.double().cpu(), b.double().cpu(), padding=1)))
    # data gradient (transposed + flipped weights): the mask applied while staging == the select pass first, under the same bound
    gw_ = _word(lib, g)
    gm = torch.where(mask, g, torch.zeros((), device="cuda"))
    ref_gx, _, _ = _fwd(lib, gm, w, None, Cin, 1, HF.ACT_NONE, masked_entry=False, x_word=gw_)
    gx, _, gxw = _fwd(lib, g, w, None, Cin, 1, HF.ACT_NONE, in_mask=mask, x_word=gw_)
    assert torch.equal(gx, ref_gx)
    assert gxw.max().item() == gx.abs().max().item()
    _close(gx, F.conv_transpose2d(gm.double().cpu(), w.double().cpu(), padding=1))
    # a mask on both sides at once (a ReLU layer between two ReLU layers)
    m2 = torch.rand(N, Cin, H, W, device="cuda") > 0.4
    both, om, _ = _fwd(lib, x, w, b, Cout, 0, HF.ACT_RELU, in_mask=m2, want_mask=True)
    ref, _, _ = _fwd(lib, torch.where(m2, x, torch.zeros((), device="cuda")), w, b, Cout, 0, HF.ACT_RELU, masked_entry=False, x_word=_word(lib, x))
    assert torch.equal(both, ref) and torch.equal(om, ref > 0)


WSHAPES = SHAPES + [(1, 24, 9, 37, 33), (2, 20, 17, 50, 48)]          # + dword staging (W % 4 != 0)


@pytest.mark.parametrize("shape", WSHAPES)
def test_weight_gradient_on_fp16_pieces(shape):
    lib = sstem_native.load_library()
    N, Cin, H, W, Cout = shape
    torch.manual_seed(52)
    x = torch.randn(N, Cin, H, W, device="cuda") * 3.0; g = torch.randn(N, Cout, H, W, device="cuda") * 1e-4
    mask = torch.rand(N, Cout, H, W, device="cuda") > 0.5
    gm = torch.where(mask, g, torch.zeros((), device="cuda"))
    xw, gw_ = _word(lib, x), _word(lib, g)
    ws_n = lib.sstem_conv3x3_wgrad_workspace_floats_algo(N, Cin, H, W, Cout, HF.ALGO_MFMA_F16X3); ws = torch.empty(ws_n, device="cuda")

    def run(gt, m, acc=0, gw=None, gb=None):
        gw = torch.empty(Cout, Cin, 3, 3, device="cuda") if gw is None else gw
        gb = torch.empty(Cout, device="cuda") if gb is None else gb
        rc = lib.sstem_conv3x3_backward_weight_scaled_masked_f32(x.data_ptr(), xw.data_ptr(), gt.data_ptr(), gw_.data_ptr(),
                                                                 m.data_ptr() if m is not None else None, gw.data_ptr(), gb.data_ptr(),
                                                                 ws.data_ptr(), ws_n, N, Cin, H, W, Cout, acc, None)
        assert rc == 0
        torch.cuda.synchronize()
        return gw, gb
    gw0, gb0 = run(gm, None)
    gw1, gb1 = run(g, mask)
    assert torch.equal(gw0, gw1) and torch.equal(gb0, gb1)
    xd = x.double().cpu().requires_grad_(False); gd = gm.double().cpu()
    ref = torch.nn.grad.conv2d_weight(xd, (Cout, Cin, 3, 3), gd, padding=1)
    _close(gw0, ref); _close(gb0, gd.sum((0, 2, 3)))
    gw2, gb2 = run(g, mask, 1, gw0.clone(), gb0.clone())
    _close(gw2, 2 * ref); _close(gb2, 2 * gd.sum((0, 2, 3)))
    again, _ = run(g, mask)
    assert torch.equal(again, gw1)                       # fixed-order sums: bit-reproducible


@pytest.mark.parametrize("xs,gs", [(1e-6, 1.0), (3e5, 1e-9), (1.0, 4e4), (1e12, 1e-20)])
def test_weight_gradient_on_fp16_pieces_is_scale_invariant(xs, gs):
    lib = sstem_native.load_library()
    N, Cin, H, W, Cout = 2, 48, 20, 40, 70
    torch.manual_seed(53)
    x = torch.randn(N, Cin, H, W, device="cuda") * xs; g = torch.randn(N, Cout, H, W, device="cuda") * gs
    ws_n = lib.sstem_conv3x3_wgrad_workspace_floats_algo(N, Cin, H, W, Cout, HF.ALGO_MFMA_F16X3); ws = torch.empty(ws_n, device="cuda")
    gw = torch.empty(Cout, Cin, 3, 3, device="cuda")
    assert lib.sstem_conv3x3_backward_weight_scaled_masked_f32(x.data_ptr(), _word(lib, x).data_ptr(), g.data_ptr(), _word(lib, g).data_ptr(),
                                                               None, gw.data_ptr(), None, ws.data_ptr(), ws_n, N, Cin, H, W, Cout, 0, None) == 0
    torch.cuda.synchronize()
    ref = torch.nn.grad.conv2d_weight(x.double().cpu(), (Cout, Cin, 3, 3), g.double().cpu(), padding=1)
    err = (gw.double().cpu() - ref).abs().max().item()
    assert torch.isfinite(gw).all() and err <= 2e-6 * ref.abs().max().item()


# ---- through hipnn: ALGO_AUTO sends the recorded launches of X6-sized layers to these entry points ------------------------------------
import torch.nn as nn                                    # noqa: E402
from hipnn import FusedSequential                        # noqa: E402


@pytest.fixture(autouse=True)
def _reset_algo():
    yield
    HF.set_algorithm(HF.ALGO_AUTO)


def _net(seed, chans):
    torch.manual_seed(seed)
    return [nn.Conv2d(ci, co, 3, padding=1) for ci, co in chans]


def _rel_close(a, ref, rel=2e-5):
    """max|a - ref| <= rel * max|ref| with NO absolute term (gradients of a mean are 1e-6-sized; test_conv_split_gpu._close adds 1e-6)."""
    ref = ref.detach().double().cpu()
    err = (a.detach().double().cpu() - ref).abs().max().item()
    assert err <= rel * ref.abs().max().item(), "max err %.3e vs scale %.3e" % (err, ref.abs().max().item())


@pytest.mark.parametrize("W", [32, 30])                 # masks inside the launches (W % 4 == 0) and as passes of their own
@pytest.mark.parametrize("on", [True, False])
def test_training_step_on_fp16_pieces_against_float64(W, on, monkeypatch):
    """Conv+ReLU x3 with a skip add (two consumers of one activation) + a Conv without activation, sized so that ALGO_AUTO picks the split
    kernels (>= 128 input channels): output, input gradient and every parameter gradient within 2e-5 of the largest value of float64
    torch -- no absolute term, the gradients of a mean are 1e-6-sized.  The launches really are the fp16 ones (their outputs carry a
    bound; X6's do not); knob off = X6, held to the same bound.
    The float64 chain takes its ReLU decisions from the GPU activations (a pre-activation within rounding of zero may land on either
    side, and ONE flipped element moves the gradients below it by a percent of their largest value: measured, X6 and fp32 alike --
    the loss is not differentiable there, so such a difference says nothing about the arithmetic)."""
    chans = [(128, 128), (128, 128), (128, 160), (160, 128)]
    x0 = torch.randn(2, 128, 24, W, generator=torch.Generator().manual_seed(3))
    monkeypatch.setattr(HF, "_AUTO_F16_TRAIN", on)
    convs = _net(71, chans)
    net = [FusedSequential(c, nn.ReLU()).cuda() for c in convs[:3]] + [FusedSequential(convs[3]).cuda()]
    x = x0.cuda().requires_grad_(True)
    a = net[0](x); bb = net[1](a); c = net[2](a + bb); out = net[3](c)
    assert (HF.amax_word_of(a) is not None) == on and (HF.amax_word_of(out) is not None) == on
    if on:
        assert float(HF.amax_word_of(a).max()) == float(a.detach().abs().max())
    out.square().mean().backward()
    got = [out.detach(), x.grad] + [p.grad for m in net for p in m.parameters()]
    masks = [(t.detach() > 0).cpu() for t in (a, bb, c)]
    convs = _net(71, chans)
    for cv in convs:
        cv.double()
    xd = x0.double().requires_grad_(True)
    zero = torch.zeros((), dtype=torch.float64)
    ra = torch.where(masks[0], convs[0](xd), zero); rb = torch.where(masks[1], convs[1](ra), zero)
    rc = torch.where(masks[2], convs[2](ra + rb), zero); rout = convs[3](rc)
    rout.square().mean().backward()
    ref = [rout, xd.grad] + [p.grad for cv in convs for p in cv.parameters()]
    for g, r in zip(got, ref):
        _rel_close(g, r)


def test_group_packing_of_fp16_pieces_after_the_optimiser_step(monkeypatch):
    """FlatAdam.step re-packs every fp16 pair workspace with one clear + one bound launch + one pack launch
    (sstem_conv3x3_pack_weights_group_f16): the same bits as the per-layer launches (header = the layer's bound, then the image), three
    training steps give the same parameters bit for bit with the group on and off, and a writer other than FlatAdam sends the layer
    back to its own pack launches."""
    import train_utils
    from dataparallel import FlatGradBucket
    lib = sstem_native.load_library()
    shapes = [(128, 136), (136, 130), (130, 160), (160, 129)]          # (>= 128 channels on both sides: forward and data gradient on one id)

    def build():
        torch.manual_seed(61)
        layers = []
        for ci, co in shapes:
            layers += [nn.Conv2d(ci, co, 3, padding=1), nn.ReLU()]
        return FusedSequential(*layers).cuda()

    x = torch.randn(2, 128, 12, 32, device="cuda").requires_grad_(True)       # every layer has a data gradient: four pairs
    finals = []
    for group in (True, False):
        monkeypatch.setattr(HF, "_PACK_GROUP", group)
        net = build()
        flat = train_utils.FlatParams(net.parameters())
        bucket = FlatGradBucket(net.parameters())
        opt = train_utils.FlatAdam(flat.flat, bucket.flat, lr=1e-3)
        convs = [m for m in net if isinstance(m, nn.Conv2d)]
        for step in range(3):
            bucket.zero()
            net(x).square().mean().backward()
            opt.step()
            if group and step == 0:
                n_f16 = 0
                for c in convs:
                    for key, s in c.weight.__dict__.get("_sstem_pack_slots", {}).items():
                        assert s.sig == (c.weight._version, c.weight.data_ptr())
                        if key[0] != HF.ALGO_MFMA_F16X3:
                            continue
                        n_f16 += 1
                        ref_f, ref_t = torch.empty_like(s.ws_f), torch.empty_like(s.ws_t)
                        assert lib.sstem_conv3x3_pack_weights_f32(c.weight.data_ptr(), c.in_channels, c.out_channels, HF.ALGO_MFMA_F16X3,
                                                                  ref_f.data_ptr(), ref_t.data_ptr(), None) == 0
                        torch.cuda.synchronize()
                        # bound + image (the amax word behind the image is scratch of the per-layer form)
                        n_f = lib.sstem_conv3x3_packed_floats(c.in_channels, c.out_channels, HF.ALGO_MFMA_F16X3) - 1024
                        n_t = lib.sstem_conv3x3_packed_floats(c.out_channels, c.in_channels, HF.ALGO_MFMA_F16X3) - 1024
                        assert float(s.ws_f[0]) == float(c.weight.detach().abs().max()) == float(s.ws_t[0]) == float(ref_f[0])
                        # (the header is 16 bytes: the bound and three floats nobody reads)
                        assert torch.equal(s.ws_f[4:n_f].view(torch.int32), ref_f[4:n_f].view(torch.int32))
                        assert torch.equal(s.ws_t[4:n_t].view(torch.int32), ref_t[4:n_t].view(torch.int32))
                assert n_f16 == 4
        finals.append(flat.flat.clone())
        if group:
            c = convs[1]
            s = next(iter(c.weight._sstem_pack_slots.values()))
            with torch.no_grad():
                c.weight.mul_(0.5)
            assert s.sig != (c.weight._version, c.weight.data_ptr())
            out = net(x)
            ref = x
            for m in net:
                ref = F.conv2d(ref, m.weight, m.bias, padding=1) if isinstance(m, nn.Conv2d) else F.relu(ref)
            _close(out, ref, 1e-4)
            assert s.sig == (c.weight._version, c.weight.data_ptr())
    assert torch.equal(finals[0], finals[1])


@pytest.mark.parametrize("shape", [(2, 6, 40, 64, 32), (2, 32, 40, 64, 2), (1, 24, 33, 50, 20), (2, 128, 8, 8, 128)])
@pytest.mark.parametrize("forced", [False, True])
def test_narrow_layers_and_a_forced_fp16_id_train_against_float64(shape, forced):
    """Layers of a few channels (ALGO_AUTO: forward and data gradient on the fp32 MFMA kernels, the weight gradient on fp16 pieces since
    round 5), odd sizes (dword staging), an 8 x 8 map (too few pixels: the bound-free kernels), and the same under a FORCED fp16 id (its
    recorded launches: fp16 pieces where the kernels take the layer, X6 otherwise): input, weight and bias gradients against float64."""
    N, Cin, H, W, Cout = shape
    if forced:
        HF.set_algorithm(HF.ALGO_MFMA_F16X3)
    torch.manual_seed(81)
    conv = nn.Conv2d(Cin, Cout, 3, padding=1)
    net = FusedSequential(conv, nn.ReLU()).cuda()
    x0 = torch.randn(N, Cin, H, W)
    x = x0.cuda().requires_grad_(True)
    out = net(x)
    out.square().sum().backward()
    ref = nn.Conv2d(Cin, Cout, 3, padding=1).double()
    ref.load_state_dict({k: v.double().cpu() for k, v in conv.state_dict().items()})
    xd = x0.double().requires_grad_(True)
    pre = ref(xd)
    rout = torch.where((out.detach() > 0).cpu(), pre, torch.zeros((), dtype=torch.float64))     # the GPU's ReLU decisions (see above)
    rout.square().sum().backward()
    _rel_close(out, rout); _rel_close(x.grad, xd.grad)
    _rel_close(conv.weight.grad, ref.weight.grad); _rel_close(conv.bias.grad, ref.bias.grad)
