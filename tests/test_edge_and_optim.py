"""uint8 edge (f3) and flat Adam / LR schedule / checkpoint layout (f4)."""
import math
import os

import numpy as np
import pytest
import torch

import train_utils


def test_poly_lr_matches_reference_formula():
    # calculate_lr (main_ms.py:127-135) with the shipped config: base 1e-3, end 1e-4, warmup 0, decay 100000, power 1.5
    cfg = dict(base_lr=0.001, end_lr=0.0001, warmup_iters=0, decay_iters=100000, power=1.5)
    assert train_utils.poly_lr(0, **cfg) == pytest.approx(0.001)
    assert train_utils.poly_lr(50000, **cfg) == pytest.approx(0.0009 * (0.5 ** 1.5) + 0.0001)
    assert train_utils.poly_lr(100000, **cfg) == 0.0001 and train_utils.poly_lr(10 ** 6, **cfg) == 0.0001
    w = dict(base_lr=0.01, end_lr=0.001, warmup_iters=10, decay_iters=100, power=2.0)
    assert train_utils.poly_lr(5, **w) == pytest.approx(0.009 * 0.25 + 0.001)
    assert train_utils.poly_lr(55, **w) == pytest.approx(0.009 * (1 - 45 / 100) ** 2 + 0.001)


def test_checkpoint_layout(tmp_path):
    net = torch.nn.Conv2d(2, 3, 3)
    path = train_utils.save_checkpoint(net, 1000, str(tmp_path), data_parallel_prefix=True)
    assert os.path.basename(path) == "model-001000.ckpt"
    ck = torch.load(path)
    assert set(ck) == {"current_iter", "valid_result", "model_weights"} and ck["valid_result"] is None
    assert all(k.startswith("module.") for k in ck["model_weights"])
    other = torch.nn.Conv2d(2, 3, 3)
    assert train_utils.load_checkpoint(other, path) == 1000
    assert torch.equal(other.weight, net.weight)


def test_numpy_u8_division_forms_agree():
    """Gray2Tensor divides in float64 then casts, the CLI divides in float32: identical for all 256 values,
    so one kernel (float32 IEEE division) serves both."""
    k = np.arange(256, dtype=np.uint8)
    assert np.array_equal((k / 255.).astype("float32"), k.astype(np.float32) / 255.0)


@pytest.mark.gpu
def test_u8_edge_kernels_bit_exact():
    from utils.gray2tensor import Gray2Tensor, TrainTensor2mask, gray_to_tensor, tensor_to_gray
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    img[0, :256 if img.shape[1] >= 256 else img.shape[1]] = np.arange(min(256, img.shape[1]), dtype=np.uint8)
    t = gray_to_tensor(img, replicas=3)
    want = np.repeat((img.astype(np.float32) / 255.0)[None], 3, 0)[None]
    assert t.shape == (1, 3, 37, 53) and np.array_equal(t.cpu().numpy(), want)
    assert np.array_equal(Gray2Tensor(img).cpu().numpy(), (img / 255.).astype("float32")[None, None])
    allk = gray_to_tensor(np.arange(256, dtype=np.uint8).reshape(16, 16)).cpu().numpy().ravel()
    assert np.array_equal(allk, np.arange(256, dtype=np.float32) / np.float32(255))
    # float -> uint8: truncation, wrap-around instead of clamping, NaN / huge -> 0 (numpy on x86-64)
    v = np.array([[0.0, 0.5, 1.0, 256.0 / 255, -1.0 / 255, 300.7 / 255, 0.999999, -0.001, float("nan"), 1e20, -1e20, 0.50196]],
                 np.float32)
    with np.errstate(invalid="ignore"):
        want = (v * 255).astype(np.uint8)
    assert np.array_equal(tensor_to_gray(torch.from_numpy(v).cuda()), want)
    pred = rng.random((64, 64), dtype=np.float32) * 1.2 - 0.1
    with np.errstate(invalid="ignore"):
        assert np.array_equal(tensor_to_gray(torch.from_numpy(pred).cuda()), (pred * 255).astype(np.uint8))
    clamped = np.clip(pred, 0, 1)
    assert np.array_equal(TrainTensor2mask(torch.from_numpy(pred).cuda()[None, None]), (clamped * 255).astype(np.uint8))


@pytest.mark.gpu
def test_flat_adam_matches_torch_adam():
    import dataparallel as dp
    torch.manual_seed(0)
    def make():
        torch.manual_seed(1)
        return torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(8, 2, 3, padding=1)).cuda()
    ref, net = make(), make()
    opt_ref = torch.optim.Adam(ref.parameters(), lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
    flat = train_utils.FlatParams(net.parameters())
    bucket = dp.FlatGradBucket(net.parameters())
    opt = train_utils.FlatAdam(flat.flat, bucket.flat, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
    x = torch.randn(4, 3, 10, 10, device="cuda"); t = torch.randn(4, 2, 10, 10, device="cuda")
    for it in range(5):
        lr = train_utils.poly_lr(it, 1e-3, 1e-4, 0, 10, 1.5)
        for g in opt_ref.param_groups:
            g["lr"] = lr
        opt_ref.zero_grad(); torch.nn.functional.l1_loss(ref(x), t).backward(); opt_ref.step()
        bucket.zero(); torch.nn.functional.l1_loss(net(x), t).backward()
        assert bucket.check_views()
        opt.step(lr=lr)
    for a, b in zip(ref.parameters(), net.parameters()):
        assert (a - b).abs().max().item() <= 1e-6 * max(1.0, a.abs().max().item())


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 1, 256, 256), (16, 1, 256, 256), (1, 1, 7, 5), (3, 2, 33, 31), (1, 1, 2048, 2048)])
def test_native_l1_loss_and_gradient_in_one_launch(shape):
    """train_utils.L1MeanLoss (include/sstem_io.h, sstem_l1_mean_forward_grad_f32) against nn.L1Loss + autograd (the reference's
    criterion, sff_scripts_fusion/main_fusion.py:252): the loss within the order of an fp32 sum (float64 as the judge), the gradient bit
    for bit (sign(pred - target) / n, zero on ties), repeated calls bit-identical (fixed-order sums, the workspace left clean), and
    the backward pass started at the network's output gives the parameter gradients autograd gives from the loss."""
    g = torch.Generator().manual_seed(21)
    pred = torch.rand(shape, generator=g).cuda().requires_grad_()
    target = torch.rand(shape, generator=g).cuda()
    with torch.no_grad():
        target.view(-1)[::7] = pred.detach().view(-1)[::7]                      # ties: gradient exactly zero there
    crit = train_utils.L1MeanLoss(pred.device)
    loss, grad = crit(pred, target)
    ref = torch.nn.functional.l1_loss(pred, target)
    ref.backward()
    exact = (pred.detach().double() - target.double()).abs().mean().item()
    assert abs(loss.item() - exact) <= 4e-7 * exact and abs(ref.item() - exact) <= 4e-7 * exact
    assert torch.equal(grad, pred.grad)
    for _ in range(3):
        l2, g2 = crit(pred, target)
        assert torch.equal(l2, loss) and torch.equal(g2, grad)
    # unaligned views take the scalar path
    p1, t1 = pred.detach().view(-1)[1:], target.view(-1)[1:]
    l3, g3 = crit(p1, t1)
    assert abs(l3.item() - (p1.double() - t1.double()).abs().mean().item()) <= 1e-6
    assert torch.equal(g3, torch.sign(p1 - t1) / p1.numel())
    # a network in front: pred.backward(grad) == loss.backward()
    torch.manual_seed(3)
    net = torch.nn.Conv2d(shape[1], shape[1], 3, padding=1).cuda()
    x = torch.rand(shape, generator=g).cuda()
    torch.nn.functional.l1_loss(net(x), target).backward()
    want = [p.grad.clone() for p in net.parameters()]
    net.zero_grad()
    out = net(x)
    _, go = crit(out, target)
    out.backward(go)
    for p, w in zip(net.parameters(), want):          # (torch's own convolution gradient is not bit-reproducible from call to call)
        assert torch.allclose(p.grad, w, rtol=1e-5, atol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 3, 130, 200), (1, 2, 64, 128), (2, 5, 256, 256), (1, 3, 250, 512), (1, 2, 37, 1030), (3, 51, 128, 128)])
def test_wide_bilinear_upsample_kernel_equals_the_one_column_kernel_bit_for_bit(shape, monkeypatch):
    """Outputs at least 256 columns wide take the kernel that stores four adjacent columns per thread as 16-byte vectors (round 4): the same
    expression per output, its rounding spelled out (lerp2 in misc_kernels.hip), as the one-column kernel (SSTEM_UPSAMPLE_WIDE=0) => the same
    bits, ragged tiles included."""
    import hipnn.functional as HF
    g = torch.Generator().manual_seed(10)
    x = torch.randn(*shape, generator=g).cuda()
    wide = HF.upsample_bilinear2x(x)
    monkeypatch.setenv("SSTEM_UPSAMPLE_WIDE", "0")
    narrow = HF.upsample_bilinear2x(x)
    assert torch.equal(wide, narrow)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 3, 5, 8), (1, 51, 16, 32), (1, 1, 1, 2), (3, 2, 7, 6), (1, 4, 33, 70),
                                   (2, 3, 130, 200), (1, 2, 64, 128), (3, 1, 3, 64)])      # the last three: the tiled kernel (several / ragged tiles)
def test_native_bilinear_upsample_matches_torch(shape):
    """nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True) (model_interp.py:17, networks.py:27) as one native
    launch on the inference path.  The arithmetic reference is torch's own fp32 kernel (what the reference delegates to and
    the goldens were made with); the yardstick is float64.  Measured on MI355X at (1,51,16,32): torch CPU and GPU fp32 are
    4.7e-6 from float64 (one ulp of the source coordinate dst*(in-1)/(out-1) at a small output between large neighbours), the
    native kernel 2.0e-6.  So the criterion is: never further from float64 than torch's fp32 kernel (+ rounding slack), and
    within the sum of the two distances from torch's result -- not a fixed bound on |native - torch|, which would measure
    torch's rounding."""
    import torch.nn.functional as F
    import hipnn.functional as HF
    g = torch.Generator().manual_seed(9)
    x = torch.randn(*shape, generator=g)
    got = HF.upsample_bilinear2x(x.cuda()).cpu()
    ref32 = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
    ref64 = F.interpolate(x.double(), scale_factor=2, mode="bilinear", align_corners=True)
    assert got.shape == ref32.shape
    scale = float(ref64.abs().max())
    err_native = float((got.double() - ref64).abs().max())
    err_torch = float((ref32.double() - ref64).abs().max())
    assert err_native <= 1.25 * err_torch + 2e-7 * scale, "native %.2e vs torch fp32 %.2e from float64" % (err_native, err_torch)
    assert float((got - ref32).abs().max()) <= err_native + err_torch + 1e-7 * scale
    # the module dispatch: a FusedSequential child takes the native launch under no_grad and torch's op when recording
    from hipnn import FusedSequential
    seq = FusedSequential(torch.nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True)).cuda()
    with torch.no_grad():
        assert torch.equal(seq(x.cuda()).cpu(), got)          # planes of these sizes take the native launch
    # recording: native forward + native backward (a deterministic gather; torch scatters with atomic adds, so its own result is
    # not bit-reproducible either).  Yardstick: float64 autograd; criterion as above -- not further from it than torch's fp32
    # gradient (+ slack), and the same bits on a second run.
    go = torch.randn(ref32.shape, generator=g)
    xg = x.cuda().requires_grad_()
    yg = seq(xg)
    assert torch.equal(yg.detach().cpu(), got)
    yg.backward(go.cuda())
    xt = x.clone().requires_grad_()
    F.interpolate(xt, scale_factor=2, mode="bilinear", align_corners=True).backward(go)
    xd = x.double().requires_grad_()
    F.interpolate(xd, scale_factor=2, mode="bilinear", align_corners=True).backward(go.double())
    gscale = float(xd.grad.abs().max())
    gerr_native = float((xg.grad.cpu().double() - xd.grad).abs().max())
    gerr_torch = float((xt.grad.double() - xd.grad).abs().max())
    assert gerr_native <= 1.25 * gerr_torch + 5e-7 * gscale, "native grad %.2e vs torch fp32 %.2e from float64" % (gerr_native, gerr_torch)
    xg2 = x.cuda().requires_grad_()
    seq(xg2).backward(go.cuda())
    assert torch.equal(xg.grad, xg2.grad)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 5, 16, 32), (1, 3, 9, 13), (3, 2, 7, 8), (1, 1, 2, 2), (2, 4, 33, 66)])
@pytest.mark.parametrize("kind", ["max", "avg"])
def test_native_pool2x2_matches_torch_bit_for_bit(shape, kind):
    """nn.MaxPool2d(2) / nn.AvgPool2d((2,2),(2,2)) through hipnn.functional.pool_module: output and input gradient equal torch's on
    the same GPU tensors bit for bit (ties -- all-zero windows after a ReLU -- go to the first element, odd sizes floor, NaN propagates)."""
    import torch.nn as nn
    import hipnn.functional as HF
    torch.manual_seed(81)
    m = nn.MaxPool2d(2) if kind == "max" else nn.AvgPool2d((2, 2), (2, 2))
    x = torch.relu(torch.randn(*shape, device="cuda"))                 # many exact ties at zero
    x[0, 0, 0, 1] = float("nan")
    xa = x.clone().requires_grad_(True); xb = x.clone().requires_grad_(True)
    ya = HF.pool_module(m, xa); yb = m(xb)
    assert ya.shape == yb.shape
    assert torch.equal(torch.nan_to_num(ya, nan=-7.0), torch.nan_to_num(yb, nan=-7.0))
    g = torch.randn_like(yb)
    ya.backward(g); yb.backward(g)
    assert torch.equal(xa.grad, xb.grad)
    with torch.no_grad():
        assert torch.equal(torch.nan_to_num(HF.pool_module(m, x), nan=-7.0), torch.nan_to_num(m(x), nan=-7.0))
    # another geometry goes to the module itself
    if min(shape[2:]) < 3:
        return
    m3 = nn.MaxPool2d(3)
    assert torch.equal(torch.nan_to_num(HF.pool_module(m3, x), nan=-7.0), torch.nan_to_num(m3(x), nan=-7.0))


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1, 8, 16, 32, 5), (3, 4, 9, 66, 7), (2, 16, 64, 128, 16)])
def test_skip_cat_upsample_writes_into_the_concatenated_tensor(shape):
    """networks.Up: cat([skip, up(x)], 1).  Under no_grad the up-sampled planes are written straight into their half of the result
    (hipnn.functional.skip_cat_upsample2x): the same bits as the up-sampling launch + torch.cat; when recording, and for a skip of
    another size (the reference's F.pad), the generic path with the same values as the module's own."""
    import torch.nn as nn
    import torch.nn.functional as F
    import hipnn.functional as HF
    N, C, H, W, Cs = shape
    torch.manual_seed(5)
    up = nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True)
    x = torch.randn(N, C, H, W, device="cuda"); skip = torch.randn(N, Cs, 2 * H, 2 * W, device="cuda")
    with torch.no_grad():
        got = HF.skip_cat_upsample2x(up, skip, x)
        want = torch.cat([skip, HF.upsample_bilinear2x(x)], 1)
    assert got.shape == want.shape and torch.equal(got, want)
    xg = x.clone().requires_grad_()
    rec = HF.skip_cat_upsample2x(up, skip, xg)
    assert rec.requires_grad and torch.equal(rec.detach(), want)
    rec.sum().backward()
    assert xg.grad is not None and torch.isfinite(xg.grad).all()
    big = torch.randn(N, Cs, 2 * H + 3, 2 * W + 2, device="cuda")                  # odd skip: padded like the reference (networks.py:226-229)
    with torch.no_grad():
        got = HF.skip_cat_upsample2x(up, big, x)
        ref = torch.cat([big, F.pad(HF.upsample_bilinear2x(x), [1, 1, 1, 2])], 1)
    assert torch.equal(got, ref)


@pytest.mark.gpu
@pytest.mark.parametrize("bf16", [False, True])
@pytest.mark.parametrize("size", [(32, 48), (128, 64)])
def test_sp_unet_with_skips_stored_in_place_equals_the_concatenating_forward(size, bf16, monkeypatch):
    """networks.UNet on one image under no_grad: every encoder output is stored by its last launch inside the tensor the decoder
    concatenates (FusedSequential(out=...)), the up-sampled half by the up-sampling launch -- the same launches on the same values as
    the forward with four torch.cat: bit-identical."""
    import networks
    torch.manual_seed(77)
    net = networks.FusionNet(1, 1).cuda().eval()
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5)
    a = torch.rand(1, 1, *size, device="cuda"); b = torch.rand(1, 1, *size, device="cuda")
    import contextlib
    import hipnn.functional as HF
    with torch.no_grad(), (HF.algorithm(HF.ALGO_MFMA_BF16) if bf16 else contextlib.nullcontext()):      # the bf16-operand id stores through conv3x3_bf16io
        assert net._skips_in_place(a)
        fast = net(a, b)
        monkeypatch.setattr(networks.UNet, "_skips_in_place", lambda self, x: False)
        ref = net(a, b)
    assert torch.equal(fast, ref)
    two = torch.rand(2, 1, *size, device="cuda")
    monkeypatch.undo()
    with torch.no_grad():
        assert not net._skips_in_place(two)
    assert not net._skips_in_place(a)            # grad mode on: the recording path
