"""GPU parity of ONE training step (forward, loss, backward) of the model API against the REFERENCE classes' own step on
CPU -- tests/golden/make_step_goldens.py ("pinned by loss value + selected gradient norms").  Same
deterministic weights and inputs on both sides; every conv / sepconv / warp forward and backward here runs through the
native kernels.

Tolerances are DERIVED, not picked: the generator also runs every reference step in float64 and stores, per quantity, how far
the reference's own fp32 result is from it (`*_norm_cond`, `*_grad{k}_cond`).  The forward is well-conditioned (losses agree
to 1e-8: the loss must match within 2e-5); the gradients of the BatchNorm-bearing nets are not (reference fp32 vs fp64: up to
5.5e-4 in norm, 2.5e-3 elementwise on the first conv -- ReLU / max-pool decisions and BN cancellation), those of the BN-free
IFNet are (1e-6).  A gradient norm must match within max(2e-5, COND_FACTOR x the step's conditioning) relative, plus a floor of 1e-5 of
the step's largest norm, where the step's conditioning is the LARGEST stored deviation among its live parameters: one
fp32-vs-fp64 difference per parameter is a single noisy sample (measured: deviations of 3-5e-4 here on parameters whose own
sample happened to be 5-8e-5, in nets whose other parameters show 3.5-5.5e-4), so the scale is taken per step -- 2e-5 for
the IFNet step, 1.4e-3 .. 2.2e-3 for the BatchNorm nets; a structural error (concat order, BN handling, a wrong mask)
would be O(1), not 1e-3.  A gradient stored in full must match within max(2e-5, COND_FACTOR x its own conditioning) of its largest
element (those are many-element maxima, already stable).
The floor is for gradients that are zero by construction: a conv bias that feeds a train-mode BatchNorm cannot change the
loss (BN removes the per-channel mean), so both sides hold rounding noise there (~1e-7 beside norms of 0.01-1; 17-18 such
biases per U-Net) -- the floor asserts that they stay at rounding level here too.  Parameters the reference leaves without a
gradient must have none here either."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import networks
from model.model_fusionnet import FusionNet as SffFusionNet
from model.model_interp import IFNet as SffIFNet
from model.model_unet import UNet as SffUNet
from utils.image_warp_torch import SpatialTransformation
from weight_recipe import fill_, input_for

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("conv_algo_matrix")]
SEED = 555
# COND_FACTOR: 4 in round 1.  Round 2 measured the scatter of THIS library against itself: the SP UNet step computed by six
# arithmetic-equivalent configurations (BatchNorm statistics from the conv store or from its own pass, chunk lengths, split-K on / off,
# first- / second-generation weight gradient -- every op of each within 2e-5 of float64) moves the first conv's gradient by 4.1e-3 ..
# 7.5e-3 of its largest element and the worst gradient norm by 0.8e-3 .. 1.5e-3, i.e. 3.4x .. 6.1x and 2.4x .. 4.3x the ONE
# fp32-vs-fp64 sample the generator stored (ReLU / max-pool decisions flip): a bound of 4x one noise sample was inside the noise.
# 8x still separates rounding scatter (<= 7.5e-3) from structural errors (O(1)).
LOSS_REL, BASE_REL, COND_FACTOR, NORM_FLOOR = 2e-5, 2e-5, 8.0, 1e-5


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "steps.npz")), json.load(open(os.path.join(golden_dir, "steps_names.json")))


def _check(net, loss, tag, gold):
    z, names = gold
    ref_loss = float(z[tag + "_loss"])
    assert abs(loss.item() - ref_loss) <= LOSS_REL * abs(ref_loss), "loss %.8g vs reference %.8g" % (loss.item(), ref_loss)
    params = dict(net.named_parameters())
    assert list(params) == names[tag]["params"]                      # same parameters, same order as the reference class
    worst = (0.0, None, 0.0)
    norms_ref = z[tag + "_grad_norms"]
    floor = NORM_FLOOR * float(norms_ref.max())
    step_cond = float(z[tag + "_norm_cond"][norms_ref > floor].max())      # the step's noise scale (see the module docstring)
    for n, ref in zip(names[tag]["params"], norms_ref):
        g = params[n].grad
        if ref < 0:
            assert g is None, "%s has a gradient; the reference leaves it None" % n
            continue
        assert g is not None, "%s has no gradient" % n
        assert torch.isfinite(g).all()
        got = float(g.double().norm())
        rel_tol = max(BASE_REL, COND_FACTOR * step_cond) if ref > floor else 0.0
        if ref > floor:
            worst = max(worst, (abs(got - ref) / ref, n, rel_tol))
        assert abs(got - ref) <= rel_tol * ref + floor, \
            "|grad %s| = %.6g vs reference %.6g (rel %.2e, allowed %.2e + floor %.1e)" % (n, got, ref, abs(got - ref) / (ref + 1e-30), rel_tol, floor)
    for k, n in enumerate(names[tag]["full"]):
        ref = z["%s_grad%d" % (tag, k)].astype(np.float64)
        tol = max(BASE_REL, COND_FACTOR * float(z["%s_grad%d_cond" % (tag, k)]))
        got = params[n].grad.detach().cpu().double().numpy()
        err = np.abs(got - ref).max() / np.abs(ref).max()
        assert err <= tol, "grad %s: max err / max|g| = %.3e, allowed %.3e (COND_FACTOR x the reference's own fp32-vs-fp64 %.1e)" % (
            n, err, tol, float(z["%s_grad%d_cond" % (tag, k)]))
    print("%s: loss %.8g (reference %.8g); worst gradient-norm deviation %.2e at %s (allowed %.2e)" % (tag, loss.item(), ref_loss, worst[0], worst[1], worst[2]))
    return worst


def test_sff_fusion_step_matches_reference(gold):
    """main_fusion.py:227-251: frozen FusionNet -> flow -> back-warp of the first three channels -> UNet (train) -> L1 -> backward.
    The flow and the warp are compared with the reference's on their own; the U-Net step then runs on the reference's warped
    frames, so that its gradients are compared on identical inputs."""
    z, _ = gold
    flow_net = SffFusionNet(6, 2, 32).eval(); fill_(flow_net, SEED + 7); flow_net.cuda()
    net = SffUNet(6, 1).train(); fill_(net, SEED + 6); net.cuda()
    warp = SpatialTransformation(use_gpu=True)
    inp = input_for(SEED, "step_in", (2, 6, 64, 64)).cuda(); target = input_for(SEED, "step_tg", (2, 1, 64, 64)).cuda()
    with torch.no_grad():
        flow = flow_net(inp)
    ref_flow = z["sff_fusion_flow"].astype(np.float64)                      # up to +-27.6 px
    dflow = np.abs(flow[:, :, ::4, ::4].cpu().double().numpy() - ref_flow).max()
    assert dflow <= 2e-4 * np.abs(ref_flow).max(), "flow differs by %.2e px" % dflow      # 50 fused conv layers, as test_models_gpu
    warped = warp(inp[:, :3].detach(), flow.permute(0, 2, 3, 1))
    ref_w = torch.from_numpy(z["sff_fusion_warped"]).cuda()
    # uniform-noise frames: neighbouring pixels differ by up to 1, so a flow off by d px moves a bilinear sample by up to ~d
    assert (warped - ref_w).abs().max().item() <= 2.0 * 2e-4 * np.abs(ref_flow).max() + 1e-6
    x = inp.clone(); x[:, :3] = ref_w
    loss = F.l1_loss(net(x), target)
    loss.backward()
    _check(net, loss, "sff_fusion", gold)
    rm = net.state_dict()["conv_encode1.1.running_mean"].cpu().double().numpy()
    assert np.abs(rm - z["sff_fusion_bn_running_mean"]).max() <= 2e-5 * np.abs(z["sff_fusion_bn_running_mean"]).max() + 1e-7


def test_sff_ifnet_step_matches_reference(gold):
    """main_ms.py:187-206: IFNet -> L1 -> backward, through the in-place additive skips and the sepconv gradients."""
    net = SffIFNet(51).train(); fill_(net, SEED); net.cuda()
    x = input_for(SEED, "ifstep_in", (1, 6, 64, 64)).cuda(); target = input_for(SEED, "ifstep_tg", (1, 1, 64, 64)).cuda()
    loss = F.l1_loss(net(x), target)
    loss.backward()
    _check(net, loss, "sff_ifnet", gold)


@pytest.mark.parametrize("which", ["sp_unet", "sp_fusionnet"])
def test_sp_unet_and_fusionnet_step_match_reference(which, gold):
    if which == "sp_unet":
        net = networks.UNet(1, 1).train(); fill_(net, SEED + 2); net.cuda()
        out = net(input_for(SEED, "spu_in", (2, 1, 64, 64)).cuda()); tg = input_for(SEED, "spu_tg", (2, 1, 64, 64)).cuda()
    else:
        net = networks.FusionNet(1, 1).train(); fill_(net, SEED + 3); net.cuda()
        out = net(input_for(SEED, "spf_a", (2, 1, 64, 64)).cuda(), input_for(SEED, "spf_b", (2, 1, 64, 64)).cuda())
        tg = input_for(SEED, "spf_tg", (2, 1, 64, 64)).cuda()
    loss = F.l1_loss(out, tg)
    loss.backward()
    _check(net, loss, which, gold)


def test_sp_joint_step_matches_reference(gold):
    """sp_scripts_train/main_fusion.py:178-257: IFNet twice on the same input, UNet on the two degraded frames, FusionNet on
    the mask-weighted pairs, six L1 losses summed, ONE backward through all three nets (sepconv gradients + U-Nets together).
    The IFNet's dead kernel heads get no gradient, as in the reference."""
    z, _ = gold
    vfi = networks.IFNet().train(); fill_(vfi, SEED + 1)
    den = networks.UNet(1, 1).train(); fill_(den, SEED + 2)
    fus = networks.FusionNet(1, 1).train(); fill_(fus, SEED + 3)
    net = torch.nn.ModuleDict({"vfi": vfi, "den": den, "fus": fus}).cuda()
    im = [input_for(SEED, "spj_im%d" % k, (2, 1, 64, 64)).cuda() for k in range(6)]
    mk = [(input_for(SEED, "spj_mask%d" % k, (2, 1, 64, 64)) > 0.5).float().cuda() for k in range(2)]
    inputs_vfi = torch.cat((im[0], im[0], im[0], im[5], im[5], im[5]), 1)
    v1 = torch.unsqueeze(net["vfi"](inputs_vfi)[:, 0], 1)
    v2 = torch.unsqueeze(net["vfi"](inputs_vfi)[:, 1], 1)
    d1 = net["den"](im[2]); d2 = net["den"](im[4])
    p1 = net["fus"](torch.mul(v1, 1 - mk[0]), torch.mul(d1, mk[0]))
    p2 = net["fus"](torch.mul(v2, 1 - mk[1]), torch.mul(d2, mk[1]))
    ls = [F.l1_loss(v1, im[1]), F.l1_loss(v2, im[3]), F.l1_loss(d1, im[1]), F.l1_loss(d2, im[3]), F.l1_loss(p1, im[1]), F.l1_loss(p2, im[3])]
    for got, ref, what in zip(ls, z["sp_joint_losses"], ("vfi1", "vfi2", "denoise1", "denoise2", "fusion1", "fusion2")):
        assert abs(got.item() - ref) <= LOSS_REL * abs(ref), "loss %s: %.8g vs reference %.8g" % (what, got.item(), ref)
    loss = (ls[0] + ls[2] + ls[4]) + (ls[1] + ls[3] + ls[5])
    loss.backward()
    _check(net, loss, "sp_joint", gold)


def test_sff_ifnet_step_bf16_operands_matches_reference_emulation(golden_dir):
    """BASELINE config 5 arithmetic ("bf16 activations with fp32 sepconv accumulate"), PINNED: the reference IFNet class ran this
    step on CPU with every 3x3 convolution's operands rounded to bf16 exactly where the opt-in id SSTEM_CONV_MFMA_BF16 rounds
    them (forward: x, w; data gradient: g, w; weight gradient: x, g; fp32 sums; everything else fp32) --
    tests/golden/make_bf16_step_golden.py.  Two correct implementations differ by summation order, which near a bf16 rounding
    boundary flips an intermediate value by one bf16 ulp: the generator measured that sensitivity (same rounding points, fp64
    sums: loss 2.4e-4, gradient norms up to 1.5e-2) and the tolerances here are 4 x those, floor 2e-5 -- the derivation the
    fp32 step tests use.  The fp32 step golden is 2.2e-4 (loss) / 1.4e-2 (norms) away: what the id costs, for the record."""
    import hipnn.functional as HF
    z = np.load(os.path.join(golden_dir, "steps_bf16.npz")); names = json.load(open(os.path.join(golden_dir, "steps_bf16_names.json")))
    tag = "sff_ifnet_bf16"
    net = SffIFNet(51).train(); fill_(net, SEED); net.cuda()
    x = input_for(SEED, "ifstep_in", (1, 6, 64, 64)).cuda(); target = input_for(SEED, "ifstep_tg", (1, 1, 64, 64)).cuda()
    with HF.algorithm(HF.ALGO_MFMA_BF16):
        loss = F.l1_loss(net(x), target)
        loss.backward()
    ref_loss = float(z[tag + "_loss"])
    loss_tol = max(LOSS_REL, 4.0 * abs(ref_loss - float(z[tag + "_loss64"])) / abs(ref_loss))
    assert abs(loss.item() - ref_loss) <= loss_tol * abs(ref_loss), "loss %.8g vs reference emulation %.8g (allowed %.1e)" % (loss.item(), ref_loss, loss_tol)
    params = dict(net.named_parameters())
    assert list(params) == names[tag]["params"]
    norms_ref = z[tag + "_grad_norms"]
    floor = NORM_FLOOR * float(norms_ref.max())
    step_cond = float(z[tag + "_norm_cond"][norms_ref > floor].max())
    tol = max(BASE_REL, 4.0 * step_cond)
    worst = 0.0
    for n, ref in zip(names[tag]["params"], norms_ref):
        g = params[n].grad
        if ref < 0:
            assert g is None
            continue
        assert g is not None and torch.isfinite(g).all()
        got = float(g.double().norm())
        if ref > floor:
            worst = max(worst, abs(got - ref) / ref)
        assert abs(got - ref) <= (tol * ref if ref > floor else 0.0) + floor, "|grad %s| = %.6g vs %.6g" % (n, got, ref)
    for k, n in enumerate(names[tag]["full"]):
        ref = z["%s_grad%d" % (tag, k)].astype(np.float64)
        t = max(BASE_REL, 4.0 * float(z["%s_grad%d_cond" % (tag, k)]))
        err = np.abs(params[n].grad.detach().cpu().double().numpy() - ref).max() / np.abs(ref).max()
        assert err <= t, "grad %s: %.3e allowed %.3e" % (n, err, t)
    print("bf16 IFNet step: loss %.8g (reference emulation %.8g, allowed %.1e); worst gradient-norm deviation %.2e (allowed %.2e)"
          % (loss.item(), ref_loss, loss_tol, worst, tol))


def test_sp_joint_step_single_interpolation_pass_equals_the_two_pass_dataflow():
    """steps.SPJointStep(single_vfi_pass=True): the reference evaluates the interpolation net twice on the same input and takes one
    channel of each evaluation (sp_scripts_train/main_fusion.py:213-214).  One evaluation gives both: the same loss bit for bit (the same
    forward launches on the same values) and the same gradients up to the order in which the two channels' contributions are added
    (1e-5 of each bucket's largest gradient allowed; the interpolation net's bucket is where they differ)."""
    import steps
    dev = torch.device("cuda")
    two = steps.SPJointStep(dev, global_batch=2, size=64, overlap=False)
    one = steps.SPJointStep(dev, global_batch=2, size=64, overlap=False, single_vfi_pass=True)      # same seed: same weights, same data
    two.forward_backward(); one.forward_backward()
    torch.cuda.synchronize()
    assert one.loss.item() == two.loss.item()
    for a, b in zip(one.buckets, two.buckets):
        scale = b.flat.abs().max().item()
        assert scale > 0 and (a.flat - b.flat).abs().max().item() <= 1e-5 * scale
    assert torch.equal(one.buckets[2].flat, two.buckets[2].flat)        # the fusion net sees identical inputs and gradients
