"""Generates tests/golden/steps.npz: ONE training step of the reference's model classes on CPU (this container only),
restating the step shapes of SURVEY 8(a) a13 at a reduced size:

  sff_fusion : sff_scripts_fusion/main_fusion.py:227-251 -- frozen FusionNet (eval, no grad) -> flow, NHWC -> the first three
               input channels are back-warped by it (utils/image_warp_torch.py) -> UNet (train mode: batch statistics) ->
               L1 loss -> backward.  PAD = 0 (config: TRAIN.pad 0), no weight decay.
  sff_ifnet  : sff_scripts_interp/main_ms.py:187-206 -- IFNet -> L1 loss -> backward (the sepconv op and its gradients are
               the CPU oracle: the reference has no CPU implementation of it).
  sp_unet, sp_fusionnet : sp_scripts_train/networks.py classes in train mode -> L1 -> backward.
  sp_joint   : sp_scripts_train/main_fusion.py:178-257 (config: loss_type L1, if_fusion_loss_only False, PAD 0) -- the
               interpolation IFNet runs TWICE on cat(img_1 x3, img_4 x3) (channel 0 of the first pass, channel 1 of the second),
               the denoising UNet on the two degraded frames, the FusionNet on mask-weighted pairs (interp * (1 - mask),
               denoised * mask); six L1 losses against img_2 / img_3, summed; one backward through all three nets -- the only
               step that runs the sepconv gradients together with the U-Nets.

Conditioning.  Each step is also run in float64 (same classes, same fp32 inputs and weights cast up): the forward is
well-conditioned (losses agree to 1e-8), the gradients of the BatchNorm-bearing nets are not -- the reference's OWN fp32 and
fp64 gradients differ by up to 5e-4 in norm and up to 1.8e-2 elementwise (ReLU / max-pool decisions and BN cancellation),
while the BN-free IFNet agrees to 1e-6.  The per-quantity deviations are stored (`*_norm_cond`, `*_grad{k}_cond`) and the
GPU test derives its tolerances from them instead of picking one.  For the fusion step the U-Net input (the warped
frames) is stored in full so that the U-Net gradients are compared on identical inputs; the flow and the warp are
compared separately.

Stored per step: the loss, the L2 norm of every parameter's gradient (state-dict order of named_parameters; -1 where the
reference leaves the gradient None), two small gradients in full and, for the fusion step, a BatchNorm running mean after
the step's forward.  Weights and inputs come from tests/weight_recipe.py on both sides.

One intervention, arithmetic-neutral, is needed to run the reference IFNet's backward on this container's torch 2.10:
model_interp.py:74-83 adds its skips in place (`x += x512`) onto the output of `nn.ReLU(inplace=False)` (:19).  torch 0.4,
which the reference targets, differentiated ReLU through its INPUT, so overwriting the output was legal; torch 2.x keeps the
OUTPUT and raises "modified by an inplace operation".  The IFNet is therefore constructed with `torch.nn.ReLU` temporarily
replaced by a module with the same forward kernel (`torch.relu`) whose backward takes its mask from the input
(`grad * (x > 0)`, the same mask as `result > 0`): same forward values, same gradients, same module tree and state-dict
keys.  The build's own IFNet handles the in-place skips natively (hipnn saves the sign mask); nothing here is shipped.

Run from the repo root:  python tests/golden/make_step_goldens.py
"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_model_goldens as G  # noqa: E402  (stubs + loader of the reference files; also puts tests/ on the path)
from weight_recipe import fill_, input_for  # noqa: E402

SEED = 555


class _ReluFromInput(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.relu(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return g * (x > 0).to(g.dtype)


class InputMaskReLU(torch.nn.Module):
    def __init__(self, inplace=False):
        super().__init__()
        assert not inplace

    def forward(self, x):
        return _ReluFromInput.apply(x)


def grads_of(net, full):
    names, norms, out = [], [], {}
    for n, p in net.named_parameters():
        names.append(n)
        norms.append(-1.0 if p.grad is None else float(p.grad.double().norm()))
    for n in full:
        out[n] = dict(net.named_parameters())[n].grad.numpy().copy()
    return names, np.asarray(norms, np.float64), out


def record(out, names, tag, build, run, full):
    """build() -> a fresh reference net (fp32, recipe weights); run(net, dtype) -> loss.  Runs the step in fp32 (the golden)
    and in fp64 (the conditioning), stores both."""
    net = build()
    loss = run(net, torch.float32)
    loss.backward()
    nm, norms, g = grads_of(net, full)
    net64 = build().double()
    loss64 = run(net64, torch.float64)
    loss64.backward()
    p64 = dict(net64.named_parameters())
    cond = np.zeros_like(norms)
    for i, n in enumerate(nm):
        if norms[i] >= 0:
            cond[i] = abs(norms[i] - float(p64[n].grad.norm())) / (float(p64[n].grad.norm()) + 1e-300)
    names[tag] = {"params": nm, "full": full}
    out[tag + "_loss"] = np.float64(loss.item())
    out[tag + "_loss64"] = np.float64(loss64.item())
    out[tag + "_grad_norms"] = norms
    out[tag + "_norm_cond"] = cond
    for k, n in enumerate(full):
        out["%s_grad%d" % (tag, k)] = g[n]
        ref64 = p64[n].grad.numpy()
        out["%s_grad%d_cond" % (tag, k)] = np.float64(np.abs(g[n].astype(np.float64) - ref64).max() / np.abs(ref64).max())
    return net


def main():
    torch.set_num_threads(8)
    G.install_stubs()
    out, names = {}, {}

    # ---- SFF fusion step
    mf = G.load_ref("sff_scripts_fusion/model/model_fusionnet.py", "ref_model_fusionnet")
    mu = G.load_ref("sff_scripts_fusion/model/model_unet.py", "ref_model_unet")
    mw = G.load_ref("sff_scripts_fusion/utils/image_warp_torch.py", "ref_warp")
    flow_net = mf.FusionNet(input_nc=6, output_nc=2, ngf=32).eval(); fill_(flow_net, SEED + 7)
    warp = mw.SpatialTransformation(use_gpu=False)
    inp = input_for(SEED, "step_in", (2, 6, 64, 64)); target = input_for(SEED, "step_tg", (2, 1, 64, 64))
    with torch.no_grad():
        flow = flow_net(inp)
    warped = warp(inp[:, :3].detach(), flow.permute(0, 2, 3, 1))
    x = inp.clone(); x[:, :3] = warped

    def build_unet():
        n = mu.UNet(in_channel=6, out_channel=1).train(); fill_(n, SEED + 6); return n
    net = record(out, names, "sff_fusion", build_unet, lambda n, dt: F.l1_loss(n(x.to(dt)), target.to(dt)),
                 ["conv_encode1.0.weight", "final_layer.3.weight"])
    out["sff_fusion_flow"] = flow[:, :, ::4, ::4].contiguous().numpy()
    out["sff_fusion_warped"] = warped.numpy().copy()          # in full: the U-Net input of the GPU test
    out["sff_fusion_bn_running_mean"] = net.state_dict()["conv_encode1.1.running_mean"].numpy().copy()

    # ---- SFF IFNet step
    mi = G.load_ref("sff_scripts_interp/model/model_interp.py", "ref_model_interp_step")

    def build_ifnet():
        real_relu = torch.nn.ReLU
        torch.nn.ReLU = InputMaskReLU          # see the module docstring; restored right after construction
        try:
            n = mi.IFNet(kernel_size=51).train()
        finally:
            torch.nn.ReLU = real_relu
        fill_(n, SEED)
        return n
    xi = input_for(SEED, "ifstep_in", (1, 6, 64, 64)); ti = input_for(SEED, "ifstep_tg", (1, 1, 64, 64))
    # (the oracle-bound sepconv stub computes in fp32 in both runs: the op is the oracle's, not torch's)
    record(out, names, "sff_ifnet", build_ifnet, lambda n, dt: F.l1_loss(n(xi.to(dt)).to(dt), ti.to(dt)),
           ["conv32.0.weight", "upconv51_1.7.bias"])

    # ---- SP UNet / FusionNet, train mode
    mn = G.load_ref("sp_scripts_train/networks.py", "ref_networks_step")
    xu = input_for(SEED, "spu_in", (2, 1, 64, 64)); tu = input_for(SEED, "spu_tg", (2, 1, 64, 64))

    def build_spunet():
        n = mn.UNet(1, 1).train(); fill_(n, SEED + 2); return n
    record(out, names, "sp_unet", build_spunet, lambda n, dt: F.l1_loss(n(xu.to(dt)), tu.to(dt)),
           ["inc.double_conv.0.weight", "outc.conv.weight"])
    fa = input_for(SEED, "spf_a", (2, 1, 64, 64)); fb = input_for(SEED, "spf_b", (2, 1, 64, 64)); ft = input_for(SEED, "spf_tg", (2, 1, 64, 64))

    def build_spfus():
        n = mn.FusionNet(1, 1).train(); fill_(n, SEED + 3); return n
    record(out, names, "sp_fusionnet", build_spfus, lambda n, dt: F.l1_loss(n(fa.to(dt), fb.to(dt)), ft.to(dt)),
           ["inc.double_conv.0.weight", "outc.conv.weight"])

    # ---- SP joint step: three nets in one container so that one backward fills every gradient
    im = [input_for(SEED, "spj_im%d" % k, (2, 1, 64, 64)) for k in range(6)]      # img_1, img_2, img_2_degra, img_3, img_3_degra, img_4
    mk = [(input_for(SEED, "spj_mask%d" % k, (2, 1, 64, 64)) > 0.5).float() for k in range(2)]
    parts = []

    def build_joint():
        real_relu = torch.nn.ReLU
        torch.nn.ReLU = InputMaskReLU          # the SP IFNet has the same in-place skips (networks.py:19,93-102)
        try:
            vfi = mn.IFNet().train()
        finally:
            torch.nn.ReLU = real_relu
        fill_(vfi, SEED + 1)
        den = mn.UNet(1, 1).train(); fill_(den, SEED + 2)
        fus = mn.FusionNet(1, 1).train(); fill_(fus, SEED + 3)
        return torch.nn.ModuleDict({"vfi": vfi, "den": den, "fus": fus})

    def run_joint(n, dt):
        i = [t.to(dt) for t in im]; m = [t.to(dt) for t in mk]
        inputs_vfi = torch.cat((i[0], i[0], i[0], i[5], i[5], i[5]), 1)
        v1 = torch.unsqueeze(n["vfi"](inputs_vfi)[:, 0], 1)
        v2 = torch.unsqueeze(n["vfi"](inputs_vfi)[:, 1], 1)
        d1 = n["den"](i[2]); d2 = n["den"](i[4])
        p1 = n["fus"](torch.mul(v1, 1 - m[0]), torch.mul(d1, m[0]))
        p2 = n["fus"](torch.mul(v2, 1 - m[1]), torch.mul(d2, m[1]))
        ls = [F.l1_loss(v1, i[1]), F.l1_loss(v2, i[3]), F.l1_loss(d1, i[1]), F.l1_loss(d2, i[3]), F.l1_loss(p1, i[1]), F.l1_loss(p2, i[3])]
        parts.append([float(x.item()) for x in ls])
        return (ls[0] + ls[2] + ls[4]) + (ls[1] + ls[3] + ls[5])
    record(out, names, "sp_joint", build_joint, run_joint,
           ["vfi.conv32.0.weight", "den.inc.double_conv.0.weight", "fus.outc.conv.weight"])
    out["sp_joint_losses"] = np.asarray(parts[0], np.float64)      # vfi1, vfi2, denoise1, denoise2, fusion1, fusion2 (fp32 run)

    np.savez_compressed(os.path.join(HERE, "steps.npz"), **out)
    with open(os.path.join(HERE, "steps_names.json"), "w") as f:
        json.dump(names, f, indent=0)
    for k, v in out.items():
        v = np.asarray(v)
        print("%-30s %-16s absmax %.5g" % (k, v.shape, np.abs(v).max()))
    print("steps.npz", os.path.getsize(os.path.join(HERE, "steps.npz")), "bytes")


if __name__ == "__main__":
    main()
