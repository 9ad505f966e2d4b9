"""Generates tests/golden/steps_bf16.npz: the SFF IFNet training step (sff_scripts_interp/main_ms.py:187-206) of the REFERENCE
class under the arithmetic of BASELINE config 5 ("bf16 activations with fp32 sepconv accumulate"), on CPU, this container only.

The reference has no bf16 path, so the reduced-precision arithmetic is stated here as an emulation around the reference's own
modules: while the step runs, every nn.Conv2d(3x3, stride 1, padding 1) of the reference IFNet goes through a function that
does what the build's opt-in id SSTEM_CONV_MFMA_BF16 does (include/sstem_conv.h) --

    forward        conv2d(bf16(x), bf16(w)) + b        operands rounded to nearest-even bf16, exact products, fp32 sums
    data gradient  conv_transpose(bf16(g), bf16(w))
    weight grad    correlate(bf16(x), bf16(g))
    bias grad      sum of the fp32 g

-- everything else (ReLU, pooling, bilinear up-sampling, the additive skips, the sepconv op = the fp32 CPU oracle, the L1 loss)
is the reference's fp32 arithmetic.  Same recipe weights and inputs as the fp32 step golden (steps.npz, tag sff_ifnet).

Conditioning: a rounded operand sits at most half a bf16 ulp from a rounding boundary, so two correct implementations that sum
in different orders can round a few intermediate activations differently.  The step is therefore also run with the SAME
rounding points but float64 sums; the deviation of the fp32-sum run from it (loss, every gradient norm, two gradients in full)
is stored, and the GPU test takes its tolerance from it (4 x the step's largest deviation, floor 2e-5), exactly as the fp32
step tests do.  Also stored: how far this step is from the fp32 step golden (what the id costs; information, not a bound).

Run from the repo root:  python tests/golden/make_bf16_step_golden.py
"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_model_goldens as G  # noqa: E402
import make_step_goldens as S  # noqa: E402
from weight_recipe import fill_, input_for  # noqa: E402

SEED = 555


def q(t):
    """round to nearest-even bf16, back in the tensor's own dtype"""
    return t.to(torch.float32).to(torch.bfloat16).to(t.dtype)


class ConvBf16Operands(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return F.conv2d(q(x), q(w), b, padding=1)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        gx = torch.nn.grad.conv2d_input(x.shape, q(w), q(g), padding=1)
        gw = torch.nn.grad.conv2d_weight(q(x), w.shape, q(g), padding=1)
        return gx, gw, (g.sum((0, 2, 3)) if ctx.has_bias else None)


class bf16_conv_operands:
    """with-block: nn.Conv2d(3x3, s1, p1) forward = ConvBf16Operands"""

    def __enter__(self):
        self.orig = torch.nn.Conv2d._conv_forward

        def patched(mod, inp, weight, bias):
            if mod.kernel_size == (3, 3) and mod.stride == (1, 1) and mod.padding == (1, 1) and mod.dilation == (1, 1) and mod.groups == 1:
                return ConvBf16Operands.apply(inp, weight, bias)
            return self.orig(mod, inp, weight, bias)
        torch.nn.Conv2d._conv_forward = patched

    def __exit__(self, *exc):
        torch.nn.Conv2d._conv_forward = self.orig
        return False


def main():
    torch.set_num_threads(8)
    G.install_stubs()
    mi = G.load_ref("sff_scripts_interp/model/model_interp.py", "ref_model_interp_bf16step")

    def build():
        real_relu = torch.nn.ReLU
        torch.nn.ReLU = S.InputMaskReLU          # see make_step_goldens.py: arithmetic-neutral, needed on torch >= 2
        try:
            n = mi.IFNet(kernel_size=51).train()
        finally:
            torch.nn.ReLU = real_relu
        fill_(n, SEED)
        return n
    xi = input_for(SEED, "ifstep_in", (1, 6, 64, 64)); ti = input_for(SEED, "ifstep_tg", (1, 1, 64, 64))
    full = ["conv32.0.weight", "upconv51_1.7.bias"]
    out, names = {}, {}

    def run(dtype):
        net = build().to(dtype)
        with bf16_conv_operands():
            loss = F.l1_loss(net(xi.to(dtype)).to(dtype), ti.to(dtype))
            loss.backward()
        return net, loss

    net, loss = run(torch.float32)
    nm, norms, g = S.grads_of(net, full)
    net64, loss64 = run(torch.float64)
    p64 = dict(net64.named_parameters())
    cond = np.zeros_like(norms)
    for i, n in enumerate(nm):
        if norms[i] >= 0:
            cond[i] = abs(norms[i] - float(p64[n].grad.norm())) / (float(p64[n].grad.norm()) + 1e-300)
    names["sff_ifnet_bf16"] = {"params": nm, "full": full}
    out["sff_ifnet_bf16_loss"] = np.float64(loss.item())
    out["sff_ifnet_bf16_loss64"] = np.float64(loss64.item())
    out["sff_ifnet_bf16_grad_norms"] = norms
    out["sff_ifnet_bf16_norm_cond"] = cond
    for k, n in enumerate(full):
        out["sff_ifnet_bf16_grad%d" % k] = g[n]
        ref64 = p64[n].grad.numpy()
        out["sff_ifnet_bf16_grad%d_cond" % k] = np.float64(np.abs(g[n].astype(np.float64) - ref64).max() / np.abs(ref64).max())
    # distance from the fp32 step golden (information)
    z = np.load(os.path.join(HERE, "steps.npz"))
    live = z["sff_ifnet_grad_norms"] > 0
    out["sff_ifnet_bf16_vs_fp32_loss_rel"] = np.float64(abs(loss.item() - float(z["sff_ifnet_loss"])) / abs(float(z["sff_ifnet_loss"])))
    out["sff_ifnet_bf16_vs_fp32_norm_rel_max"] = np.float64((np.abs(norms[live] - z["sff_ifnet_grad_norms"][live]) / z["sff_ifnet_grad_norms"][live]).max())
    np.savez_compressed(os.path.join(HERE, "steps_bf16.npz"), **out)
    with open(os.path.join(HERE, "steps_bf16_names.json"), "w") as f:
        json.dump(names, f, indent=0)
    for k, v in out.items():
        v = np.asarray(v)
        print("%-40s %-12s absmax %.6g" % (k, v.shape, np.abs(v).max()))
    print("loss fp32-sum %.9g  fp64-sum %.9g  fp32 golden %.9g" % (loss.item(), loss64.item(), float(z["sff_ifnet_loss"])))
    print("largest norm deviation fp32-sum vs fp64-sum: %.3e" % cond[norms > 1e-5 * norms.max()].max())


if __name__ == "__main__":
    main()
