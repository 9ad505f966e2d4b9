"""CPU twin of the metric's literal path -- "interp + fusion fwd" -- for `bench.py`'s cpu_baseline leg and the tests ONLY.

BASELINE.md 3(ii): the reference delegates the arithmetic of its networks to stock ``torch.nn`` modules and has no CPU path for the
sepconv op; the CPU baseline of the worded metric is therefore "the model classes on torch CPU ops + the CPU restatement of the op".
The product classes refuse CPU tensors (every block is a native launch), so this file executes THEIR module trees -- the children of
every ``hipnn.FusedSequential`` are the stock ``nn.Conv2d / BatchNorm2d / ReLU / ...`` the reference builds, with the reference's
state_dict keys -- through ``torch.nn.Sequential.forward`` on the CPU and restates the few lines of dataflow between the blocks:

    IFNet        sff_scripts_interp/model/model_interp.py:55-107   (sepconv = the oracle, oracle/sepconv_c.py)
    FusionNet    sff_scripts_fusion/model/model_fusionnet.py:45-62,115-145
    UNet         sff_scripts_fusion/model/model_unet.py:76-104
    chain        sff_scripts_interp/inference_singleImage.py:55-71 + sff_scripts_fusion/inference.py:126-153  (warp = oracle/warp_numpy.py)

Pinned by tests/test_oracle.py against tests/golden/sff_chain.npz (the chain composed from the REFERENCE classes).  Test / benchmark
infrastructure: nothing under sstem-restoration_amd/ imports it.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def _seq(block, x):
    """The children of a FusedSequential, run as the nn.Sequential the reference builds."""
    return nn.Sequential.forward(block, x)


def ifnet(net, x, sepconv_forward):
    """model_interp.py:55-107 on CPU tensors; sepconv_forward(padded [B,3,H+50,W+50], vertical, horizontal) -> [B,3,H,W] (numpy)."""
    i1, i2 = x[:, :3], x[:, 3:6]
    pool = net.pool
    x = _seq(net.conv32, x); x = pool(x)
    x64 = _seq(net.conv64, x); x = pool(x64)
    x128 = _seq(net.conv128, x); x = pool(x128)
    x256 = _seq(net.conv256, x); x = pool(x256)
    x512 = _seq(net.conv512, x); x = pool(x512)
    x = _seq(net.conv512x512, x)
    x = _seq(net.upsamp512, x) + x512
    x = _seq(net.upconv256, x)
    x = _seq(net.upsamp256, x) + x256
    x = _seq(net.upconv128, x)
    x = _seq(net.upsamp128, x) + x128
    x = _seq(net.upconv64, x)
    x = _seq(net.upsamp64, x) + x64
    k2h = _seq(net.upconv51_1, x); k2v = _seq(net.upconv51_2, x)
    k1h = _seq(net.upconv51_3, x); k1v = _seq(net.upconv51_4, x)
    p2 = net.pad(i2).contiguous().numpy(); p1 = net.pad(i1).contiguous().numpy()
    y = sepconv_forward(p2, k2v.contiguous().numpy(), k2h.contiguous().numpy()) + \
        sepconv_forward(p1, k1v.contiguous().numpy(), k1h.contiguous().numpy())
    return torch.from_numpy(y.mean(axis=1, keepdims=True).astype(np.float32))


def _residual_block(blk, x):
    """Conv_residual_conv (model_fusionnet.py:45-62): conv_1 -> three-conv block -> add -> conv_3."""
    head = _seq(blk.conv_1, x)
    b3 = blk.conv_2
    t = _seq(b3[1], _seq(b3[0], head))
    t = b3[3](b3[2](t))
    return _seq(blk.conv_3, t + head)


def fusionnet(net, x):
    """model_fusionnet.py:115-145 (four encoder levels, bridge, four decoder levels averaging with the skips)."""
    skips = []
    for k in range(1, net.LEVELS + 1):
        s = _residual_block(getattr(net, "down_%d" % k), x)
        skips.append(s)
        x = getattr(net, "pool_%d" % k)(s)
    x = _residual_block(net.bridge, x)
    for k in range(1, net.LEVELS + 1):
        d = _seq(getattr(net, "deconv_%d" % k), x)
        x = _residual_block(getattr(net, "up_%d" % k), (d + skips[-k]) / 2)
    return net.out(x)


def unet(net, x):
    """model_unet.py:76-104 (up-sampled tensor first in every concatenation)."""
    e1 = _seq(net.conv_encode1, x)
    e2 = _seq(net.conv_encode2, net.conv_maxpool1(e1))
    e3 = _seq(net.conv_encode3, net.conv_maxpool2(e2))
    b = _seq(net.bottleneck, net.conv_maxpool3(e3))
    d2 = _seq(net.conv_decode3, torch.cat((b, e3), 1))
    d1 = _seq(net.conv_decode2, torch.cat((d2, e2), 1))
    return _seq(net.final_layer, torch.cat((d1, e1), 1))


@torch.no_grad()
def restore_sff(models, prev, nxt, sff, sepconv_forward, warp):
    """sff_pipeline.restore_sff on CPU tensors: (pred, interp, flow, warped).  warp(image [B,C,H,W], flow [B,2,H,W]; numpy) -> [B,C,H,W]
    (oracle/warp_numpy.warp: channel 0 = dx, 1 = dy, as the reference's SpatialTransformation reads the permuted flow)."""
    B, _, H, W = prev.shape
    x = torch.cat((prev.expand(B, 3, H, W), nxt.expand(B, 3, H, W)), 1)
    interp = ifnet(models["interp"], x, sepconv_forward)
    inputs = torch.cat((sff.expand(B, 3, H, W), interp.expand(B, 3, H, W)), 1).contiguous()
    flow = fusionnet(models["flow"], inputs)
    warped = torch.from_numpy(np.ascontiguousarray(warp(inputs[:, :3].contiguous().numpy(), flow.contiguous().numpy())))
    inputs[:, :3] = warped
    pred = unet(models["fusion"], inputs)
    return pred, interp, flow, warped
