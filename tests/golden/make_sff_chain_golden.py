"""Generates tests/golden/sff_chain.npz: the headline metric's literal path, "interp + fusion fwd", composed from the REFERENCE
classes the way the reference's two SFF inference scripts compose them (this container only):

    inputs = cat(prev x3, next x3) / 255 ; interp = IFNet(inputs)               sff_scripts_interp/inference_singleImage.py:55-71
    [ interp -> (interp*255).astype(uint8) PNG -> /255 ]                        :76 and sff_scripts_fusion/inference.py:126-136
    inputs = cat(sff x3, interp x3) ; flow = FusionNet(6,2,32)(inputs)          inference.py:126-142
    warped = SpatialTransformation(inputs[:, :3].clone(), flow.permute(0,2,3,1))  :145-150
    inputs[:, :3] = warped ; pred = UNet(6,1)(inputs)                           :152-153

with TEST.pad = 0 (the shipped configs), every net in eval mode, one 64 x 64 tile, two tiles in the batch.  Stored in BOTH
spellings: the chain kept in fp32 (`*_f`) and with the PNG round trip between the two scripts (`*_q`).  The sepconv op inside the
IFNet is the CPU oracle (the reference has no CPU implementation of it; make_model_goldens.py).  Weights: tests/weight_recipe.py --
the IFNet gets the CLI golden's damped kernel heads (interp stays inside [0,1], so the uint8 truncation means something), the flow
net's last convolution is damped (`sff_flow_weights_`: displacements of a few pixels instead of tens) -- both recipes are pure
functions of parameter names, applied identically to the build's classes by the GPU test.

Conditioning: the reference chain is also run in float64 (same classes, weights and inputs cast up; the sepconv oracle stays
fp32 arithmetic on rounded operands, as in the other generators) and `<key>_cond` stores how far the reference's own fp32 result is
from it relative to the largest element; the GPU test derives its tolerance from that.

Run from the repo root:  python tests/golden/make_sff_chain_golden.py
"""
import copy
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_model_goldens as G  # noqa: E402  (stubs + loader of the reference files; also puts tests/ on the path)
from weight_recipe import cli_weights_, fill_, sff_chain_inputs, sff_flow_weights_  # noqa: E402

SEED = 555


def chain(interp_net, flow_net, unet, warp, prev, nxt, sff, dtype, quantise, forced_u8=None):
    """forced_u8: the PNG to read instead of this run's own (the float64 conditioning run reads the fp32 run's file: a pixel that
    truncates differently in float64 is a different input, not rounding noise of the second script)."""
    x = torch.cat((prev, prev, prev, nxt, nxt, nxt), 1).to(dtype)          # np.repeat(img, 3, 0) + concatenate
    interp = interp_net(x)
    u8 = None
    if quantise:          # (pred * 255).astype(np.uint8) -> PNG -> astype(float32) / 255.0
        u8 = (np.squeeze(interp.to(torch.float32).numpy(), 1) * 255).astype(np.uint8) if forced_u8 is None else forced_u8
        interp = torch.from_numpy(u8.astype(np.float32) / 255.0)[:, None].to(dtype)
    inputs = torch.cat((sff, sff, sff, interp, interp, interp), 1).to(dtype)
    flow = flow_net(inputs)
    input_sff = inputs[:, :3].clone()
    warped = warp(input_sff, flow.permute(0, 2, 3, 1))
    inputs[:, :3] = warped
    pred = unet(inputs)
    return {"interp": interp, "flow": flow, "warped": warped, "pred": pred}, u8


def main():
    torch.set_num_threads(8)
    G.install_stubs()
    mi = G.load_ref("sff_scripts_interp/model/model_interp.py", "ref_model_interp_chain")
    mf = G.load_ref("sff_scripts_fusion/model/model_fusionnet.py", "ref_model_fusionnet_chain")
    mu = G.load_ref("sff_scripts_fusion/model/model_unet.py", "ref_model_unet_chain")
    mw = G.load_ref("sff_scripts_fusion/utils/image_warp_torch.py", "ref_warp_chain")
    interp_net = mi.IFNet(kernel_size=51).eval(); cli_weights_(interp_net, SEED + 8)
    flow_net = mf.FusionNet(input_nc=6, output_nc=2, ngf=32).eval(); sff_flow_weights_(flow_net, SEED + 7)
    unet = mu.UNet(in_channel=6, out_channel=1).eval(); fill_(unet, SEED + 6)
    warp = mw.SpatialTransformation(use_gpu=False)
    prev, nxt, sff = (torch.from_numpy(a) for a in sff_chain_inputs(2, 64, 64))
    nets64 = [copy.deepcopy(n).double() for n in (interp_net, flow_net, unet)]
    out = {}
    with torch.no_grad():
        for tag, quantise in (("f", False), ("q", True)):
            r32, u8 = chain(interp_net, flow_net, unet, warp, prev, nxt, sff, torch.float32, quantise)
            r64, _ = chain(*nets64, warp, prev, nxt, sff, torch.float64, quantise, forced_u8=u8)
            for k, v in r32.items():
                out["%s_%s" % (k, tag)] = v.numpy().copy()
                out["%s_%s_cond" % (k, tag)] = np.float64((v.double() - r64[k]).abs().max().item() / max(r64[k].abs().max().item(), 1e-30))
            if u8 is not None:
                out["interp_u8"] = u8
        # how close the fp32 interpolated frame comes to a truncation boundary (in uint8 steps): the pixels whose uint8 value a
        # deviation inside north_star's tolerance could flip
        fr = interp_net(torch.cat((prev, prev, prev, nxt, nxt, nxt), 1)).numpy() * 255.0
        out["interp_u8_margin"] = np.minimum(fr - np.floor(fr), np.ceil(fr) - fr).astype(np.float32)[:, 0]
    np.savez_compressed(os.path.join(HERE, "sff_chain.npz"), **out)
    for k, v in sorted(out.items()):
        if k.endswith("_cond"):
            print("%-18s fp32 vs fp64 of the reference: %.3e of the largest element" % (k, float(v)))
        else:
            print("%-18s %-16s min %.4g max %.4g" % (k, v.shape, v.min(), v.max()))
    print("sff_chain.npz", os.path.getsize(os.path.join(HERE, "sff_chain.npz")), "bytes")


if __name__ == "__main__":
    main()
