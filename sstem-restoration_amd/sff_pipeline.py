"""SFF restoration forward (interpolation + unfolding flow + fusion) for a batch of tiles, and its tile-sharded driver --
the literal wording of the headline metric, "interp + fusion fwd".

Dataflow of the reference's SFF inference scripts:

    interp  = IFNet(cat(prev x3, next x3))                      sff_scripts_interp/inference_singleImage.py:55-71
    inputs  = cat(sff x3, interp x3)                            sff_scripts_fusion/inference.py:126-136
    flow    = FusionNet(6, 2, 32)(inputs)                       :142  (the unfolding flow predictor, eval)
    warped  = SpatialTransformation(inputs[:, :3], flow.permute(0, 2, 3, 1))    :145-150
    inputs[:, :3] = warped ;  pred = UNet(6, 1)(inputs)         :152-153  (the fusion net, eval)

The reference writes the interpolated frame to a PNG between the two scripts (uint8 truncation, utils of f3); here the chain
stays on the GPU in fp32 -- ``quantise_interp=True`` reproduces the PNG round trip (``(x*255).astype(uint8) / 255``) for callers
that want the files' arithmetic.  Tiles are independent: ranks own tiles round-robin, weights are broadcast once.
"""
import torch

import dataparallel as dp
import sstem_native
from model.model_fusionnet import FusionNet
from model.model_interp import IFNet
from model.model_unet import UNet
from utils.image_warp_torch import SpatialTransformation

_warp = SpatialTransformation(use_gpu=True)


def _png_round_trip(t):
    """(t*255).astype(uint8) / 255 with the native edge kernels (include/sstem_io.h: numpy's truncation, no clamp)."""
    t = t.contiguous()
    u8 = torch.empty(t.shape, dtype=torch.uint8, device=t.device)
    out = torch.empty_like(t)
    lib = sstem_native.load_library()
    with torch.cuda.device(t.device):
        s = torch.cuda.current_stream().cuda_stream
        sstem_native.check(lib.sstem_f32_to_gray_u8(t.data_ptr(), u8.data_ptr(), t.numel(), 0, s), "sstem_f32_to_gray_u8")
        sstem_native.check(lib.sstem_gray_u8_to_f32(u8.data_ptr(), out.data_ptr(), t.numel(), 1, s), "sstem_gray_u8_to_f32")
    return out


def build_models(device):
    return {"interp": IFNet(51).eval().to(device), "flow": FusionNet(6, 2, 32).eval().to(device),
            "fusion": UNet(6, 1).eval().to(device)}


@torch.no_grad()
def interpolate(models, prev, nxt, quantise_interp=False):
    """Stage 1 (inference_singleImage.py:55-76): the interpolated section [B,1,H,W]; ``quantise_interp`` = the PNG the reference
    writes between its two scripts (*255, uint8 truncation without clamp, /255 on reading)."""
    interp = models["interp"].interpolate_gray(prev, nxt)
    return _png_round_trip(interp) if quantise_interp else interp


@torch.no_grad()
def fuse(models, sff, interp):
    """Stage 2 (sff_scripts_fusion/inference.py:126-153): flow on cat(sff x3, interp x3), back-warp of the SFF channels, fusion UNet.
    Returns (pred [B,1,H,W], flow [B,2,H,W], warped_sff [B,3,H,W])."""
    B, _, H, W = sff.shape
    inputs = torch.cat((sff.expand(B, 3, H, W), interp.expand(B, 3, H, W)), 1)
    flow = models["flow"](inputs)
    warped = _warp(inputs[:, :3], flow.permute(0, 2, 3, 1))
    inputs[:, :3] = warped
    pred = models["fusion"](inputs)
    return pred, flow, warped


@torch.no_grad()
def restore_sff(models, prev, nxt, sff, quantise_interp=False):
    """prev, nxt: the neighbouring sections; sff: the folded section; float32 [B,1,H,W] in [0,1] on the GPU, H and W
    multiples of 32.  Returns (pred [B,1,H,W], interp [B,1,H,W], flow [B,2,H,W], warped_sff [B,3,H,W])."""
    interp = interpolate(models, prev, nxt, quantise_interp)
    pred, flow, warped = fuse(models, sff, interp)
    return pred, interp, flow, warped


def restore_sharded(models, tiles, rank, world):
    """tiles: list of (prev, next, sff) GPU tensors (or callables producing them).  Weights must already be identical on every
    rank (``dataparallel.broadcast_module``).  Returns {index: pred} for the tiles this rank owns."""
    out = {}
    for idx in dp.shard_indices(len(tiles), rank, world):
        t = tiles[idx]
        if callable(t):
            t = t()
        out[idx] = restore_sff(models, *t)[0]
    return out
