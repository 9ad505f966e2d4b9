"""SP full restoration pipeline (interpolation + correction + fusion) for one tile set, and its
tile-sharded multi-GPU driver.

Dataflow of the reference ``sp_scripts_test/test_fusion.py:59-124`` (BASELINE config #4):

    inputs_vfi = cat(im1 x3, im4 x3)                       :105-106
    vfi_pred1, vfi_pred2 = IFNet(inputs_vfi)[:, 0], [:, 1] :107-108  (the reference runs the net twice on
                                                           the same input; one pass gives both channels)
    denoise_k = UNet(degraded_k)                           :110-111
    a_k = vfi_pred_k * mask_r_k ;  b_k = denoise_k * mask_k:113-116
    pred_k = FusionNet(a_k, b_k)                           :120-121
    uint8 = (pred*255).astype('uint8')  (no clamp)         utils/gray2tensor.py:14-20

Tile sets are independent (fully convolutional nets, eval-mode BatchNorm is point-wise): ranks own
tile sets round-robin, weights are broadcast once, there is no per-tile communication.
"""
import numpy as np
import torch

import dataparallel as dp
from networks import FusionNet, IFNet, UNet


def build_models(device):
    return {"vfi": IFNet().eval().to(device), "denoise": UNet(1, 1).eval().to(device),
            "fusion": FusionNet(1, 1).eval().to(device)}


def load_reference_checkpoints(models, model_path):
    """``model_vfi.ckpt / model_denoise.ckpt / model_fusion.ckpt`` with a 'model_weights' dict, no prefix strip
    (test_fusion.py:44-51)."""
    for key, fname in (("vfi", "model_vfi.ckpt"), ("denoise", "model_denoise.ckpt"), ("fusion", "model_fusion.ckpt")):
        ckpt = torch.load(model_path + fname, map_location="cpu")
        models[key].load_state_dict(ckpt["model_weights"])


def gray2tensor(img_u8, device):
    """utils/gray2tensor.py:7-12: uint8 [H,W] -> float32 [1,1,H,W] / 255 on the device."""
    return torch.from_numpy(np.ascontiguousarray(img_u8).astype(np.float32) / 255.0)[None, None].to(device)


def tensor2gray(t):
    """utils/gray2tensor.py:14-20: *255, astype uint8 (truncation, no clamp)."""
    return (np.squeeze(t.detach().cpu().numpy()) * 255).astype("uint8")


def crop32(*imgs):
    """Crop every image to multiples of 32 (test_fusion.py:77-87)."""
    h, w = imgs[0].shape[:2]
    if h % 32 != 0 or w % 32 != 0:
        imgs = tuple(i[:h - h % 32, :w - w % 32] for i in imgs)
    return imgs


@torch.no_grad()
def restore_tile_set(models, im1, im2_degra, im2_mask, im3_degra, im3_mask, im4, vfi_twice=False):
    """All arguments float32 [B,1,H,W] in [0,1] on the GPU (masks: 1 inside the degraded region).
    Returns (pred1, pred2, vfi_pred1, vfi_pred2, denoise1, denoise2)."""
    mask2_r = 1.0 - im2_mask
    mask3_r = 1.0 - im3_mask
    # inputs_vfi = cat(im1 x3, im4 x3) (test_fusion.py:105-106): the planes themselves go to the network's gray entry
    vfi = models["vfi"].interpolate_gray(im1, im4)
    vfi_pred1 = vfi[:, 0:1]
    vfi_pred2 = (models["vfi"].interpolate_gray(im1, im4) if vfi_twice else vfi)[:, 1:2]
    denoise1 = models["denoise"](im2_degra)
    denoise2 = models["denoise"](im3_degra)
    pred1 = models["fusion"](vfi_pred1 * mask2_r, denoise1 * im2_mask)
    pred2 = models["fusion"](vfi_pred2 * mask3_r, denoise2 * im3_mask)
    return pred1, pred2, vfi_pred1, vfi_pred2, denoise1, denoise2


def restore_sharded(models, tile_sets, rank, world):
    """tile_sets: list of 6-tuples of GPU tensors (or callables producing them).  Weights must already be
    identical on every rank (``dataparallel.broadcast_module``).  Returns {index: (pred1, pred2)} for the tile
    sets this rank owns."""
    out = {}
    for idx in dp.shard_indices(len(tile_sets), rank, world):
        ts = tile_sets[idx]
        if callable(ts):
            ts = ts()
        res = restore_tile_set(models, *ts)
        out[idx] = (res[0], res[1])
    return out
