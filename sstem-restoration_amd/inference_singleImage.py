#!/usr/bin/env python
"""Two grayscale PNGs -> IFNet -> one interpolated PNG, on MI355X.

Counterpart of the reference CLI ``sff_scripts_interp/inference_singleImage.py:22-79``: same flags
(``-c/--cfg -id/--model_id -i1/--img1 -i2/--img2 -o/--output``), same YAML keys (``TRAIN.kernel_size``,
``TEST.pad``), same checkpoint layout (``{'model_weights': {...}}`` with the 7-character ``module.``
prefix stripped unconditionally, ``:42-47``) and the same I/O conversions: each frame replicated to 3
identical channels, ``/255`` to float32, output ``(pred*255).astype(uint8)`` WITHOUT clamping (``:55-76``).
Two optional flags (``--ckpt``, ``--config-dir``) override the reference's hard-wired relative paths.
The network runs on the GPU only: like the reference, a CPU-only host ends in ``NotImplementedError``
from the sepconv op.
"""
import argparse
import os
import sys
import time
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F
import yaml
from PIL import Image

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)

import hipnn.functional as HF  # noqa: E402
from model.model_interp import IFNet  # noqa: E402
from utils.gray2tensor import gray_to_tensor, tensor_to_gray  # noqa: E402


def load_config(cfg_name, config_dir=None):
    path = os.path.join(config_dir or os.path.join(_HERE, "config"), cfg_name + ".yaml")
    with open(path, "r") as f:
        return yaml.safe_load(f)      # the reference's bare yaml.load(f) is an error on PyYAML >= 6


def load_model(cfg, ckpt_path, device):
    model = IFNet(kernel_size=cfg["TRAIN"]["kernel_size"]).to(device)
    checkpoint = torch.load(ckpt_path, map_location="cpu")
    new_state_dict = OrderedDict()
    for k, v in checkpoint["model_weights"].items():
        new_state_dict[k[7:]] = v            # remove "module." exactly as the reference does
    model.load_state_dict(new_state_dict)
    return model.to(device).eval()


def read_pair(img1_path, img2_path, device):
    """The two frames as [1,1,H,W] float32 planes in [0,1].  One uint8 plane per frame goes to the GPU and is divided by
    255 there (utils/gray2tensor); the reference's x3 replication of each frame (:55-61) is implied by
    ``IFNet.interpolate_gray`` and never materialised for the local convolutions."""
    return [gray_to_tensor(np.asarray(Image.open(p)), replicas=1, device=device) for p in (img1_path, img2_path)]


def interpolate(model, frames, pad, device):
    """Returns the fp32 prediction as a [H,W] GPU tensor.  frames: the two [1,1,H,W] planes of read_pair, or the
    reference's [1,6,H,W] input tensor (then the generic forward runs)."""
    with torch.no_grad():
        if isinstance(frames, (list, tuple)):
            f1, f2 = (F.pad(f.to(device), (pad, pad, pad, pad)) for f in frames)
            pred = model.interpolate_gray(f1, f2)
        else:
            pred = model(F.pad(frames.to(device), (pad, pad, pad, pad)))
    pred = F.pad(pred, (-pad, -pad, -pad, -pad))
    return pred[0, 0]


def read_pair_u8(img1_path, img2_path, device):
    """The two frames as ONE uint8 tensor [1,2,H,W] on the GPU: what the PNGs hold, nothing converted on the host."""
    a, b = (np.asarray(Image.open(p)) for p in (img1_path, img2_path))
    if a.dtype != np.uint8 or b.dtype != np.uint8 or a.ndim != 2 or a.shape != b.shape:
        raise TypeError("expected two 8-bit grayscale images of one size, got %s %s / %s %s" % (a.dtype, a.shape, b.dtype, b.shape))
    return torch.from_numpy(np.stack((a, b))[None]).to(device)


def interpolate_u8(model, frames_u8):
    """(pred float32 [H,W], image uint8 [H,W] numpy) from the uint8 frames: the first convolution reads the bytes, the fused apply
    stores the truncated image (IFNet.interpolate_gray_u8; reference :55-66 and :76 without the tensors in between)."""
    with torch.no_grad():
        pred, img = model.interpolate_gray_u8(frames_u8)
    return pred[0, 0], img[0].cpu().numpy()


def to_uint8(pred):
    return tensor_to_gray(pred)               # (pred*255) truncated, no clamp (reference :76), on the GPU


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('-c', '--cfg', type=str, default='ms_l1loss_decay')
    parser.add_argument('-id', '--model_id', type=str, default='interp')
    parser.add_argument('-i1', '--img1', type=str, default=None)
    parser.add_argument('-i2', '--img2', type=str, default=None)
    parser.add_argument('-o', '--output', type=str, default=None)
    parser.add_argument('--ckpt', type=str, default=None, help="default: ../trained_models/<id>/<id>.ckpt")
    parser.add_argument('--config-dir', type=str, default=None, help="default: ./config next to this script")
    args = parser.parse_args(argv)

    print('cfg_file: ' + args.cfg + '.yaml')
    cfg = load_config(args.cfg, args.config_dir)
    device = torch.device('cuda:0' if torch.cuda.is_available() else 'cpu')
    ckpt_path = args.ckpt or os.path.join('../trained_models', args.model_id, args.model_id + '.ckpt')
    model = load_model(cfg, ckpt_path, device)

    print('Inference...')
    t1 = time.time()
    pad = cfg["TEST"]["pad"]
    frames_u8 = read_pair_u8(args.img1, args.img2, device) if pad == 0 else None
    if frames_u8 is not None and HF.first_layer_u8_ok(frames_u8, model.conv32[0]):
        pred, image = interpolate_u8(model, frames_u8)        # uint8 in, uint8 out: both conversions inside the network's own launches
    else:                                                     # a padded test configuration, or a width the uint8 first layer does not take
        pred = interpolate(model, read_pair(args.img1, args.img2, device), pad, device)
        image = to_uint8(pred)
    Image.fromarray(image).save(args.output)
    print('COST TIME: ', (time.time() - t1))
    return pred.cpu().numpy()


if __name__ == "__main__":
    main()
