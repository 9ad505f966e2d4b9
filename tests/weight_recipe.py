"""Deterministic, torch-RNG-independent weights for the golden tests.

The same recipe fills the stub-imported REFERENCE classes when the goldens are generated (this
container only) and the build's own classes when the GPU tests run, so only inputs and outputs are
committed -- never a state dict.  Every tensor is a pure function of (seed, parameter name, shape).
"""
import hashlib

import numpy as np
import torch


def _rng(seed, name):
    h = hashlib.sha256(("%d:%s" % (seed, name)).encode()).digest()
    return np.random.default_rng(int.from_bytes(h[:8], "little"))


def tensor_for(seed, name, shape):
    rng = _rng(seed, name)
    shape = tuple(shape)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=torch.long)
    if len(shape) == 4:                                  # conv / conv-transpose weight: He-uniform
        fan_in = shape[1] * shape[2] * shape[3]
        a = (6.0 / fan_in) ** 0.5
        return torch.from_numpy(rng.uniform(-a, a, shape).astype(np.float32))
    if leaf == "running_var":
        return torch.from_numpy(rng.uniform(0.5, 1.5, shape).astype(np.float32))
    if leaf == "running_mean":
        return torch.from_numpy(rng.uniform(-0.2, 0.2, shape).astype(np.float32))
    if leaf == "weight":                                 # BatchNorm gamma
        return torch.from_numpy(rng.uniform(0.8, 1.2, shape).astype(np.float32))
    return torch.from_numpy(rng.uniform(-0.05, 0.05, shape).astype(np.float32))   # biases / beta


def fill_(module, seed):
    """Overwrite every parameter and buffer of `module` in place; returns the sorted key list."""
    sd = module.state_dict()
    new = {k: tensor_for(seed, k, v.shape).to(v.dtype) for k, v in sd.items()}
    module.load_state_dict(new, strict=True)
    return sorted(sd.keys())


def input_for(seed, name, shape):
    return torch.from_numpy(_rng(seed, "input:" + name).random(tuple(shape), dtype=np.float32))


def cli_weights_(net, seed):
    """IFNet weights for the CLI golden: the standard recipe, then every kernel head's last conv
    (``upconv51_k.7``) gets weights x 2e-4 and a constant bias: 1/51 for the vertical heads (_2, _4),
    0.5/51 for the horizontal heads (_1, _3).  The interpolated frame is then ~ the mean of the two box-
    filtered inputs plus a small input-dependent part, i.e. inside [0,1]."""
    fill_(net, seed)
    with torch.no_grad():
        for k, bias in ((1, 0.5 / 51), (2, 1.0 / 51), (3, 0.5 / 51), (4, 1.0 / 51)):
            conv = getattr(net, "upconv51_%d" % k)[7]
            conv.weight.mul_(2e-4)
            conv.bias.fill_(bias)


def cli_frames(h, w):
    """Two synthetic 8-bit grayscale frames (smooth gradients + noise; seeds 555/556 as SURVEY 8d)."""
    out = []
    for seed in (555, 556):
        rng = np.random.default_rng(seed)
        yy, xx = np.mgrid[0:h, 0:w]
        img = 96 + 60 * np.sin(xx / 17.0 + seed) * np.cos(yy / 23.0) + rng.integers(-20, 21, (h, w))
        out.append(np.clip(img, 0, 255).astype(np.uint8))
    return out


def sff_flow_weights_(net, seed):
    """SFF flow FusionNet weights for the chain golden (tests/golden/make_sff_chain_golden.py): the standard recipe with the last
    convolution (``out``) x 0.25 -- displacements of a few pixels, as an unfolding flow has, instead of tens."""
    fill_(net, seed)
    with torch.no_grad():
        net.out.weight.mul_(0.25)
        net.out.bias.mul_(0.25)


def sff_chain_inputs(b, h, w):
    """prev / next / folded section of `b` synthetic tiles as float32 [b,1,h,w] = uint8 images / 255 (what the reference reads
    from its PNGs): smooth structure + noise, the folded section a darker, shifted copy of the mean of its neighbours."""
    yy, xx = np.mgrid[0:h, 0:w]
    prev, nxt, sff = [], [], []
    for k in range(b):
        rng = np.random.default_rng(557 + k)
        base = 110 + 55 * np.sin(xx / 9.0 + k) * np.cos(yy / 13.0 - k)
        p = np.clip(base + rng.integers(-25, 26, (h, w)), 0, 255).astype(np.uint8)
        n = np.clip(np.roll(base, 2, 1) + rng.integers(-25, 26, (h, w)), 0, 255).astype(np.uint8)
        s = np.clip(0.8 * np.roll(base, (1, -3), (0, 1)) + rng.integers(-25, 26, (h, w)), 0, 255).astype(np.uint8)
        prev.append(p); nxt.append(n); sff.append(s)
    return tuple((np.stack(a)[:, None].astype(np.float32) / 255.0) for a in (prev, nxt, sff))
