"""Generates tests/golden/models_*.npz from the REFERENCE model classes (this container only).

Run from the repo root:  python tests/golden/make_model_goldens.py

How: the reference's model files import packages that do not exist here (torchvision, skimage,
torch.utils.ffi through libs.sepconv).  Those imports are unused by the classes, so empty in-memory
stub modules are injected into sys.modules and the reference files are imported as they lie under
/root/reference (nothing is copied).  The dense-conv arithmetic is then the reference's own module
graph executed by this container's torch CPU kernels.  The sepconv op has no CPU implementation in the
reference (SeparableConvolution.py:47-48), so `SeparableConvolution.apply` is bound to the CPU oracle
(oracle/sepconv_oracle.c) for the whole-IFNet goldens.

Only inputs' seeds, outputs (sub-sampled where large) and state-dict key lists are stored; weights
come from tests/weight_recipe.py on both sides.
"""
import importlib
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.normpath(os.path.join(HERE, "..", ".."))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

from oracle import sepconv_c  # noqa: E402
from weight_recipe import fill_, input_for  # noqa: E402

SEED = 555


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _OracleSepconv(torch.autograd.Function):
    # The C oracle is fp32.  float64 tensors (the conditioning runs of make_step_goldens.py) are rounded to fp32 for the op
    # and its gradients and cast back: the op stays the oracle's in both precisions.
    @staticmethod
    def _np(t):
        return np.ascontiguousarray(t.detach().to(torch.float32).numpy())

    @staticmethod
    def forward(ctx, inp, ver, hor):
        out = sepconv_c.forward(_OracleSepconv._np(inp), _OracleSepconv._np(ver), _OracleSepconv._np(hor))
        ctx.save_for_backward(inp, ver, hor)
        return torch.from_numpy(out).to(inp.dtype)

    @staticmethod
    def backward(ctx, grad):       # used by make_step_goldens.py; (zeros, gradVertical, gradHorizontal) as SeparableConvolution.py:55-77
        inp, ver, hor = ctx.saved_tensors
        gi, gv, gh = sepconv_c.backward(_OracleSepconv._np(grad), _OracleSepconv._np(inp), _OracleSepconv._np(ver),
                                        _OracleSepconv._np(hor))
        return torch.from_numpy(gi).to(inp.dtype), torch.from_numpy(gv).to(ver.dtype), torch.from_numpy(gh).to(hor.dtype)


def install_stubs():
    tv = _stub("torchvision")
    tv.utils = _stub("torchvision.utils")
    tv.datasets = _stub("torchvision.datasets")
    tv.transforms = _stub("torchvision.transforms")
    sk = _stub("skimage")
    sk.morphology = _stub("skimage.morphology")
    libs = _stub("libs"); libs.__path__ = []
    sep = _stub("libs.sepconv"); sep.__path__ = []
    _stub("libs.sepconv.SeparableConvolution", SeparableConvolution=_OracleSepconv)


def load_ref(path, modname):
    spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, path))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def sub(t):
    """Sub-sample a [B,51,H,W] head for storage."""
    return t[:, ::5, ::4, ::4].contiguous().numpy()


def _trunk(net, x):
    """model_interp.py:60-83 re-run explicitly: the trunk output in front of the four kernel heads"""
    t = net.conv32(x); t = net.pool(t); x64 = net.conv64(t); x128 = net.conv128(net.pool(x64))
    x256 = net.conv256(net.pool(x128)); x512 = net.conv512(net.pool(x256)); t = net.conv512x512(net.pool(x512))
    t = net.upsamp512(t) + x512; t = net.upconv256(t); t = net.upsamp256(t) + x256; t = net.upconv128(t)
    t = net.upsamp128(t) + x128; t = net.upconv64(t); t = net.upsamp64(t) + x64
    return t


def main():
    import copy
    torch.set_num_threads(8)
    install_stubs()
    out = {}
    keys = {}

    def cond(key, o32, o64):
        """<key>_cond: how far the REFERENCE's own fp32 result is from its float64 result, relative to the largest element -- the
        conditioning of that model on that input, measured, not assumed.  tests/test_models_gpu.py takes its tolerance from it."""
        out[key + "_cond"] = np.float64((o32.double() - o64).abs().max().item() / max(o64.abs().max().item(), 1e-30))

    def in_f64(net):
        return copy.deepcopy(net).double()

    with torch.no_grad():
        # ---- SFF IFNet (sff_scripts_interp/model/model_interp.py:9-107)
        m = load_ref("sff_scripts_interp/model/model_interp.py", "ref_model_interp")
        net = m.IFNet(kernel_size=51).eval()
        keys["sff_ifnet"] = fill_(net, SEED)
        x = input_for(SEED, "sff_ifnet", (1, 6, 64, 64))
        o = net(x); out["sff_ifnet_out"] = o.numpy()
        # the trunk up to the four kernel heads (model_interp.py:60-89), re-run explicitly
        t = _trunk(net, x)
        k2h, k1v = net.upconv51_1(t), net.upconv51_4(t)
        out["sff_ifnet_k2h"] = sub(k2h); out["sff_ifnet_k1v"] = sub(k1v)
        out["sff_ifnet_trunk64"] = t[:, ::8, ::2, ::2].contiguous().numpy()
        n64 = in_f64(net); x64_ = x.double(); t64 = _trunk(n64, x64_)
        cond("sff_ifnet_out", o, n64(x64_)); cond("sff_ifnet_trunk64", t, t64)
        cond("sff_ifnet_k2h", k2h, n64.upconv51_1(t64)); cond("sff_ifnet_k1v", k1v, n64.upconv51_4(t64))

        # ---- SP IFNet / UNet / FusionNet (sp_scripts_train/networks.py)
        m = load_ref("sp_scripts_train/networks.py", "ref_networks"); sys.modules["ref_networks"] = m
        net = m.IFNet().eval()
        keys["sp_ifnet"] = fill_(net, SEED + 1)
        x = input_for(SEED, "sp_ifnet", (1, 6, 64, 64))
        o = net(x); out["sp_ifnet_out"] = o.numpy(); cond("sp_ifnet_out", o, in_f64(net)(x.double()))
        for mode in ("eval", "train"):
            net = m.UNet(1, 1)
            keys["sp_unet"] = fill_(net, SEED + 2)
            net.train(mode == "train")
            x = input_for(SEED, "sp_unet", (2, 1, 32, 32))
            o = net(x); out["sp_unet_%s" % mode] = o.numpy(); cond("sp_unet_%s" % mode, o, in_f64(net)(x.double()))
            net = m.FusionNet(1, 1)
            keys["sp_fusionnet"] = fill_(net, SEED + 3)
            net.train(mode == "train")
            a = input_for(SEED, "sp_fusion_a", (2, 1, 32, 32)); b = input_for(SEED, "sp_fusion_b", (2, 1, 32, 32))
            o = net(a, b); out["sp_fusionnet_%s" % mode] = o.numpy(); cond("sp_fusionnet_%s" % mode, o, in_f64(net)(a.double(), b.double()))
        # building blocks on their own (train-mode BN statistics included)
        blk = m.DoubleConv(3, 8, 5).train(); fill_(blk, SEED + 4)
        x = input_for(SEED, "dc", (2, 3, 12, 10))
        o = blk(x); out["sp_doubleconv_train"] = o.numpy(); cond("sp_doubleconv_train", o, in_f64(blk)(x.double()))
        blk = m.Up(16, 4, True).eval(); fill_(blk, SEED + 5)
        u1, u2 = input_for(SEED, "up1", (1, 8, 5, 6)), input_for(SEED, "up2", (1, 8, 11, 13))
        o = blk(u1, u2); out["sp_up_eval"] = o.numpy(); cond("sp_up_eval", o, in_f64(blk)(u1.double(), u2.double()))

        # ---- SFF fusion UNet / FusionNet (sff_scripts_fusion/model)
        m = load_ref("sff_scripts_fusion/model/model_unet.py", "ref_model_unet")
        for mode in ("eval", "train"):
            net = m.UNet(in_channel=6, out_channel=1)
            keys["sff_unet"] = fill_(net, SEED + 6)
            net.train(mode == "train")
            x = input_for(SEED, "sff_unet", (2, 6, 32, 32))
            o = net(x); out["sff_unet_%s" % mode] = o.numpy(); cond("sff_unet_%s" % mode, o, in_f64(net)(x.double()))
        m = load_ref("sff_scripts_fusion/model/model_fusionnet.py", "ref_model_fusionnet")
        for mode in ("eval", "train"):
            net = m.FusionNet(input_nc=6, output_nc=2, ngf=32)
            keys["sff_fusionnet"] = fill_(net, SEED + 7)
            net.train(mode == "train")
            x = input_for(SEED, "sff_fusionnet", (2, 6, 32, 32))
            o = net(x); out["sff_fusionnet_%s" % mode] = o.numpy(); cond("sff_fusionnet_%s" % mode, o, in_f64(net)(x.double()))

        # ---- SP full pipeline (test_fusion.py:105-121) on one 64x64 tile set, recipe weights
        m = sys.modules["ref_networks"]
        vfi = m.IFNet().eval(); fill_(vfi, SEED + 1)
        den = m.UNet(1, 1).eval(); fill_(den, SEED + 2)
        fus = m.FusionNet(1, 1).eval(); fill_(fus, SEED + 3)
        im = [input_for(SEED, "pipe%d" % k, (1, 1, 64, 64)) for k in range(4)]          # im1, im2_degra, im3_degra, im4
        masks = [(input_for(SEED, "mask%d" % k, (1, 1, 64, 64)) > 0.5).float() for k in range(2)]
        inputs_vfi = torch.cat((im[0], im[0], im[0], im[3], im[3], im[3]), 1)
        vfi_pred1 = torch.unsqueeze(vfi(inputs_vfi)[:, 0], 1)
        vfi_pred2 = torch.unsqueeze(vfi(inputs_vfi)[:, 1], 1)
        d1 = den(im[1]); d2 = den(im[2])
        p1 = fus(torch.mul(vfi_pred1, 1 - masks[0]), torch.mul(d1, masks[0])); p2 = fus(torch.mul(vfi_pred2, 1 - masks[1]), torch.mul(d2, masks[1]))
        out["sp_pipeline_pred1"] = p1.numpy(); out["sp_pipeline_pred2"] = p2.numpy()
        vfi6, den6, fus6 = in_f64(vfi), in_f64(den), in_f64(fus)
        v6 = vfi6(inputs_vfi.double()); m6 = [k.double() for k in masks]
        cond("sp_pipeline_pred1", p1, fus6(v6[:, 0:1] * (1 - m6[0]), den6(im[1].double()) * m6[0]))
        cond("sp_pipeline_pred2", p2, fus6(v6[:, 1:2] * (1 - m6[1]), den6(im[2].double()) * m6[1]))

        # ---- CLI golden (a12): two 256x256 8-bit frames -> fp32 pred and uint8 output image.
        # Weights: the recipe, then the last conv of every kernel head is damped so that V ~ 1/51 and
        # H ~ 0.5/51 (+ a small trunk-dependent part): pred stays inside [0,1] and the uint8 truncation
        # is meaningful (see tests/weight_recipe.cli_weights_).
        from weight_recipe import cli_weights_, cli_frames
        m = load_ref("sff_scripts_interp/model/model_interp.py", "ref_model_interp_cli")
        net = m.IFNet(kernel_size=51).eval()
        cli_weights_(net, SEED + 8)
        f1, f2 = cli_frames(256, 256)
        frames = np.concatenate([np.repeat(f1[None], 3, 0), np.repeat(f2[None], 3, 0)], 0)[None]
        xin = torch.from_numpy(frames.astype(np.float32) / 255.0)
        o = net(xin)
        cond("cli_pred", o, in_f64(net)(xin.double()))
        pred = np.squeeze(o.numpy())
        out["cli_pred"] = pred[::4, ::4].copy()
        out["cli_uint8"] = (pred * 255).astype(np.uint8)

    np.savez_compressed(os.path.join(HERE, "models.npz"), **out)
    with open(os.path.join(HERE, "models_state_dict_keys.json"), "w") as f:
        json.dump(keys, f, indent=0)
    for k, v in out.items():
        if k.endswith("_cond"):
            print("%-28s fp32 vs fp64 of the reference: %.3e of the largest element" % (k, float(v)))
        else:
            print("%-28s %-18s absmax %.4g" % (k, v.shape, np.abs(v).max()))
    print("models.npz", os.path.getsize(os.path.join(HERE, "models.npz")), "bytes")


if __name__ == "__main__":
    main()
