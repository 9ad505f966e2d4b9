"""CPU-side checks of the model API (no compute through the native ops): class names, constructor
signatures and state_dict keys are the reference's (golden key lists captured from the reference
classes by tests/golden/make_model_goldens.py)."""
import json
import os

import pytest
import torch

import networks
from model.model_fusionnet import FusionNet as SffFusionNet
from model.model_interp import IFNet as SffIFNet
from model.model_unet import UNet as SffUNet
from model.sepconv import FunctionSepconv, ModuleSepconv
from weight_recipe import fill_

BUILDERS = {
    "sff_ifnet": (lambda: SffIFNet(kernel_size=51), 21660468),
    "sp_ifnet": (lambda: networks.IFNet(), 23123172),
    "sp_unet": (lambda: networks.UNet(1, 1), 17266241),
    "sp_fusionnet": (lambda: networks.FusionNet(1, 1), 17266241),
    "sff_unet": (lambda: SffUNet(in_channel=6, out_channel=1), 1692963),
    "sff_fusionnet": (lambda: SffFusionNet(input_nc=6, output_nc=2, ngf=32), 19646690),
}


@pytest.fixture(scope="module")
def golden_keys(golden_dir):
    with open(os.path.join(golden_dir, "models_state_dict_keys.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("name", sorted(BUILDERS))
def test_state_dict_keys_and_param_counts_match_reference(name, golden_keys):
    build, nparams = BUILDERS[name]
    net = build()
    assert sorted(net.state_dict().keys()) == golden_keys[name]
    assert sum(p.numel() for p in net.parameters()) == nparams
    fill_(net, 1)                                                        # strict load of a full state dict


def test_checkpoint_layout_roundtrip(tmp_path):
    """main_ms.py:282-285 saves {'current_iter','valid_result','model_weights'} with a 'module.' prefix
    under DataParallel; inference strips 7 characters (inference_singleImage.py:42-47)."""
    net = SffUNet(6, 1)
    ckpt = {"current_iter": 3, "valid_result": None,
            "model_weights": {"module." + k: v for k, v in net.state_dict().items()}}
    path = os.path.join(tmp_path, "model-000003.ckpt")
    torch.save(ckpt, path)
    loaded = torch.load(path, map_location="cpu")
    other = SffUNet(6, 1)
    other.load_state_dict({k[7:]: v for k, v in loaded["model_weights"].items()}, strict=True)
    for a, b in zip(net.state_dict().values(), other.state_dict().values()):
        assert torch.equal(a, b)


def test_models_refuse_cpu_tensors():
    # the convolution blocks and the sepconv op are GPU-only: no silent CPU fallback anywhere
    with pytest.raises(NotImplementedError):
        SffUNet(6, 1)(torch.zeros(1, 6, 8, 8))
    with pytest.raises(NotImplementedError):
        FunctionSepconv(torch.zeros(1, 3, 52, 52), torch.zeros(1, 51, 2, 2), torch.zeros(1, 51, 2, 2))
    with pytest.raises(NotImplementedError):
        ModuleSepconv()(torch.zeros(1, 3, 52, 52), torch.zeros(1, 51, 2, 2), torch.zeros(1, 51, 2, 2))
