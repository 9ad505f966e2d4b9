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


def test_conv_transpose_subpixel_weights_reproduce_the_transposed_convolution():
    """hipnn.functional.convT_subpixel_weight (host side of SSTEM_LAYOUT_CONVT_PARITY, include/sstem_conv.h): a 3 x 3 convolution with
    the [4C, Cin, 3, 3] weights it builds, followed by the pixel shuffle the kernel's store performs (channel (2 py + px) C + co ->
    output pixel (2y + py, 2x + px)), IS nn.ConvTranspose2d(k3, s2, p1, output_padding 1) -- checked in float64 on the CPU, where no
    kernel is involved; taps with ky = 0 or kx = 0 are zero (the kernel never reads them)."""
    import torch
    import torch.nn.functional as F
    import hipnn.functional as HF
    torch.manual_seed(0)
    for (N, Cin, C, H, W) in ((2, 5, 4, 6, 7), (1, 16, 32, 3, 9)):
        x = torch.randn(N, Cin, H, W, dtype=torch.float64)
        w = torch.randn(Cin, C, 3, 3, dtype=torch.float64)
        b = torch.randn(C, dtype=torch.float64)
        ref = F.conv_transpose2d(x, w, b, stride=2, padding=1, output_padding=1)
        w2 = HF.convT_subpixel_weight(w)
        assert tuple(w2.shape) == (4 * C, Cin, 3, 3)
        assert float(w2[:, :, 0, :].abs().max()) == 0.0 and float(w2[:, :, :, 0].abs().max()) == 0.0
        y = F.conv2d(x, w2, b.repeat(4), padding=1)
        out = y.view(N, 2, 2, C, H, W).permute(0, 3, 4, 1, 5, 2).reshape(N, C, 2 * H, 2 * W)
        assert (out - ref).abs().max().item() <= 1e-12


def test_algorithm_choice_of_the_networks_layers_host_logic():
    """ALGO_AUTO's choice per 3x3 layer is host logic over the layer's sizes (hipnn.functional; the C-ABI answers the range queries without
    a GPU): the layers the SFF / SP networks run at the benchmark sizes land where DESIGN 4e says they do."""
    import hipnn.functional as HF
    # launches nothing is recorded for: the fp16 two-piece id wherever X6 would run, also for the full-resolution layers with < 16 input channels
    assert HF._inference_algo(8, 64, 512, 512, 64) == HF.ALGO_MFMA_F16X3
    assert HF._inference_algo(8, 6, 1024, 1024, 32) == HF.ALGO_MFMA_F16X3
    assert HF._inference_algo(1, 1, 2048, 2048, 64) == HF.ALGO_MFMA_F16X3
    # recorded launches: X6 from 128 tiles of 8 x 32 x 64 channels (or 128 input channels), the fp32 MFMA kernel below and for < 16 input channels
    assert HF._auto_algo(16, 64, 128, 128, 64) == HF.ALGO_MFMA_BF16X6
    assert HF._auto_algo(2, 64, 128, 128, 64) == HF.ALGO_MFMA_BF16X6            # 128 tiles
    assert HF._auto_algo(2, 64, 64, 64, 64) == HF.ALGO_MFMA                     # 32 tiles, 64 input channels
    assert HF._auto_algo(2, 128, 64, 64, 128) == HF.ALGO_MFMA_BF16X6            # 128 input channels: a K loop long enough to split
    assert HF._auto_algo(16, 6, 256, 256, 32) == HF.ALGO_MFMA
    # a handful of output channels at full resolution: the streaming fp32 kernel (inference, plain NCHW store)
    assert HF._stream_small_ok(8, 32, 1024, 1024, 2) and HF._stream_small_ok(8, 32, 1024, 1024, 1) and HF._stream_small_ok(8, 6, 1024, 1024, 6)
    assert not HF._stream_small_ok(8, 32, 1024, 1024, 32)                       # 32 output channels: the matrix kernels
    assert not HF._stream_small_ok(2, 32, 256, 256, 2)                          # too small to fill the chip
    assert not HF._stream_small_ok(8, 32, 1024, 1022, 2)                        # W % 4 != 0
    assert not HF._stream_small_ok(8, 64, 1024, 1024, 8)                        # 64 x 8 products per tap and pixel: the matrix kernels win
