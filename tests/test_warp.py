"""Bilinear back-warp: the numpy oracle against goldens produced by the reference module
itself (CPU), and the HIP kernel against both (GPU)."""
import os

import numpy as np
import pytest
import torch

from oracle import warp_numpy

CASES = ("small", "edge", "c1")


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "warp.npz"))


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_golden(gold, name):
    got = warp_numpy.warp(gold[name + "_img"], gold[name + "_flow"])
    assert np.abs(got - gold[name + "_out"]).max() <= 1e-6


def test_oracle_identity_and_integer_shift():
    rng = np.random.default_rng(0)
    img = rng.random((1, 2, 6, 7), dtype=np.float32)
    zero = np.zeros((1, 2, 6, 7), np.float32)
    assert np.array_equal(warp_numpy.warp(img, zero), img)
    flow = zero.copy(); flow[:, 0] = 2; flow[:, 1] = -1          # out[y,x] = img[y-1, x+2], zero outside
    want = np.zeros_like(img); want[:, :, 1:, :-2] = img[:, :, :-1, 2:]
    assert np.array_equal(warp_numpy.warp(img, flow), want)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_kernel_matches_reference_golden(gold, name):
    from utils.image_warp_torch import SpatialTransformation
    warp = SpatialTransformation(use_gpu=True)
    img = torch.from_numpy(gold[name + "_img"]).cuda()
    flow = torch.from_numpy(gold[name + "_flow"]).cuda().permute(0, 2, 3, 1)     # [B,H,W,2] as the reference passes it
    got = warp(img, flow).cpu().numpy()
    assert np.abs(got - gold[name + "_out"]).max() <= 1e-6


@pytest.mark.gpu
def test_kernel_matches_oracle_large_and_exact_cases():
    from utils.image_warp_torch import SpatialTransformation
    warp = SpatialTransformation(use_gpu=True)
    rng = np.random.default_rng(1)
    img = rng.random((2, 3, 70, 130), dtype=np.float32)
    flow = (rng.standard_normal((2, 2, 70, 130)) * 4).astype(np.float32)
    got = warp(torch.from_numpy(img).cuda(), torch.from_numpy(flow).cuda().permute(0, 2, 3, 1)).cpu().numpy()
    assert np.abs(got - warp_numpy.warp(img, flow)).max() <= 1e-6
    # integer displacements are pure index work: bit-exact
    flow = np.round(flow)
    got = warp(torch.from_numpy(img).cuda(), torch.from_numpy(flow).cuda().permute(0, 2, 3, 1)).cpu().numpy()
    assert np.array_equal(got, warp_numpy.warp(img, flow))
    with pytest.raises(NotImplementedError):
        warp(torch.zeros(1, 1, 2, 2), torch.zeros(1, 2, 2, 2))
