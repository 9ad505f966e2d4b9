"""Generates tests/golden/warp.npz from the REFERENCE module (this container only):
``sff_scripts_fusion/utils/image_warp_torch.py`` is pure torch and imports as it lies under /root/reference.
Run from the repo root:  python tests/golden/make_warp_golden.py"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
spec = importlib.util.spec_from_file_location("ref_warp", "/root/reference/sff_scripts_fusion/utils/image_warp_torch.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

from oracle import warp_numpy  # noqa: E402

rng = np.random.default_rng(555)
out = {}
for name, (B, C, H, W, amp) in {"small": (2, 3, 9, 13, 3.0), "edge": (1, 3, 8, 8, 20.0), "c1": (1, 1, 5, 7, 1.5)}.items():
    img = rng.random((B, C, H, W), dtype=np.float32)
    flow = (rng.standard_normal((B, 2, H, W)) * amp).astype(np.float32)
    if name == "edge":           # integer and half-integer displacements, far out-of-range vectors
        flow[0, :, :2] = np.round(flow[0, :, :2]); flow[0, :, 2:4] = np.round(flow[0, :, 2:4]) + 0.5
        flow[0, 0, 7] = 1e6; flow[0, 1, 6] = -1e6
    warp = ref.SpatialTransformation(use_gpu=False)
    want = warp(torch.from_numpy(img), torch.from_numpy(flow).permute(0, 2, 3, 1)).numpy()
    got = warp_numpy.warp(img, flow)
    assert np.abs(got - want).max() <= 1e-6, (name, np.abs(got - want).max())
    out[name + "_img"] = img; out[name + "_flow"] = flow; out[name + "_out"] = want
np.savez_compressed(os.path.join(HERE, "warp.npz"), **out)
print("wrote warp.npz", os.path.getsize(os.path.join(HERE, "warp.npz")), "bytes")
