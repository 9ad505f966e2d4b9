"""Generates tests/golden/sepconv_kat.npz (run from the repo root: python tests/golden/make_sepconv_golden.py).

The reference cannot produce these values (its op is CUDA-only and does not build here), so
the vectors come from the C restatement oracle/sepconv_oracle.c and are cross-checked against
the independent float64 restatement before being written.  Small on purpose (tens of KB).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, ".."))

from oracle import sepconv_c, sepconv_numpy  # noqa: E402
from sepconv_cases import make_case  # noqa: E402

inp, ver, hor, grad = make_case(555, 2, 3, 6, 10, kind="softmax")  # seed 555: ms_l1loss_decay.yaml:33
ver = (ver * 51).astype(np.float32)  # O(1) coefficients
hor = (hor * 51).astype(np.float32)
out = sepconv_c.forward(inp, ver, hor)
_, gv, gh = sepconv_c.backward(grad, inp, ver, hor)
ref = sepconv_numpy.forward(inp, ver, hor)
_, gv2, gh2 = sepconv_numpy.backward(grad, inp, ver, hor)
for a, b in ((out, ref), (gv, gv2), (gh, gh2)):
    assert np.abs(a - b).max() / np.abs(b).max() < 2e-5
np.savez_compressed(os.path.join(HERE, "sepconv_kat.npz"), input=inp, vertical=ver, horizontal=hor,
                    grad_output=grad, output=out, grad_vertical=gv, grad_horizontal=gh)
print("wrote sepconv_kat.npz", os.path.getsize(os.path.join(HERE, "sepconv_kat.npz")), "bytes")
