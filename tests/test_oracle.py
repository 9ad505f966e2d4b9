"""CPU tests of the oracle itself (no GPU): the C restatement against analytic known answers,
against the independent float64 restatement, and against the committed golden fixtures."""
import os

import numpy as np
import pytest

from oracle import sepconv_c, sepconv_numpy
from sepconv_cases import make_case, onehot_expected


@pytest.mark.parametrize("shape", [(2, 3, 12, 20), (1, 1, 5, 7), (1, 3, 1, 1), (3, 2, 9, 4)])
def test_c_matches_independent_restatement(shape):
    inp, ver, hor, grad = make_case(1, *shape)
    out = sepconv_c.forward(inp, ver, hor)
    ref = sepconv_numpy.forward(inp, ver, hor)
    scale = np.abs(ref).max() + 1e-6
    assert np.abs(out - ref).max() / scale < 2e-5
    if shape[1] == 3:
        gi, gv, gh = sepconv_c.backward(grad, inp, ver, hor)
        _, gv2, gh2 = sepconv_numpy.backward(grad, inp, ver, hor)
        assert not gi.any()  # kernel.cu:152-206 never writes gradInput
        assert np.abs(gv - gv2).max() / (np.abs(gv2).max() + 1e-6) < 2e-5
        assert np.abs(gh - gh2).max() / (np.abs(gh2).max() + 1e-6) < 2e-5


def test_onehot_is_exact_gather():
    inp, ver, hor, _ = make_case(2, 2, 3, 10, 13, kind="onehot")
    out = sepconv_c.forward(inp, ver, hor)
    assert np.array_equal(out, onehot_expected(inp, ver, hor))


def test_box_filter():
    inp, ver, hor, _ = make_case(3, 1, 3, 6, 8, kind="box")
    out = sepconv_c.forward(inp, ver, hor)
    win = np.lib.stride_tricks.sliding_window_view(inp.astype(np.float64), (51, 51), axis=(2, 3))
    ref = win.mean(axis=(-1, -2))
    assert np.abs(out - ref).max() < 1e-5


def test_backward_rejects_non_three_channels():
    inp, ver, hor, grad = make_case(4, 1, 2, 3, 3)
    with pytest.raises(RuntimeError):
        sepconv_c.backward(grad, inp, ver, hor)


def test_gradcheck_shape_of_reference():
    """The reference's only known-answer-style check (model_interp.py:109-119): input (2,3,51,51),
    V,H (2,51,1,1), eps=1e-2, atol=rtol=1e-2 -- here by central differences on the oracle."""
    rng = np.random.default_rng(5)
    inp = rng.standard_normal((2, 3, 51, 51)).astype(np.float32)
    ver = rng.standard_normal((2, 51, 1, 1)).astype(np.float32)
    hor = rng.standard_normal((2, 51, 1, 1)).astype(np.float32)
    g = np.ones((2, 3, 1, 1), np.float32)
    _, gv, gh = sepconv_c.backward(g, inp, ver, hor)
    eps = 1e-2
    for arr, grad in ((ver, gv), (hor, gh)):
        for b in range(2):
            for f in (0, 17, 50):
                old = arr[b, f, 0, 0]
                arr[b, f, 0, 0] = old + eps
                up = sepconv_numpy.forward(inp, ver, hor).sum()
                arr[b, f, 0, 0] = old - eps
                dn = sepconv_numpy.forward(inp, ver, hor).sum()
                arr[b, f, 0, 0] = old
                num = (up - dn) / (2 * eps)
                assert abs(num - grad[b, f, 0, 0]) <= 1e-2 + 1e-2 * abs(num)


def test_openmp_build_is_bitwise_identical():
    inp, ver, hor, grad = make_case(6, 2, 3, 8, 9)
    assert np.array_equal(sepconv_c.forward(inp, ver, hor), sepconv_c.forward(inp, ver, hor, omp=True))
    a = sepconv_c.backward(grad, inp, ver, hor)
    b = sepconv_c.backward(grad, inp, ver, hor, omp=True)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))


def test_golden_fixture(golden_dir):
    path = os.path.join(golden_dir, "sepconv_kat.npz")
    z = np.load(path)
    out = sepconv_c.forward(z["input"], z["vertical"], z["horizontal"])
    assert np.array_equal(out, z["output"])
    _, gv, gh = sepconv_c.backward(z["grad_output"], z["input"], z["vertical"], z["horizontal"])
    assert np.array_equal(gv, z["grad_vertical"])
    assert np.array_equal(gh, z["grad_horizontal"])
    # and the float64 restatement agrees with the stored values
    ref = sepconv_numpy.forward(z["input"], z["vertical"], z["horizontal"])
    assert np.abs(z["output"] - ref).max() / np.abs(ref).max() < 2e-5


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference is only present in the build container")
def test_committed_goldens_are_reproduced_by_their_generators(repo_root):
    """tests/golden/regenerate_and_diff.py: every generator re-run against the reference reproduces its committed fixture
    (a drifted generator or a hand-edited fixture would otherwise go unnoticed)."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(repo_root, "tests", "golden", "regenerate_and_diff.py")], cwd=repo_root,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]


def test_cpu_twin_of_the_metric_path_reproduces_the_reference_chain(golden_dir):
    """tests/cpu_twin.py (bench.py's CPU baseline of "interp + fusion fwd": the product classes' module trees run as the stock
    nn.Sequential the reference builds, the oracle sepconv, the numpy warp) against the chain composed from the REFERENCE classes
    (tests/golden/sff_chain.npz, make_sff_chain_golden.py): the same torch CPU kernels on the same weights -- 1e-6 of range."""
    import torch
    import cpu_twin
    import sff_pipeline
    from oracle import warp_numpy
    from weight_recipe import cli_weights_, fill_, sff_chain_inputs, sff_flow_weights_
    gold = np.load(os.path.join(golden_dir, "sff_chain.npz"))
    models = sff_pipeline.build_models("cpu")
    cli_weights_(models["interp"], 555 + 8); sff_flow_weights_(models["flow"], 555 + 7); fill_(models["fusion"], 555 + 6)
    prev, nxt, sff = (torch.from_numpy(a) for a in sff_chain_inputs(2, 64, 64))
    pred, interp, flow, warped = cpu_twin.restore_sff(models, prev, nxt, sff, sepconv_c.forward, warp_numpy.warp)
    for got, key in ((interp, "interp_f"), (flow, "flow_f"), (warped, "warped_f"), (pred, "pred_f")):
        ref = np.asarray(gold[key], np.float64)
        err = np.abs(got.double().numpy() - ref).max() / np.abs(ref).max()
        assert err <= 1e-6, "%s: %.3e of range" % (key, err)
