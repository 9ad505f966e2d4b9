"""GPU parity of the headline metric's literal path -- "interp + fusion fwd" (sff_pipeline.restore_sff: IFNet -> unfolding-flow
FusionNet -> back-warp -> fusion UNet) -- against the chain composed from the REFERENCE classes the way the reference's two SFF
inference scripts compose them (sff_scripts_interp/inference_singleImage.py:55-76 + sff_scripts_fusion/inference.py:126-153;
tests/golden/make_sff_chain_golden.py), in both spellings: the chain kept in fp32 and with the uint8 PNG between the two scripts."""
import json
import os

import numpy as np
import pytest
import torch

import sff_pipeline
from weight_recipe import cli_weights_, fill_, sff_chain_inputs, sff_flow_weights_

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("conv_algo_matrix")]
SEED = 555
NORTH_STAR_REL = 1e-4          # fp32 restored pixels within 1e-4 (of the output range)
COND_FACTOR = 8.0              # ... or 8x the reference's own fp32-vs-fp64 deviation where the chain is worse conditioned (test_steps_gpu.py)
measured = {}


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "sff_chain.npz"))


@pytest.fixture()
def models():
    m = sff_pipeline.build_models("cuda")
    cli_weights_(m["interp"], SEED + 8); sff_flow_weights_(m["flow"], SEED + 7); fill_(m["fusion"], SEED + 6)
    for net in m.values():
        net.cuda().eval()
    return m


def _inputs():
    return tuple(torch.from_numpy(a).cuda() for a in sff_chain_inputs(2, 64, 64))


def _close(got, gold, key):
    a = got.detach().cpu().double().numpy(); ref = np.asarray(gold[key], np.float64)
    assert a.shape == ref.shape
    tol = max(NORTH_STAR_REL, COND_FACTOR * float(gold[key + "_cond"]))
    scale = np.abs(ref).max()
    err = np.abs(a - ref).max() / scale
    measured[key] = max(measured.get(key, 0.0), err)
    assert err <= tol, "%s: %.3e of the largest element (allowed %.2e)" % (key, err, tol)
    return err


def test_restore_sff_matches_the_reference_chain(gold, models):
    prev, nxt, sff = _inputs()
    pred, interp, flow, warped = sff_pipeline.restore_sff(models, prev, nxt, sff)
    _close(interp, gold, "interp_f")
    _close(flow, gold, "flow_f")
    _close(warped, gold, "warped_f")
    _close(pred, gold, "pred_f")
    # the metric's "PSNR vs ref": [0,1] pixels, so the mean squared deviation IS the PSNR
    mse = float(((pred.cpu().double().numpy() - gold["pred_f"]) ** 2).mean())
    psnr = 10 * np.log10(1.0 / max(mse, 1e-30))
    measured["pred_f_psnr_db"] = min(measured.get("pred_f_psnr_db", 1e9), psnr)
    assert psnr >= 100.0          # a restored image 100 dB from the reference's moves no PSNR-vs-target figure by 0.01 dB


def test_restore_sff_with_the_png_between_the_two_scripts(gold, models):
    """quantise_interp=True reproduces (pred*255).astype(uint8) -> PNG -> /255.  The truncation is a step function: a pixel whose fp32
    value lies within north_star's tolerance of an integer may land on the other side.  Checked: (1) every uint8 that differs from the
    reference's differs by one step and sits on such a boundary pixel; (2) stage 2 on the REFERENCE's PNG reproduces the reference's
    flow / warped / pred; (3) with no pixel flipped (the case on every box so far) the whole chain does."""
    prev, nxt, sff = _inputs()
    pred, interp, flow, warped = sff_pipeline.restore_sff(models, prev, nxt, sff, quantise_interp=True)
    u8 = torch.round(interp[:, 0] * 255).cpu().numpy().astype(np.int32)           # interp is k/255 exactly representable to < 1e-7
    want = gold["interp_u8"].astype(np.int32)
    differs = u8 != want
    measured["interp_u8_flipped_pixels"] = max(measured.get("interp_u8_flipped_pixels", 0), int(differs.sum()))
    assert np.abs(u8 - want).max() <= 1
    assert (gold["interp_u8_margin"][differs] <= 255 * NORTH_STAR_REL).all()
    ref_png = torch.from_numpy(gold["interp_u8"].astype(np.float32) / 255.0)[:, None].cuda()
    p2, f2, w2 = sff_pipeline.fuse(models, sff, ref_png)
    _close(f2, gold, "flow_q"); _close(w2, gold, "warped_q"); _close(p2, gold, "pred_q")
    if not differs.any():
        assert torch.equal(interp, ref_png)
        assert torch.equal(pred, p2) and torch.equal(flow, f2) and torch.equal(warped, w2)


def test_restore_sharded_owns_tiles_round_robin(models):
    prev, nxt, sff = _inputs()
    tiles = [(prev, nxt, sff), (nxt, prev, sff), lambda: (prev, nxt, sff), (sff, nxt, prev), (prev, prev, sff)]
    direct = [sff_pipeline.restore_sff(models, *(t() if callable(t) else t))[0] for t in tiles]
    seen = {}
    for rank in range(2):
        own = sff_pipeline.restore_sharded(models, tiles, rank=rank, world=2)
        assert sorted(own) == list(range(rank, len(tiles), 2))
        seen.update(own)
    assert sorted(seen) == list(range(len(tiles)))
    for i, d in enumerate(direct):
        assert torch.equal(seen[i], d)
    assert torch.equal(direct[0], direct[2]) and not torch.equal(direct[0], direct[1])


def test_zz_report_measured_deviations(gold, conv_algo_matrix, repo_root):
    """Writes what the tests above measured to gpurun_out/ (kept under profiles/ per round)."""
    assert measured
    rows = {k: {"measured": v, "reference_fp32_vs_fp64": float(gold[k + "_cond"]),
                "allowed": max(NORTH_STAR_REL, COND_FACTOR * float(gold[k + "_cond"]))} if k + "_cond" in gold.files else v
            for k, v in sorted(measured.items())}
    out_dir = os.path.join(repo_root, "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "sff_chain_golden_deviations_%s.json" % conv_algo_matrix), "w") as f:
            json.dump(rows, f, indent=1)
