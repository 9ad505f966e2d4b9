#!/usr/bin/env python
"""Evidence for tests/test_steps_gpu.py's COND_FACTOR (round-2 verdict, item 8): how far do ARITHMETIC-EQUIVALENT configurations of this
library scatter on one training step, compared with the ONE fp32-vs-fp64 sample of the reference that the step golden stores?
Round 2 measured it on the SP UNet step only; this runs the SFF fusion U-Net step (tests/golden/steps.npz, tag sff_fusion) and the SP
UNet step under several configurations -- fp32 MFMA / X6 / AUTO convolutions, BatchNorm statistics from the conv store or from their own
pass, split over K on or off -- each in a process of its own (the knobs are read once), and prints the spread of the first convolution's
gradient (elementwise, of its largest element) and of the gradient norms (worst relative), next to the stored conditioning.

    python tools/probe_step_noise.py            # parent: spawns the configurations, prints the table
"""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "sstem-restoration_amd"), os.path.join(REPO, "tests"), REPO):
    if p not in sys.path:
        sys.path.insert(0, p)

CONFIGS = [
    ("auto", {}),
    ("fp32 mfma", {"SSTEM_CONV_ALGO": "mfma"}),
    ("x6 forced", {"SSTEM_CONV_ALGO": "bf16x6"}),
    ("auto, BN statistics from the conv store", {"SSTEM_BN_FUSED_STATS": "1"}),
    ("auto, no split over K", {"SSTEM_CONV_KSPLIT": "0"}),
    ("fp32 mfma, no split over K, BN statistics from the conv store", {"SSTEM_CONV_ALGO": "mfma", "SSTEM_CONV_KSPLIT": "0", "SSTEM_BN_FUSED_STATS": "1"}),
    ("auto, no mask fusion, no side stream", {"SSTEM_MASK_FUSION": "0", "SSTEM_SIDE_WGRAD": "0"}),
]


def child(tag):
    import numpy as np
    import torch
    import torch.nn.functional as F
    from weight_recipe import fill_, input_for
    SEED = 555
    z = np.load(os.path.join(REPO, "tests", "golden", "steps.npz"))
    if tag == "sff_fusion":
        from model.model_unet import UNet
        net = UNet(6, 1).train(); fill_(net, SEED + 6); net.cuda()
        inp = input_for(SEED, "step_in", (2, 6, 64, 64)).cuda(); target = input_for(SEED, "step_tg", (2, 1, 64, 64)).cuda()
        x = inp.clone(); x[:, :3] = torch.from_numpy(z["sff_fusion_warped"]).cuda()
        loss = F.l1_loss(net(x), target)
    else:
        import networks
        net = networks.UNet(1, 1).train(); fill_(net, SEED + 2); net.cuda()
        x = input_for(SEED, "spu_in", (2, 1, 64, 64)).cuda(); target = input_for(SEED, "spu_tg", (2, 1, 64, 64)).cuda()
        loss = F.l1_loss(net(x), target)
    loss.backward()
    params = list(net.named_parameters())
    first = params[0][1].grad.detach().cpu().double().numpy()
    norms = [float(p.grad.double().norm()) if p.grad is not None else -1.0 for _, p in params]
    print("PROBE " + json.dumps({"loss": loss.item(), "first": first.ravel().tolist(), "norms": norms}))


def main():
    import numpy as np
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        return child(sys.argv[2])
    z = np.load(os.path.join(REPO, "tests", "golden", "steps.npz"))
    for tag in ("sff_fusion", "sp_unet"):
        res = []
        for name, env in CONFIGS:
            e = dict(os.environ); e.update(env)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", tag], env=e, capture_output=True, text=True)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("PROBE ")]
            if r.returncode != 0 or not line:
                print("%s / %s failed: %s" % (tag, name, r.stderr[-300:]))
                continue
            res.append((name, json.loads(line[-1][6:])))
        norms_ref = z[tag + "_grad_norms"]
        live = norms_ref > 1e-5 * norms_ref.max()
        step_cond = float(z[tag + "_norm_cond"][live].max())
        first_cond = float(z[tag + "_grad0_cond"])
        ref_first = z[tag + "_grad0"].astype(np.float64).ravel()
        print("\n%s step: stored conditioning (reference fp32 vs fp64, ONE sample): gradient norms %.2e, first convolution's gradient %.2e" % (tag, step_cond, first_cond))
        print("  %-66s %-12s %-14s %-14s" % ("configuration", "loss", "first grad vs", "worst norm vs"))
        print("  %-66s %-12s %-14s %-14s" % ("", "", "the golden", "the golden"))
        firsts, normss = [], []
        for name, d in res:
            f = np.asarray(d["first"]); n = np.asarray(d["norms"])
            ef = np.abs(f - ref_first).max() / np.abs(ref_first).max()
            en = (np.abs(n - norms_ref)[live] / norms_ref[live]).max()
            firsts.append(f); normss.append(n)
            print("  %-66s %-12.8g %-14.2e %-14.2e" % (name, d["loss"], ef, en))
        sf = max(np.abs(a - b).max() for a in firsts for b in firsts) / np.abs(ref_first).max()
        sn = max((np.abs(a - b)[live] / norms_ref[live]).max() for a in normss for b in normss)
        print("  spread between configurations: first gradient %.2e (%.1f x the stored sample), gradient norms %.2e (%.1f x)" % (sf, sf / first_cond, sn, sn / step_cond))


if __name__ == "__main__":
    main()
