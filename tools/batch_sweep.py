#!/usr/bin/env python
"""Developer probe: the SFF restoration forward / IFNet forward on 8 tiles of 1024^2 as ONE batch of 8 or as sub-batches of 1 / 2 / 4 in turn
(activations of a sub-batch may stay in the 256 MB last-level cache between producer and consumer; small grids fill the chip worse)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "sstem-restoration_amd"))
import torch
import steps
dev = torch.device("cuda")
what = sys.argv[1] if len(sys.argv) > 1 else "sff"
fw = steps.SFFRestoreForward(dev, batch=8, size=1024) if what == "sff" else steps.IFNetForward(dev, batch=8, size=1024)
def run(sub):
    outs = []
    for i in range(0, 8, sub):
        if what == "sff":
            outs.append(fw._restore(fw.models, fw.prev[i:i + sub], fw.nxt[i:i + sub], fw.sff[i:i + sub])[0])
        else:
            outs.append(fw.net.interpolate_gray(fw.f1[i:i + sub], fw.f2[i:i + sub]))
    return outs
with torch.no_grad():
    t0 = time.time()
    while time.time() - t0 < 0.8:
        run(8); torch.cuda.synchronize()
    ref = torch.cat(run(8))
    for sub in (8, 4, 2, 1, 8):
        run(sub); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            o = run(sub)
        e1.record(); torch.cuda.synchronize()
        same = torch.equal(torch.cat(o), ref)
        print("%s: sub-batch %d: %.2f ms per 8 tiles   (bit-identical to one batch of 8: %s)" % (what, sub, e0.elapsed_time(e1) / 5, same), flush=True)
