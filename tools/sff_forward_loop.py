#!/usr/bin/env python
"""Developer aid: the metric's literal path (steps.SFFRestoreForward: IFNet -> flow FusionNet -> back-warp -> fusion UNet, B tiles of
S x S) a few times under the default (AUTO) algorithm ids, for `rocprofv3 --kernel-trace --output-format csv`; one forward is what
lies between two launches of the fused apply: `python tools/step_timeline.py <trace.csv> sepconv_gray_mfma`.
Usage: python tools/sff_forward_loop.py [--batch 8] [--size 1024] [--steps 4]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sstem-restoration_amd"))
import torch                                   # noqa: E402
import steps                                   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--size", type=int, default=1024)
ap.add_argument("--steps", type=int, default=4)
a = ap.parse_args()
job = steps.SFFRestoreForward(torch.device("cuda"), batch=a.batch, size=a.size)
for _ in range(2):
    job.step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.steps):
    job.step()
e1.record(); torch.cuda.synchronize()
print("restore_sff forward, %d x %d^2: %.2f ms" % (a.batch, a.size, e0.elapsed_time(e1) / a.steps))
