#!/usr/bin/env python
"""Developer aid: the SP full pipeline (sp_pipeline.restore_tile_set, one 2048^2 tile set, eval) a few times under the default algorithm ids,
for `rocprofv3 --kernel-trace --output-format csv` + tools/step_timeline.py <trace> <marker>.  python tools/sp_pipeline_loop.py [--steps 4]"""
import argparse, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sstem-restoration_amd"))
import torch
import sp_pipeline as SP
ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=4); ap.add_argument("--size", type=int, default=2048)
a = ap.parse_args()
dev = torch.device("cuda"); torch.manual_seed(555)
models = SP.build_models(dev)
g = torch.Generator(device=dev); g.manual_seed(555)
S2 = a.size
im = [torch.rand(1, 1, S2, S2, device=dev, generator=g) for _ in range(4)]
mk = [(torch.rand(1, 1, S2, S2, device=dev, generator=g) > 0.5).float() for _ in range(2)]
ts = (im[0], im[1], mk[0], im[2], mk[1], im[3])
with torch.no_grad():
    for _ in range(2):
        SP.restore_tile_set(models, *ts)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.steps):
        SP.restore_tile_set(models, *ts)
        torch.zeros(1, device=dev).sign_()          # a marker launch between tile sets (sign_kernel)
    e1.record(); torch.cuda.synchronize()
print("restore_tile_set, %d^2: %.2f ms" % (S2, e0.elapsed_time(e1) / a.steps))
