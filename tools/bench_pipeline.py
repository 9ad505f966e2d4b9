#!/usr/bin/env python
"""Developer benchmark of BASELINE config #4: the SP full pipeline (interp + correction + fusion) on
2048x2048 tile sets, tile-sharded over the ranks (1 GPU here; N under torch.distributed.run)."""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "sstem-restoration_amd"))
import torch  # noqa: E402
import dataparallel as dp  # noqa: E402
import sp_pipeline  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=2048)
ap.add_argument("--tiles", type=int, default=2, help="tile sets per GPU")
ap.add_argument("--bf16", action="store_true", help="3x3 convolutions under the opt-in bf16-operand id (fp32 tensors, fp32 sums)")
a = ap.parse_args()
rank, world, dev = dp.init_from_env()
if a.bf16:
    import hipnn.functional as HF
    HF.set_algorithm(HF.ALGO_MFMA_BF16)
torch.manual_seed(555)
models = sp_pipeline.build_models(dev)
for m in models.values():
    dp.broadcast_module(m)
S = a.size
g = torch.Generator(device=dev); g.manual_seed(555 + rank)


def tile_set():
    im = [torch.rand(1, 1, S, S, device=dev, generator=g) for _ in range(4)]
    mk = [(torch.rand(1, 1, S, S, device=dev, generator=g) > 0.5).float() for _ in range(2)]
    return (im[0], im[1], mk[0], im[2], mk[1], im[3])


sets = [tile_set() for _ in range(a.tiles * world)]
sp_pipeline.restore_sharded(models, sets[:world], rank, world)      # warm-up
torch.cuda.synchronize(); dp.barrier()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
out = sp_pipeline.restore_sharded(models, sets, rank, world)
e1.record(); torch.cuda.synchronize(); dp.barrier()
ms = e0.elapsed_time(e1)
if rank == 0:
    n = len(sets)
    print("SP pipeline %dx%d: %d tile sets on %d GPU(s) in %.1f ms -> %.1f ms per tile set per GPU, %.2f restored MP/s total "
          "(2 restored images per set)" % (S, S, n, world, ms, ms / a.tiles, 2 * n * S * S / 1e6 / (ms * 1e-3)), flush=True)
dp.shutdown()
