#!/usr/bin/env python
"""Developer benchmark of the step shapes of SURVEY.md 8(a) a13 (the objects of sstem-restoration_amd/steps.py) on one GPU,
or on N ranks under torch.distributed.run: whole-IFNet interpolation forward (C2), the SFF fusion training step (C3), the
SFF IFNet training step (C5 share) and the SP joint step.  HIP-event timed."""
import argparse
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "sstem-restoration_amd"))
import torch  # noqa: E402
import dataparallel as dp  # noqa: E402
import steps  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--what", default="ifnet,fusion_step,sp_joint_step")
ap.add_argument("--sp-batch", type=int, default=16, help="GLOBAL batch of the SP joint step")
ap.add_argument("--sp-size", type=int, default=256)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--ifnet-batch", type=int, default=8)
ap.add_argument("--ifnet-size", type=int, default=1024)
ap.add_argument("--fusion-batch", type=int, default=16, help="GLOBAL batch (split over ranks)")
ap.add_argument("--ifnet-step-batch", type=int, default=8, help="GLOBAL batch of the IFNet training step (config 5: 64 over 8 GPUs = 8 per GPU)")
ap.add_argument("--graph", action="store_true", help="replay forward+backward of the training steps from a HIP graph (train_utils.GraphedCallable)")
ap.add_argument("--prefetch-flow", action="store_true", help="fusion step: flow net of the next batch on a second stream")
ap.add_argument("--bf16", action="store_true", help="3x3 convolutions under the opt-in bf16-operand id (BASELINE config 5); tensors stay fp32")
ap.add_argument("--bf16-fp32-wgrad", action="store_true", help="with --bf16: keep the fp32 weight-gradient kernel")
a = ap.parse_args()
a.what = set(a.what.split(","))     # exact names (a substring test once ran `ifnet` inside `ifnet_step` profiles)
rank, world, dev = dp.init_from_env()
if a.bf16:
    import hipnn.functional as HF
    HF.set_algorithm(HF.ALGO_MFMA_BF16)
    HF.set_bf16_weight_gradient(not a.bf16_fp32_wgrad)
PREC = "bf16 conv operands" if a.bf16 else "fp32"
G = " [HIP graph]" if a.graph else ""


def timeit(fn, n):
    # warm up for at least 3 steps AND 0.7 s: allocator, workspaces and packed-weight caches settle, and the chip reaches its clocks
    # (a step timed within ~100 ms of an idle GPU ran 30-50 % slow: IFNet step 10-12 ms alone against 7.5 ms after another benchmark)
    # (several ranks: the same step count everywhere -- a step may contain a collective, so a time-based loop would let one rank enter an
    # all-reduce its neighbours never reach; three probe steps, then 0.7 s worth by the slowest rank's clock)
    t0 = time.time()
    for _ in range(3):
        fn(); torch.cuda.synchronize()
    per = (time.time() - t0) / 3
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([per], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        per = float(t.item())
    for _ in range(max(0, min(2000, int(0.7 / max(per, 1e-5)) - 3))):
        fn()
    torch.cuda.synchronize()
    dp.barrier()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize(); dp.barrier()
    return e0.elapsed_time(e1) / n


if "ifnet" in a.what:
    fw = steps.IFNetForward(dev, batch=a.ifnet_batch, size=a.ifnet_size)
    ms = timeit(fw.step, a.iters)
    if rank == 0:
        B, S = a.ifnet_batch, a.ifnet_size
        print("SFF IFNet forward [" + PREC + "] (gray frame pairs)  B=%d %dx%d per GPU x %d GPU(s): %.2f ms  -> %.1f restored MP/s total, %.1f conv TFLOP/s per GPU"
              % (B, S, S, world, ms, world * B * S * S / 1e6 / (ms * 1e-3), fw.flop_per_step() / ms / 1e9), flush=True)
    del fw
    torch.cuda.empty_cache()

if "fusion_step" in a.what:
    st = steps.FusionStep(dev, global_batch=a.fusion_batch, graph=a.graph, prefetch_flow=a.prefetch_flow)
    ms = timeit(st.step, a.iters)
    ar = st.time_allreduce()
    if rank == 0:
        print("SFF fusion step [" + PREC + "]%s  global batch %d (%d per GPU x %d): %.2f ms/step -> %.1f samples/s, %.1f conv TFLOP/s per GPU; "
              "grad bucket %.1f MB, all-reduce %.3f ms" % (G, a.fusion_batch, st.batch, world, ms, a.fusion_batch / (ms * 1e-3),
                                                            st.flop_per_step() / ms / 1e9, st.bucket_bytes[0] / 1e6, ar), flush=True)
    del st
    torch.cuda.empty_cache()

if "ifnet_step" in a.what:
    st = steps.IFNetStep(dev, global_batch=a.ifnet_step_batch, graph=a.graph)
    ms = timeit(st.step, a.iters)
    if rank == 0:
        print("SFF IFNet training step (" + PREC + ")%s  global batch %d (%d per GPU x %d) 256x256: %.2f ms/step -> %.1f samples/s, %.1f conv TFLOP/s per GPU "
              "(3x forward flops); grad bucket %.1f MB" % (G, a.ifnet_step_batch, st.batch, world, ms, a.ifnet_step_batch / (ms * 1e-3),
                                                           st.flop_per_step() / ms / 1e9, st.bucket_bytes[0] / 1e6), flush=True)
    del st
    torch.cuda.empty_cache()

if "sp_joint_step" in a.what:
    st = steps.SPJointStep(dev, global_batch=a.sp_batch, size=a.sp_size, graph=a.graph)
    ms = timeit(st.step, max(2, a.iters // 2))
    if rank == 0:
        print("SP joint step [" + PREC + "]%s  global batch %d (%d per GPU x %d) %dx%d: %.1f ms/step -> %.1f samples/s; grad buckets %s MB"
              % (G, a.sp_batch, st.batch, world, a.sp_size, a.sp_size, ms, a.sp_batch / (ms * 1e-3),
                 "/".join("%.1f" % (k / 1e6) for k in st.bucket_bytes)), flush=True)
dp.shutdown()
