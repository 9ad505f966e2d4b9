#!/usr/bin/env python
"""Developer benchmark of the step shapes of SURVEY.md 8(a) a13 on one GPU (or N ranks under
torch.distributed.run): whole-IFNet interpolation forward (C2) and the SFF fusion training step (C3:
frozen FusionNet forward -> UNet -> L1 -> backward -> flat all-reduce -> Adam).  HIP-event timed."""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "sstem-restoration_amd"))
import torch  # noqa: E402
import dataparallel as dp  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--what", default="ifnet,fusion_step")
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--ifnet-batch", type=int, default=8)
ap.add_argument("--ifnet-size", type=int, default=1024)
ap.add_argument("--fusion-batch", type=int, default=16, help="GLOBAL batch (split over ranks)")
ap.add_argument("--torch-adam", action="store_true", help="torch.optim.Adam instead of the flat native update")
ap.add_argument("--graph", action="store_true", help="capture the fusion step in a HIP graph and replay it")
a = ap.parse_args()
rank, world, dev = dp.init_from_env()


def timeit(fn, n):
    fn()
    torch.cuda.synchronize(); dp.barrier()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize(); dp.barrier()
    return e0.elapsed_time(e1) / n


if "ifnet" in a.what:
    from model.model_interp import IFNet
    torch.manual_seed(555)
    net = IFNet(51).eval().to(dev)
    dp.broadcast_module(net)
    B, S = a.ifnet_batch, a.ifnet_size
    x = torch.rand(B, 6, S, S, device=dev)
    with torch.no_grad():
        ms = timeit(lambda: net(x), a.iters)
    if rank == 0:
        flop = 45.7e9 * B * (S / 256.0) ** 2
        print("SFF IFNet forward  B=%d %dx%d per GPU x %d GPU(s): %.2f ms  -> %.1f restored MP/s total, %.1f conv TFLOP/s per GPU"
              % (B, S, S, world, ms, world * B * S * S / 1e6 / (ms * 1e-3), flop / ms / 1e9), flush=True)
    del net, x
    torch.cuda.empty_cache()

if "fusion_step" in a.what:
    from model.model_fusionnet import FusionNet
    from model.model_unet import UNet
    torch.manual_seed(555)
    flow = FusionNet(6, 2, 32).eval().to(dev)
    net = UNet(6, 1).train().to(dev)
    dp.broadcast_module(flow); dp.broadcast_module(net)
    import train_utils
    flat = train_utils.FlatParams(net.parameters())
    bucket = dp.FlatGradBucket(net.parameters())
    if a.torch_adam:
        opt = torch.optim.Adam(net.parameters(), lr=1e-4, betas=(0.9, 0.999), eps=1e-8, capturable=a.graph)
    else:
        opt = train_utils.FlatAdam(flat.flat, bucket.flat, lr=1e-4, betas=(0.9, 0.999), eps=1e-8)
    b = a.fusion_batch // world
    x = torch.rand(b, 6, 256, 256, device=dev); target = torch.rand(b, 1, 256, 256, device=dev)

    from utils.image_warp_torch import SpatialTransformation
    warp = SpatialTransformation(use_gpu=True)

    def step():
        with torch.no_grad():             # frozen flow predictor + back-warp of the SFF channels (main_fusion.py:227-235)
            pred_flow = flow(x)
            x[:, :3] = warp(x[:, :3].contiguous(), pred_flow.permute(0, 2, 3, 1))
        bucket.zero()
        loss = torch.nn.functional.l1_loss(net(x), target)
        loss.backward()
        bucket.allreduce_mean()
        opt.step()
    run = step
    if a.graph:
        # warm up on a side stream (allocator + lazy inits), then capture one whole step
        s_ = torch.cuda.Stream()
        s_.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s_):
            for _ in range(3):
                step()
        torch.cuda.current_stream().wait_stream(s_)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            step()
        run = graph.replay
    ms = timeit(run, a.iters)
    if rank == 0:
        print("SFF fusion step%s  global batch %d (%d per GPU x %d): %.2f ms/step -> %.1f samples/s; grad bucket %.1f MB"
              % (" [HIP graph]" if a.graph else "", a.fusion_batch, b, world, ms, a.fusion_batch / (ms * 1e-3), bucket.nbytes / 1e6), flush=True)
dp.shutdown()
