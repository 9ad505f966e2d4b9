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
ap.add_argument("--what", default="ifnet,fusion_step,sp_joint_step")
ap.add_argument("--sp-batch", type=int, default=16, help="GLOBAL batch of the SP joint step")
ap.add_argument("--sp-size", type=int, default=256)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--ifnet-batch", type=int, default=8)
ap.add_argument("--ifnet-size", type=int, default=1024)
ap.add_argument("--fusion-batch", type=int, default=16, help="GLOBAL batch (split over ranks)")
ap.add_argument("--ifnet-step-batch", type=int, default=8, help="GLOBAL batch of the IFNet training step (config 5: 64 over 8 GPUs = 8 per GPU)")
ap.add_argument("--torch-adam", action="store_true", help="torch.optim.Adam instead of the flat native update")
ap.add_argument("--graph", action="store_true", help="capture the fusion step in a HIP graph and replay it")
ap.add_argument("--rgb-noise", action="store_true", help="IFNet forward on six independent random channels instead of two replicated grayscale frames")
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


def timeit(fn, n):
    # warm up for at least 3 steps AND 0.7 s: allocator, workspaces and packed-weight caches settle, and the chip reaches its clocks
    # (a step timed within ~100 ms of an idle GPU ran 30-50 % slow: IFNet step 10-12 ms alone against 7.5 ms after another benchmark)
    import time
    t0 = time.time(); k = 0
    while k < 3 or time.time() - t0 < 0.7:
        fn(); torch.cuda.synchronize(); k += 1
    dp.barrier()
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
    if a.rgb_noise:     # six independent random channels: the generic (three-channel) sepconv kernels
        x = torch.rand(B, 6, S, S, device=dev)
    else:               # what every caller feeds (inference_singleImage.py:55-66): two grayscale frames, each replicated x3
        f = torch.rand(B, 2, S, S, device=dev)
        x = torch.cat((f[:, :1].expand(B, 3, S, S), f[:, 1:].expand(B, 3, S, S)), 1).contiguous()
    with torch.no_grad():
        ms = timeit(lambda: net(x), a.iters)
    if rank == 0:
        flop = 45.7e9 * B * (S / 256.0) ** 2
        print("SFF IFNet forward [" + PREC + "] (%s)  B=%d %dx%d per GPU x %d GPU(s): %.2f ms  -> %.1f restored MP/s total, %.1f conv TFLOP/s per GPU"
              % ("rgb noise" if a.rgb_noise else "gray frames x3", B, S, S, world, ms, world * B * S * S / 1e6 / (ms * 1e-3), flop / ms / 1e9), flush=True)
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
        print("SFF fusion step [" + PREC + "]%s  global batch %d (%d per GPU x %d): %.2f ms/step -> %.1f samples/s; grad bucket %.1f MB"
              % (" [HIP graph]" if a.graph else "", a.fusion_batch, b, world, ms, a.fusion_batch / (ms * 1e-3), bucket.nbytes / 1e6), flush=True)
if "ifnet_step" in a.what:
    # BASELINE config 5, in fp32: sff_scripts_interp/main_ms.py:187-206 -- IFNet -> L1 -> backward -> gradient all-reduce -> Adam
    # (lr 1e-3), global batch 64 = 8 per GPU at 8 GPUs, 256x256, two grayscale frames replicated x3.  (The config asks for bf16
    # conv activations with fp32 sepconv accumulation; the conv kernels here are fp32 -- DESIGN.md section 7.)
    from model.model_interp import IFNet
    import train_utils
    torch.manual_seed(555)
    net = IFNet(51).train().to(dev)
    dp.broadcast_module(net)
    flat = train_utils.FlatParams(net.parameters())
    bucket = dp.FlatGradBucket(net.parameters())
    opt = train_utils.FlatAdam(flat.flat, bucket.flat, lr=1e-3, betas=(0.9, 0.999), eps=1e-8)
    b = a.ifnet_step_batch // world
    f = torch.rand(b, 2, 256, 256, device=dev)
    x = torch.cat((f[:, :1].expand(b, 3, 256, 256), f[:, 1:].expand(b, 3, 256, 256)), 1).contiguous()
    target = torch.rand(b, 1, 256, 256, device=dev)

    def ifnet_step():
        bucket.zero()
        loss = torch.nn.functional.l1_loss(net(x), target)
        loss.backward()
        bucket.allreduce_mean()
        opt.step()
    ms = timeit(ifnet_step, a.iters)
    if rank == 0:
        flop = 3 * 45.7e9 * b
        print("SFF IFNet training step (" + PREC + ")  global batch %d (%d per GPU x %d) 256x256: %.2f ms/step -> %.1f samples/s, %.1f conv TFLOP/s per GPU "
              "(3x forward flops); grad bucket %.1f MB" % (a.ifnet_step_batch, b, world, ms, a.ifnet_step_batch / (ms * 1e-3), flop / ms / 1e9, bucket.nbytes / 1e6), flush=True)
    del net, flat, bucket, opt
    torch.cuda.empty_cache()

if "sp_joint_step" in a.what:
    # sp_scripts_train/main_fusion.py:178-257: IFNet x2 (same input, two passes), UNet x2, FusionNet x2, six L1
    # losses, one backward (the only step that runs the sepconv backward kernels with the U-Nets), three Adams.
    import networks
    import train_utils
    torch.manual_seed(555)
    vfi = networks.IFNet().train().to(dev); den = networks.UNet(1, 1).train().to(dev); fus = networks.FusionNet(1, 1).train().to(dev)
    buckets, opts = [], []
    for m, lr in ((vfi, 1e-4 * 1e-20), (den, 1e-4 * 1e-6), (fus, 1e-4)):     # config/train_fusion.yaml:13,15 lr scales
        dp.broadcast_module(m)
        flat = train_utils.FlatParams(m.parameters())
        bk = dp.FlatGradBucket(m.parameters())
        buckets.append(bk); opts.append(train_utils.FlatAdam(flat.flat, bk.flat, lr=lr))
    b = a.sp_batch // world
    S = a.sp_size
    im = [torch.rand(b, 1, S, S, device=dev) for _ in range(6)]      # img_1, img_2, img_2_degra, img_3, img_3_degra, img_4
    mk = [(torch.rand(b, 1, S, S, device=dev) > 0.5).float() for _ in range(2)]
    l1 = torch.nn.functional.l1_loss

    def sp_step():
        for bk in buckets:
            bk.zero()
        inputs_vfi = torch.cat((im[0], im[0], im[0], im[5], im[5], im[5]), 1)
        vfi_pred1 = torch.unsqueeze(vfi(inputs_vfi)[:, 0], 1)
        vfi_pred2 = torch.unsqueeze(vfi(inputs_vfi)[:, 1], 1)
        d1 = den(im[2]); d2 = den(im[4])
        pred1 = fus(vfi_pred1 * (1 - mk[0]), d1 * mk[0])
        pred2 = fus(vfi_pred2 * (1 - mk[1]), d2 * mk[1])
        loss = (l1(vfi_pred1, im[1]) + l1(d1, im[1]) + l1(pred1, im[1])) + (l1(vfi_pred2, im[3]) + l1(d2, im[3]) + l1(pred2, im[3]))
        loss.backward()
        for bk, op in zip(buckets, opts):
            bk.allreduce_mean(); op.step()
    ms = timeit(sp_step, max(2, a.iters // 2))
    if rank == 0:
        print("SP joint step [" + PREC + "]  global batch %d (%d per GPU x %d) %dx%d: %.1f ms/step -> %.1f samples/s; grad buckets %s MB"
              % (a.sp_batch, b, world, S, S, ms, a.sp_batch / (ms * 1e-3), "/".join("%.1f" % (k.nbytes / 1e6) for k in buckets)), flush=True)
dp.shutdown()
