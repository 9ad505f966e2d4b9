#!/usr/bin/env python
"""bench.py -- throughput of the sepconv interpolation apply on MI355X.

Workload (BASELINE.json configs[1]): "SepConv 51-tap interpolation forward, batch=8 1024x1024
tiles, 1xMI355X".  One STEP = the interpolation apply of the SFF IFNet for a batch of 8 tiles
(reference sff_scripts_interp/model/model_interp.py:94-97):

    y   = sepconv(padded_i2, k2v, k2h) + sepconv(padded_i1, k1v, k1h)     # 2 op calls
    out = mean(y, dim=1, keepdim=True)                                     # [8,1,1024,1024]

executed the way the product's IFNet executes it at inference: ONE fused launch
(libs.sepconv.fused.interp_apply: replication padding folded into the tile staging, both local
convolutions, add and channel mean; include/sstem_sepconv.h).  `--unfused` times the reference-API
spelling instead (ReplicationPad2d outside the timed region, 2 SeparableConvolution.apply + add + mean),
with all inputs already resident in HBM.  Synthetic data (seed 555): GRAYSCALE frame pairs, each frame
replicated to 3 identical channels exactly as every caller of the reference builds the IFNet input
(inference_singleImage.py:55-61, test_fusion.py:105-106; north_star: "synthetic ... grayscale pairs"), and
softmax(randn) kernels.  `--rgb` draws three independent channels per frame instead (SURVEY.md 8d's
rand(8,3,...)): the kernels then cannot use their exact identical-channel path and do 3x the MFMA work.  `value` = restored megapixels per second = B*H*W/1e6 per step over
the whole job.  Independent tiles shard across GPUs with no data-path collective ("weak").

Extra objects on the JSON line:
  roofline      dominant kernel = the sepconv kernel; achieved = algorithmic bytes per launch / mean
                launch duration measured with HIP events on the launch stream inside the timed region;
                peak = 8000 GB/s (MI355X HBM3E spec).  Fused launch: 4*[2*B*3*H*W + 4*B*51*H*W + B*H*W]
                = 7,080,247,296 B (two images, four coefficient tensors, one output plane; SURVEY 8d
                counts 866.4 B per restored pixel for the unfused pair, the fused launch moves 844.0);
                unfused call: 4*[B*3*(H+50)(W+50) + 2*B*51*H*W + B*3*H*W] = 3,633,949,056 B.
  cpu_baseline  the CPU oracle (OpenMP build of oracle/sepconv_oracle.c, kind "port") timed on this
                box's host cores on a bounded sample (a few 1024x1024 tiles) of the same workload.

Usage: python bench.py [--gpus N] [--steps K] [--warmup W] [--size 1024] [--batch 8]
       (N > 1: launched by torch.distributed.run, one rank per GPU)
"""
import argparse
import json
import math
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(REPO, "sstem-restoration_amd"), REPO):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--prewarm-s", type=float, default=0.6, help="seconds of untimed steps before the warm-up steps (GPU clock ramp)")
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--algo", type=int, default=0, help="0 auto, 1 direct, 2 mfma")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--unfused", action="store_true", help="time the reference-API spelling (2 op calls + add + mean)")
    ap.add_argument("--rgb", action="store_true", help="three independent random channels per frame instead of a replicated grayscale frame")
    ap.add_argument("--traffic-json", default=os.path.join(REPO, "profiles", "traffic_latest.json"),
                    help="PMC-derived HBM bytes per launch written by tools/pmc_traffic.py (optional)")
    return ap.parse_args()


def make_inputs(B, S, device, seed, rgb=False):
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    ch = 3 if rgb else 1
    i1 = torch.rand(B, ch, S + 50, S + 50, device=device, generator=g).expand(B, 3, S + 50, S + 50).contiguous()
    i2 = torch.rand(B, ch, S + 50, S + 50, device=device, generator=g).expand(B, 3, S + 50, S + 50).contiguous()
    ks = [torch.softmax(torch.randn(B, 51, S, S, device=device, generator=g), dim=1) for _ in range(4)]
    return i1, i2, ks


def cpu_baseline(S, rgb, gpu_apply):
    """Oracle (CPU restatement of the reference kernel) on a bounded sample of the same workload:
    n tiles of the step (replication pad + 2 calls + add + mean each), n chosen from a short calibration so the
    sample is roughly 10-30 s of CPU work on this box's cores.  The first tile of the sample is also run through
    the GPU step being timed (`gpu_apply`, same inputs) and compared: that is the "PSNR vs ref" half of the metric."""
    import numpy as np
    from oracle import sepconv_c  # executed here only: the reported CPU baseline and the checker of one tile

    def tiles(n, rows):
        # the bench's data: grayscale frames replicated to 3 channels (or 3 independent channels with --rgb),
        # UNPADDED; softmax-normalised kernels
        rng = np.random.default_rng(555)
        ch = 3 if rgb else 1
        i1 = np.repeat(rng.random((n, ch, rows, S), dtype=np.float32), 3 // ch, axis=1)
        i2 = np.repeat(rng.random((n, ch, rows, S), dtype=np.float32), 3 // ch, axis=1)
        ks = []
        for _ in range(4):
            a = rng.standard_normal((n, 51, rows, S), dtype=np.float32)
            e = np.exp(a - a.max(axis=1, keepdims=True))
            ks.append((e / e.sum(axis=1, keepdims=True)).astype(np.float32))
        return i1, i2, ks

    def apply(i1, i2, ks):
        # model_interp.py:90-97: y = sepconv(pad(i2), k2v, k2h) + sepconv(pad(i1), k1v, k1h); mean over channels
        t0 = time.perf_counter()
        pad = ((0, 0), (0, 0), (25, 25), (25, 25))
        y = sepconv_c.forward(np.pad(i2, pad, mode="edge"), ks[2], ks[3], omp=True) + \
            sepconv_c.forward(np.pad(i1, pad, mode="edge"), ks[0], ks[1], omp=True)
        out = y.mean(axis=1, keepdims=True)
        return time.perf_counter() - t0, out

    cores = sepconv_c.num_threads(omp=True)
    apply(*tiles(1, 16))                      # warm the thread pool
    t_cal, _ = apply(*tiles(1, 64))           # 1/16 of a tile
    per_tile = t_cal * (S / 64.0)
    n = int(max(1, min(8, round(36.0 / max(per_tile, 1e-3)))))  # the 64-row calibration over-predicts ~2.5x
    i1, i2, ks = tiles(n, S)
    dt, out = apply(i1, i2, ks)
    assert out.shape == (n, 1, S, S)
    res = {"value": round(n * S * S / 1e6 / dt, 5), "unit": "megapixels/s", "cores": cores, "kind": "port",
           "sample": "%d tile(s) of 3x%dx%d: replication pad + 2 oracle sepconv calls + add + mean per tile, %d OpenMP "
                     "threads, %.1f s" % (n, S, S, cores, dt)}
    # parity of the timed GPU step on the first tile of this sample (pixels in [0,1]: peak = 1)
    got = gpu_apply(i1[:1], i2[:1], [k[:1] for k in ks])
    diff = got.astype(np.float64) - out[:1].astype(np.float64)
    mse = float((diff ** 2).mean())
    res["parity"] = {"psnr_db_vs_oracle": (round(10.0 * math.log10(1.0 / mse), 2) if mse > 0 else None),
                     "max_abs_diff": float(np.abs(diff).max()), "tile": "first tile of the sample, same inputs",
                     "tolerance": "1e-4 absolute (north_star); PSNR(gpu, oracle) >= 120 dB"}
    return res


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the sepconv op has no CPU path)")
    # Rehearsal knobs for a one-GPU box (never set by the driver): SSTEM_BENCH_SINGLE_DEVICE=1 maps every rank to
    # cuda:0 and SSTEM_BENCH_BACKEND=gloo replaces RCCL (which refuses two ranks on one device).
    if os.environ.get("SSTEM_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    backend = os.environ.get("SSTEM_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    import libs.sepconv._ext.cunnex as cunnex
    from libs.sepconv.SeparableConvolution import SeparableConvolution
    cunnex.set_algorithm(args.algo)
    lib = cunnex.load_library()

    B, S = args.batch, args.size
    i1, i2, (k1v, k1h, k2v, k2h) = make_inputs(B, S, device, 555 + rank, args.rgb)
    sep = SeparableConvolution.apply
    fused = not args.unfused
    if fused:
        from libs.sepconv.fused import interp_apply
        # the fused launch takes the UNPADDED frames; make_inputs draws (S+50)^2 images, use their centres
        u1 = i1[:, :, 25:25 + S, 25:25 + S].contiguous()
        u2 = i2[:, :, 25:25 + S, 25:25 + S].contiguous()

    n_ev = (1 if fused else 2) * args.steps
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(n_ev)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(n_ev)]

    def step(k=None):
        if fused:
            if k is None:
                return interp_apply(u1, u2, k1v, k1h, k2v, k2h)
            ev0[k].record(); out = interp_apply(u1, u2, k1v, k1h, k2v, k2h); ev1[k].record()
            return out
        if k is None:
            y = sep(i2, k2v, k2h) + sep(i1, k1v, k1h)
        else:  # timed region: HIP events around each op launch, on the launch (current) stream
            ev0[2 * k].record(); a = sep(i2, k2v, k2h); ev1[2 * k].record()
            ev0[2 * k + 1].record(); b = sep(i1, k1v, k1h); ev1[2 * k + 1].record()
            y = a + b
        return torch.mean(y, dim=1, keepdim=True)

    with torch.no_grad():
        # Untimed pre-warm, before the W warm-up steps of the contract: an MI355X that has been idle needs a few hundred ms under load
        # to reach the clocks it then holds (measured: the same 20 timed steps 2.5 % slower with 5 warm-up steps = 7 ms than with
        # 500; steps of other benchmarks timed within 100 ms of idle ran 30-50 % slow).  Nothing of it is timed.
        t_pre = time.perf_counter()
        while time.perf_counter() - t_pre < args.prewarm_s:
            out = step()
            torch.cuda.synchronize()
        for _ in range(args.warmup):
            out = step()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(args.steps):
            out = step(k)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    assert out.shape == (B, 1, S, S)

    if dist is not None:
        t = torch.tensor([dt], device=device if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    kern_ms = sum(a.elapsed_time(b) for a, b in zip(ev0, ev1)) / n_ev
    if rank == 0:
        mp_per_step = world * B * S * S / 1e6
        if fused:
            alg_bytes = 4 * (2 * B * 3 * S * S + 4 * B * 51 * S * S + B * S * S)
        else:
            alg_bytes = int(lib.sstem_sepconv_forward_bytes(B, 3, S, S))
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        # label of the kernel the launcher dispatches for this input (mirrors launch_gray / launch_interp_fused /
        # launch_fwd_mfma in csrc/sepconv_kernels.hip; SSTEM_GRAY_SHAPE is the developer override read there)
        mode = 2 if fused else 0
        if args.rgb:
            kname = "sepconv_rowmajor_mfma<2,3,8,4>" if fused else "sepconv_rowmajor_mfma<0,3,16,2>"
        else:
            gshapes = {0: "4,8,3,false,2", 1: "4,8,2,true,3", 2: "4,16,2,true,3", 3: "4,16,2,true,2",
                       4: "4,8,2,false,3", 5: "4,16,2,false,3"}
            forced = os.environ.get("SSTEM_GRAY_SHAPE")
            gs = int(forced) if forced is not None else (3 if B * ((S + 63) // 64) * ((S + 63) // 64) >= 1024 else 0)
            kname = "sepconv_gray_mfma<%d,%s>" % (mode, gshapes.get(gs, gshapes[0]))
        # PMC traffic is only reported when it was collected for THIS kernel on THIS workload
        traffic = None
        try:
            with open(args.traffic_json) as f:
                tj = json.load(f)
            if tj.get("batch") == B and tj.get("size") == S and tj.get("fused", False) == fused \
                    and tj.get("rgb", False) == args.rgb and tj.get("kernel_label") == kname:
                traffic = tj.get("hbm_bytes_per_launch")
        except (OSError, ValueError):
            pass
        line = {
            "metric": "restored megapixels/sec (interp+fusion fwd) at 1024x1024; PSNR vs ref",
            "value": round(mp_per_step * args.steps / dt, 3),
            "unit": "megapixels/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic" + (" (independent channels)" if args.rgb else " (grayscale frame pairs replicated to 3 channels)"),
            "config": {"workload": "SepConv 51-tap interpolation forward (SFF IFNet apply: replication pad + 2 sepconv "
                                   "+ add + channel mean%s), batch=%d %dx%d tiles per GPU, inputs resident in HBM"
                                   % (", one fused launch" if fused else ", reference-API spelling: 2 op calls", B, S, S),
                       "batch_per_gpu": B, "tile": [S, S], "channels": 3, "taps": 51,
                       "sharding": "independent tiles per GPU, no data-path collective",
                       "frames": "rgb-noise" if args.rgb else "grayscale x3",
                       "algo": {0: "auto", 1: "direct", 2: "mfma"}[args.algo], "prewarm_s": args.prewarm_s},
            "roofline": {"bound": "hbm",
                         "kernel": kname + (" (fused interpolation apply" if fused else " (sepconv forward")
                                   + ("; launch time includes the channel-comparison kernel and the no-op generic launch)" if not args.rgb else ")"),
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "algorithmic_bytes_per_launch": alg_bytes, "launch_ms": round(kern_ms, 4)},
        }
        if world == 1 and not args.no_cpu_baseline:
            def gpu_apply(a1, a2, kk):     # the step being timed, on the baseline sample's inputs (unpadded frames)
                t = [torch.from_numpy(x).to(device) for x in (a1, a2, *kk)]
                with torch.no_grad():
                    if fused:
                        o = interp_apply(t[0], t[1], t[2], t[3], t[4], t[5])
                    else:
                        padf = torch.nn.ReplicationPad2d(25)
                        o = torch.mean(sep(padf(t[1]).contiguous(), t[4], t[5]) + sep(padf(t[0]).contiguous(), t[2], t[3]),
                                       dim=1, keepdim=True)
                torch.cuda.synchronize()
                return o.cpu().numpy()
            line["cpu_baseline"] = cpu_baseline(S, args.rgb, gpu_apply)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
