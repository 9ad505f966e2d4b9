#!/usr/bin/env python
"""bench.py -- throughput of the sepconv interpolation apply on MI355X (+ the step shapes north_star scores).

Headline workload (BASELINE.json configs[1]): "SepConv 51-tap interpolation forward, batch=8 1024x1024
tiles, 1xMI355X".  One STEP = the interpolation apply of the SFF IFNet for a batch of 8 tiles
(reference sff_scripts_interp/model/model_interp.py:94-97):

    y   = sepconv(padded_i2, k2v, k2h) + sepconv(padded_i1, k1v, k1h)     # 2 op calls
    out = mean(y, dim=1, keepdim=True)                                     # [8,1,1024,1024]

executed the way the product executes it at inference (IFNet.interpolate_gray, the CLI, sp_pipeline): ONE fused
launch on the two grayscale planes (libs.sepconv.fused.interp_apply_gray: replication padding folded into the tile
staging, both local convolutions, add and channel mean; include/sstem_sepconv.h) -- every caller of the reference
builds the network input by replicating one plane x3 (inference_singleImage.py:55-61, test_fusion.py:105-106;
north_star: "synthetic ... grayscale pairs").  The four coefficient tensors are resident in the row-segment layout [B,H,W/64,51,64] that the IFNet's kernel heads store at inference
(include/sstem_sepconv.h, "blocked coefficients": the same values, the same bits out of the apply; --nchw = the operator API's layout).
Spellings of the same step, for comparison:
  --replicated   the frames as [B,3,H,W] replicated tensors through the generic fused entry point (device-side
                 channel comparison + dispatch inside the timed span; bit-identical result)
  --rgb          three independent random channels per frame (SURVEY.md 8d's rand(8,3,...)): no identical-channel
                 path, 3x the MFMA work
  --unfused      the reference-API spelling (ReplicationPad2d outside the timed region, 2 SeparableConvolution.apply
                 + add + mean)
`value` = restored megapixels per second = B*H*W/1e6 per step over the whole job, inputs resident in HBM.
Independent tiles shard across GPUs with no data-path collective ("weak").

Extra objects on the JSON line:
  roofline      dominant kernel = the sepconv kernel; achieved = algorithmic bytes per launch / mean launch duration
                measured with HIP events on the launch stream inside the timed region; peak = 8000 GB/s (MI355X HBM3E).
                Fused launch: 4*[2*B*P*H*W + 4*B*51*H*W + B*H*W] with P = planes per frame handed over
                (1: 6,945,767,424 B at B=8 1024x1024; 3: 7,079,985,152 B; SURVEY 8d counts 866.4 B per restored pixel
                for the unfused pair); unfused call: 4*[B*3*(H+50)(W+50) + 2*B*51*H*W + B*3*H*W] = 3,633,949,056 B.
  cpu_baseline  the CPU oracle (OpenMP build of oracle/sepconv_oracle.c, kind "port") timed on this box's host cores
                on a bounded sample of the same workload; its first tile also goes through the TIMED GPU step (as image 0
                of a full batch, so the same kernel instance runs) and is compared: the "PSNR vs ref" half of the metric.
  extra         what else north_star asks to see, each with its own roofline (skip with --no-extra):
                the apply at 256x256 (B=8 and B=64); the whole SFF IFNet forward at the headline size (MFMA roofline,
                fp32 matrix peak); the SFF fusion TRAINING step of BASELINE config 3 (global batch 16 at 256x256
                STRONG-scaled over the ranks: frozen flow net -> warp -> UNet -> L1 -> backward -> one flat RCCL
                all-reduce -> Adam), with the all-reduce time and bucket size.

Usage: python bench.py [--gpus N] [--steps K] [--warmup W] [--size 1024] [--batch 8]
       N > 1 without WORLD_SIZE in the environment: this process only spawns `python -m torch.distributed.run` with N
       ranks (before it touches a GPU) and returns its exit code; under torch.distributed.run it is one rank of N.
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

# OpenMP worker threads (the CPU oracle's 128, torch's own) spin for a while after every parallel region by default; spinning threads
# take the cores the Python thread issuing the eager training-step launches runs on (the 2-sample fusion step read 5.7 ms behind the
# cpu_baseline leg against 3.9 ms in a fresh process)
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")

REPO = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(REPO, "sstem-restoration_amd"), REPO):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TF = 157.3   # same guide: dense fp32 matrix peak
MFMA_BF16_PEAK_TF = 2500.0 # same guide: dense bf16 matrix peak
# hipnn's ALGO_AUTO runs the large 3x3 layers on the split-bf16 X6 kernel (fp32 operands as three bf16 pieces, SIX bf16 MFMAs per
# product term: csrc/conv_split_kernels.hip), so the ceiling of an fp32 convolution flop there is the bf16 peak / 6
X6_FP32_EQUIV_PEAK_TF = MFMA_BF16_PEAK_TF / 6.0
# ... and its launches nothing is recorded for (inference networks, the frozen flow net) on the fp16 two-piece id: THREE fp16 MFMAs per
# product term (the dense fp16 matrix peak equals the bf16 one)
F16X3_FP32_EQUIV_PEAK_TF = MFMA_BF16_PEAK_TF / 3.0


def conv_roofline(tf, flop, inference_share=0.0, full=False):
    """roofline object of an entry whose time is 3x3 convolutions under hipnn's ALGO_AUTO: fp32-equivalent TFLOP/s against the ceiling
    of the split kernels -- the 16-bit matrix peak / 3 where the launches run on two fp16 pieces (F16X3: launches nothing is recorded
    for, and since round 5 the recorded ones too), / 6 where they run on three bf16 pieces (X6: recorded launches with
    SSTEM_CONV_AUTO_F16_TRAIN=0); inference_share = the fraction of the entry's flops nothing is recorded for (the ceilings combine by
    time: 1 / (share / p3 + (1 - share) / p6)).  `conv` names the ids in one word (DESIGN 4d / 4e spell them out); frac_fp32_mfma keeps
    the round-1 denominator (the fp32 matrix peak)."""
    import hipnn.functional as HF
    split = HF.get_algorithm() == HF.ALGO_AUTO and HF._AUTO_SPLIT
    f16_train = split and HF._AUTO_F16 and HF._AUTO_F16_TRAIN
    f16 = split and HF._AUTO_F16 and inference_share > 0.0
    if not split:
        peak, conv = MFMA_F32_PEAK_TF, "fp32-mfma"
    elif f16_train:
        peak, conv = F16X3_FP32_EQUIV_PEAK_TF, "f16x3"
    elif not f16:
        peak, conv = X6_FP32_EQUIV_PEAK_TF, "x6"
    else:
        peak = 1.0 / (inference_share / F16X3_FP32_EQUIV_PEAK_TF + (1.0 - inference_share) / X6_FP32_EQUIV_PEAK_TF)
        conv = "f16x3" if inference_share >= 1.0 else "f16x3 %.0f%% + x6" % (100.0 * inference_share)
    r = {"bound": "mfma", "conv": conv, "achieved": round(tf, 1), "peak": round(peak, 1), "frac": round(tf / peak, 4)}
    if full:          # (the extra entries stay short: the whole JSON line must fit the driver's 8 KB stdout tail)
        r["frac_fp32_mfma"] = round(tf / MFMA_F32_PEAK_TF, 4)
        r["flop_per_step"] = flop
    return r


EXTRAS = "apply256,apply_spellings,sepconv_backward,ifnet_forward,fusion_step,ifnet_step,sp_joint_step,sp_pipeline"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--prewarm-s", type=float, default=0.6, help="seconds of untimed steps before the warm-up steps (GPU clock ramp)")
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--algo", type=int, default=0, help="0 auto, 1 direct, 2 mfma")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="no `extra` entries (profiling runs)")
    ap.add_argument("--no-metric-as-worded", action="store_true", help="skip the top-level metric_as_worded object (the SFF restoration forward + its CPU baseline)")
    ap.add_argument("--extra-timeout", type=int, default=300, help="seconds after which the extra entries are abandoned and the headline line is printed alone")
    ap.add_argument("--extra-only", default=None, help="comma list of extra entries to run: " + EXTRAS)
    ap.add_argument("--fusion-batch", type=int, default=16, help="GLOBAL batch of the fusion training step (split over ranks)")
    ap.add_argument("--fusion-graph", action="store_true", help="fusion step with forward+backward replayed from a HIP graph (train_utils.GraphedCallable); "
                    "default eager: since the launch-count work of round 2 the eager step is GPU-bound (5.53 vs 5.51 ms at 2 per GPU)")
    ap.add_argument("--unfused", action="store_true", help="time the reference-API spelling (2 op calls + add + mean)")
    ap.add_argument("--replicated", action="store_true", help="frames as [B,3,H,W] replicated tensors through the generic fused entry point")
    ap.add_argument("--nchw", action="store_true", help="coefficient tensors as NCHW [B,51,H,W] (the operator API's layout) instead of the row-segment "
                    "layout [B,H,ceil(W/64),51,64] the IFNet's kernel heads store at inference (include/sstem_sepconv.h); bit-identical result")
    ap.add_argument("--rgb", action="store_true", help="three independent random channels per frame instead of a replicated grayscale frame")
    ap.add_argument("--traffic-json", default=os.path.join(REPO, "profiles", "traffic_latest.json"),
                    help="PMC-derived HBM bytes per launch written by tools/pmc_traffic.py (the fallback when the live passes fail)")
    ap.add_argument("--no-live-traffic", action="store_true", help="do not run the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of the headline "
                    "launch as child processes before the timed run; roofline.traffic then comes from --traffic-json")
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` run plainly: start N ranks as a child torch.distributed.run and pass its exit code on.
    Nothing in this process has touched a GPU (torch.cuda.device_count() does not initialise one on this image)."""
    import torch
    single = os.environ.get("SSTEM_BENCH_SINGLE_DEVICE") == "1"
    have = torch.cuda.device_count()
    if have < args.gpus and not single:
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible" % (args.gpus, have))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def live_traffic(args):
    """HBM bytes per headline launch, MEASURED in this run: two child processes run this script's headline (no extras, no CPU baseline,
    a handful of steps) under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` -- separate passes, counters only, as
    /opt/skills/guides/MI355X_MICROARCH.md prescribes -- BEFORE this process touches the GPU (a process that has initialised the GPU must
    not start another program on these boxes).  Corrections as tools/pmc_traffic.py: both counters in KiB, FETCH_SIZE counts half of a
    streaming read on gfx950.  Returns (bytes per launch, description) or (None, why not)."""
    import csv
    import glob
    import shutil
    import tempfile
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None, "rocprofv3 not found"
    needle = "sepconv_rgb_stream_mfma" if (args.rgb and not args.unfused) else ("sepconv_rowmajor_mfma" if args.rgb else "sepconv_gray_mfma")
    tmp = tempfile.mkdtemp(prefix="sstem_pmc_", dir="/tmp")
    child = [sys.executable, os.path.abspath(__file__), "--no-extra", "--no-metric-as-worded", "--no-cpu-baseline", "--no-live-traffic", "--prewarm-s", "0", "--warmup", "2",
             "--steps", "5", "--batch", str(args.batch), "--size", str(args.size), "--algo", str(args.algo)]
    for flag in ("unfused", "replicated", "nchw", "rgb"):
        if getattr(args, flag):
            child.append("--" + flag)
    vals = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            r = subprocess.run([rocprof, "--pmc", counter, "--output-format", "csv", "-d", d, "--"] + child, cwd="/tmp",
                               env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=90)
            rows = []
            for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row["Counter_Name"] == counter and needle in row["Kernel_Name"]:
                        rows.append(float(row["Counter_Value"]))
            if r.returncode != 0 or not rows:
                return None, "rocprofv3 --pmc %s pass failed (rc %d, %d rows)" % (counter, r.returncode, len(rows))
            vals[counter] = (sum(rows) / len(rows), len(rows))
    except Exception as exc:       # noqa: BLE001  (never cost the benchmark: the caller falls back to the recorded figure)
        return None, "%s: %s" % (type(exc).__name__, str(exc)[:120])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    nbytes = int(vals["FETCH_SIZE"][0] * 1024 * 2 + vals["WRITE_SIZE"][0] * 1024)
    return nbytes, "measured in this run: rocprofv3 --pmc FETCH_SIZE (x1024 x2) / WRITE_SIZE (x1024) child passes, %d launches each" % vals["FETCH_SIZE"][1]


def make_inputs(B, S, device, seed, rgb=False):
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    ch = 3 if rgb else 1
    g1 = torch.rand(B, ch, S + 50, S + 50, device=device, generator=g)
    g2 = torch.rand(B, ch, S + 50, S + 50, device=device, generator=g)
    ks = [torch.softmax(torch.randn(B, 51, S, S, device=device, generator=g), dim=1) for _ in range(4)]
    return g1, g2, ks


def physical_cores():
    try:
        import psutil
        n = psutil.cpu_count(logical=False)
        if n:
            return int(n)
    except Exception:
        pass
    return None


def cpu_baseline(S, rgb, gpu_apply_first_of_batch):
    """Oracle (CPU restatement of the reference kernel) on a bounded sample of the same workload:
    n tiles of the step (replication pad + 2 calls + add + mean each), n chosen from a short calibration so the
    sample is roughly 10-30 s of CPU work on this box's cores.  The first tile of the sample is also run through
    the GPU step being timed (as image 0 of a full batch) and compared: the "PSNR vs ref" half of the metric."""
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

    threads = sepconv_c.num_threads(omp=True)
    apply(*tiles(1, 16))                      # warm the thread pool
    t_cal, _ = apply(*tiles(1, 64))           # 1/16 of a tile
    per_tile = t_cal * (S / 64.0)
    n = int(max(1, min(8, round(36.0 / max(per_tile, 1e-3)))))  # the 64-row calibration over-predicts ~2.5x
    i1, i2, ks = tiles(n, S)
    dt, out = apply(i1, i2, ks)
    assert out.shape == (n, 1, S, S)
    res = {"value": round(n * S * S / 1e6 / dt, 5), "unit": "megapixels/s", "cores": threads,
           "physical_cores": physical_cores(), "kind": "port",
           "sample": "%d tile(s) of 3x%dx%d (pad + 2 oracle sepconv + add + mean), %d OpenMP threads, %.1f s" % (n, S, S, threads, dt)}
    # parity of the timed GPU step: the sample's first tile as image 0 of the timed batch (pixels in [0,1]: peak = 1)
    got = gpu_apply_first_of_batch(i1[:1], i2[:1], [k[:1] for k in ks])
    diff = got.astype(np.float64) - out[:1].astype(np.float64)
    mse = float((diff ** 2).mean())
    res["parity"] = {"psnr_db_vs_oracle": (round(10.0 * math.log10(1.0 / mse), 2) if mse > 0 else None),
                     "max_abs_diff": float(np.abs(diff).max()),
                     "tile": "first sample tile as image 0 of the timed batch", "tolerance": "1e-4 abs; PSNR >= 120 dB"}
    return res


def gray_kernel_label(B, S, mode, blocked=False):
    # label of the kernel the launcher dispatches (mirrors launch_gray in csrc/sepconv_kernels.hip; SSTEM_GRAY_SHAPE is the
    # developer override read there)
    gshapes = {0: "4,8,3,false,2", 1: "4,8,2,true,3", 2: "4,16,2,true,3", 3: "4,16,2,true,2",
               4: "4,8,2,false,3", 5: "4,16,2,false,3", 6: "2,8,3,false,2", 7: "4,4,3,false,2", 8: "2,4,3,false,2"}
    forced = os.environ.get("SSTEM_GRAY_SHAPE")
    gs = int(forced) if forced is not None else (3 if B * ((S + 63) // 64) * ((S + 63) // 64) >= 1024 else 0)
    if forced is None and gs == 0 and B * ((S + 63) // 64) * ((S + 31) // 32) < 512:
        gs = 7
    if forced is None and blocked and gs == 3:
        gs = 0
    # template arguments: MODE, WAVES, RPW, WPE, PFH, RING, BLK (row-segment coefficients), BF (bf16 coefficients)
    return "sepconv_gray_mfma<%d,%s,%s,false>" % (mode, gshapes.get(gs, gshapes[0]), "true" if blocked else "false")


class ApplyWorkload:
    """The headline step and its spellings on resident synthetic inputs."""

    def __init__(self, args, B, S, device, rank):
        import torch
        from libs.sepconv.SeparableConvolution import SeparableConvolution
        from libs.sepconv.fused import interp_apply, interp_apply_gray, interp_apply_gray_blocked, coef_to_blocked
        self.torch = torch
        # the product's inference path (IFNet.interpolate_gray): planes + the row-segment coefficient layout its kernel heads store
        self.bf16coef = bool(getattr(args, "bf16coef", False))       # bf16 NCHW coefficient tensors (include/sstem_sepconv.h, ..._bf16coef)
        self.blocked = not (getattr(args, "nchw", False) or args.rgb or args.replicated or args.unfused or self.bf16coef)
        self.coef_to_blocked = coef_to_blocked
        self.B, self.S, self.device = B, S, device
        self.rgb, self.unfused = args.rgb, args.unfused
        self.planes = 3 if (args.rgb or args.replicated or args.unfused) else 1
        self.sep = SeparableConvolution.apply
        self.interp_apply, self.interp_apply_gray = interp_apply, interp_apply_gray
        g1, g2, (self.k1v, self.k1h, self.k2v, self.k2h) = make_inputs(B, S, device, 555 + rank, args.rgb)
        c = slice(25, 25 + S)
        if self.unfused:           # padded 3-channel frames (the padding happens outside the timed region)
            self.i1 = g1.expand(B, 3, S + 50, S + 50).contiguous()
            self.i2 = g2.expand(B, 3, S + 50, S + 50).contiguous()
        elif self.planes == 3:     # unpadded 3-channel frames: the centres of the drawn images
            self.i1 = g1[:, :, c, c].expand(B, 3, S, S).contiguous()
            self.i2 = g2[:, :, c, c].expand(B, 3, S, S).contiguous()
        else:                      # unpadded single planes
            self.i1 = g1[:, :, c, c].contiguous()
            self.i2 = g2[:, :, c, c].contiguous()
        del g1, g2
        if self.blocked:
            self.interp_apply_gray = interp_apply_gray_blocked
            self.k1v, self.k1h, self.k2v, self.k2h = (coef_to_blocked(k) for k in (self.k1v, self.k1h, self.k2v, self.k2h))
        if self.bf16coef:
            from libs.sepconv.fused import interp_apply_gray_bf16coef
            self.interp_apply_gray = interp_apply_gray_bf16coef
            self.k1v, self.k1h, self.k2v, self.k2h = (k.bfloat16() for k in (self.k1v, self.k1h, self.k2v, self.k2h))
        self.launches_per_step = 2 if self.unfused else 1

    def alg_bytes(self, lib):
        B, S = self.B, self.S
        if self.unfused:
            return int(lib.sstem_sepconv_forward_bytes(B, 3, S, S))
        if self.bf16coef:
            return int(lib.sstem_sepconv_interp_apply_bytes_bf16coef(B, S, S, self.planes))
        return int(lib.sstem_sepconv_interp_apply_bytes(B, S, S, self.planes))

    def kernel_label(self):
        if self.rgb:
            return "sepconv_rowmajor_mfma<0,3,16,2>" if self.unfused else "sepconv_rowmajor_mfma<2,3,8,4>"
        if self.bf16coef:
            return "sepconv_gray_mfma<2,4,8,3,false,2,false,true>"                           # (..., row-segment coefficients, bf16 coefficients)
        return gray_kernel_label(self.B, self.S, 0 if self.unfused else 2, self.blocked)      # the last template argument: row-segment coefficients

    def step(self, ev=None, k=0):
        """One step; ev = (starts, ends): HIP events recorded around each op launch on the launch (current) stream."""
        torch = self.torch
        if not self.unfused:
            fn = self.interp_apply_gray if self.planes == 1 else self.interp_apply
            if ev is None:
                return fn(self.i1, self.i2, self.k1v, self.k1h, self.k2v, self.k2h)
            ev[0][k].record(); out = fn(self.i1, self.i2, self.k1v, self.k1h, self.k2v, self.k2h); ev[1][k].record()
            return out
        if ev is None:
            y = self.sep(self.i2, self.k2v, self.k2h) + self.sep(self.i1, self.k1v, self.k1h)
        else:
            ev[0][2 * k].record(); a = self.sep(self.i2, self.k2v, self.k2h); ev[1][2 * k].record()
            ev[0][2 * k + 1].record(); b = self.sep(self.i1, self.k1v, self.k1h); ev[1][2 * k + 1].record()
            y = a + b
        return torch.mean(y, dim=1, keepdim=True)

    def apply_first_of_batch(self, a1, a2, kk):
        """The timed step with image 0 of every resident tensor replaced by the given (unpadded, x3-replicated) numpy tile."""
        torch = self.torch
        S = self.S
        t1, t2 = torch.from_numpy(a1).to(self.device), torch.from_numpy(a2).to(self.device)
        ks = [torch.from_numpy(x).to(self.device) for x in kk]
        if self.unfused:
            padf = torch.nn.ReplicationPad2d(25)
            self.i1[:1] = padf(t1); self.i2[:1] = padf(t2)
        else:
            self.i1[:1] = t1[:, :self.planes]; self.i2[:1] = t2[:, :self.planes]
        if self.blocked:
            ks = [self.coef_to_blocked(x) for x in ks]
        self.k1v[:1], self.k1h[:1], self.k2v[:1], self.k2h[:1] = ks
        with torch.no_grad():
            out = self.step()
        torch.cuda.synchronize()
        assert out.shape == (self.B, 1, S, S)
        return out[:1].cpu().numpy()


def timed(torch, dist, fn_plain, fn_timed, steps, warmup, prewarm_s):
    """The contract's timing: untimed pre-warm, W warm-up steps, then exactly K steps bracketed by barrier + synchronize."""
    if dist is None:
        t_pre = time.perf_counter()
        while time.perf_counter() - t_pre < prewarm_s:
            fn_plain()
            torch.cuda.synchronize()
    elif prewarm_s > 0:
        # several ranks: the SAME number of pre-warm steps on every rank (a step may contain a collective: a rank that fits one step more
        # into its 0.7 s than its neighbour waits in an all-reduce nobody else enters -- the two-rank rehearsal of a captured step hung
        # exactly there).  One probe step, the slowest rank's time, then prewarm_s worth of steps by that clock.
        t_pre = time.perf_counter()
        fn_plain()
        torch.cuda.synchronize()
        nccl = dist.get_backend() == "nccl"
        t = torch.tensor([time.perf_counter() - t_pre], dtype=torch.float64, device="cuda" if nccl else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        n_pre = max(1, min(2000, int(prewarm_s / max(float(t.item()), 1e-5))))
        for _ in range(n_pre):
            fn_plain()
        torch.cuda.synchronize()
    for _ in range(warmup):
        fn_plain()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        fn_timed(k)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def max_over_ranks(torch, dist, dt, device, backend):
    if dist is None:
        return dt
    t = torch.tensor([dt], device=device if backend == "nccl" else "cpu", dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


LINE_LIMIT = 7800          # the driver keeps the last 8 KB of stdout: a longer line would lose its head and parse as nothing


def fit_line(line):
    """The one JSON line, guaranteed to fit the driver's stdout tail: when the entries have outgrown it, the `extra` entries give up
    first their prose (`workload`), then their roofline details, then the list is cut from the end -- and the line says so."""
    def dumps():
        return json.dumps(line, separators=(",", ":"))
    text = dumps()
    if len(text) <= LINE_LIMIT:
        return text
    extras = line.get("extra") or []
    line["extra_trimmed"] = "workload texts dropped to fit %d bytes" % LINE_LIMIT
    for e in extras:
        e.pop("workload", None)
    text = dumps()
    if len(text) <= LINE_LIMIT:
        return text
    line["extra_trimmed"] = "workload texts and roofline details dropped to fit %d bytes" % LINE_LIMIT
    for e in extras:
        r = e.get("roofline")
        if isinstance(r, dict):
            e["roofline"] = {k: r[k] for k in ("bound", "frac") if k in r}
    text = dumps()
    while len(text) > LINE_LIMIT and extras:
        extras.pop()
        line["extra_trimmed"] = "entries dropped from the end to fit %d bytes" % LINE_LIMIT
        text = dumps()
    return text


def run_entry(torch, dist, device, backend, fn, k, w=3, prewarm=0.5):
    """Seconds per step of one entry: untimed warm-up, K timed steps between barriers, max over ranks."""
    dt = timed(torch, dist, fn, lambda _k: fn(), k, w, prewarm)
    return max_over_ranks(torch, dist, dt, device, backend) / k


def fp32_mfma_only_ms(torch, dist, device, backend, fn, **kw):
    """The same entry with every 3x3 layer on the fp32 MFMA kernel (SSTEM_CONV_AUTO_SPLIT=0), for comparison; None when AUTO does
    not split in this process anyway."""
    import hipnn.functional as HF
    if not (HF.get_algorithm() == HF.ALGO_AUTO and HF._AUTO_SPLIT):
        return None
    HF._AUTO_SPLIT = False
    try:
        return round(run_entry(torch, dist, device, backend, fn, **kw) * 1e3, 3)
    finally:
        HF._AUTO_SPLIT = True


def bf16x6_train_ms(torch, dist, device, backend, fn, **kw):
    """The same training entry with its recorded launches on three bf16 pieces (X6, SSTEM_CONV_AUTO_F16_TRAIN=0: rounds 2-4), for
    comparison; None when the recorded launches are not on fp16 pieces in this process anyway.  (The step object keeps its weights:
    the pair workspaces of the other id are packed on first use.)"""
    import hipnn.functional as HF
    if not (HF.get_algorithm() == HF.ALGO_AUTO and HF._AUTO_SPLIT and HF._AUTO_F16 and HF._AUTO_F16_TRAIN):
        return None
    HF._AUTO_F16_TRAIN = False
    try:
        return round(run_entry(torch, dist, device, backend, fn, **kw) * 1e3, 3)
    finally:
        HF._AUTO_F16_TRAIN = True


def cpu_baseline_worded(torch, device, S):
    """CPU baseline of the metric AS WORDED ("interp + fusion fwd"): tests/cpu_twin.py -- the three networks' module
    trees as the stock torch.nn modules the reference builds, on torch CPU ops, + the OpenMP oracle sepconv + the numpy warp -- timed
    on this box's host cores at B = 1 256^2 (median of 3) and B = 1 SxS (median of 3); and the "PSNR vs ref" half: the GPU chain
    (sff_pipeline.restore_sff) against the twin on the 256^2 tile with the same recipe weights (tests/weight_recipe.py: outputs in
    [0,1]).  Checker / reported baseline only: nothing here is the thing measured as `value`."""
    import numpy as np
    tests_dir = os.path.join(REPO, "tests")
    if tests_dir not in sys.path:
        sys.path.insert(0, tests_dir)
    import cpu_twin                                   # tests/: benchmark + test infrastructure
    import sff_pipeline
    from oracle import sepconv_c, warp_numpy          # executed here only (the reported CPU baseline and the checker)
    from weight_recipe import cli_weights_, fill_, sff_chain_inputs, sff_flow_weights_

    cpu = sff_pipeline.build_models("cpu")
    cli_weights_(cpu["interp"], 563); sff_flow_weights_(cpu["flow"], 562); fill_(cpu["fusion"], 561)

    def sep(inp, ver, hor):
        return sepconv_c.forward(inp, ver, hor, omp=True)

    def once(size):
        prev, nxt, sff = (torch.from_numpy(a) for a in sff_chain_inputs(1, size, size))
        t0 = time.perf_counter()
        res = cpu_twin.restore_sff(cpu, prev, nxt, sff, sep, warp_numpy.warp)
        return time.perf_counter() - t0, res, (prev, nxt, sff)

    once(64)                                           # thread pools, oneDNN primitive caches
    t256 = sorted(once(256)[0] for _ in range(3))[1]
    n_big = 3 if t256 * (S / 256.0) ** 2 < 8.0 else 1      # bounded: ~10-30 s of CPU work in all (a host with few cores runs the big tile once)
    runs = [once(S)[0] for _ in range(n_big)] if S != 256 else None
    tS = sorted(runs)[len(runs) // 2] if runs else t256
    res = {"value": round(S * S / 1e6 / tS, 4), "unit": "restored megapixels/s", "kind": "port",
           "cores": torch.get_num_threads(), "physical_cores": physical_cores(), "omp_threads_sepconv": sepconv_c.num_threads(omp=True),
           "s_per_tile": {"256": round(t256, 3), str(S): round(tS, 3)},
           "sample": "tests/cpu_twin.py: IFNet + flow FusionNet + warp + UNet on torch CPU ops, oracle sepconv; B=1 256^2 (median of 3) and B=1 %d^2 (median of %d)" % (S, n_big)}
    # parity of the GPU chain on the 256^2 tile, same weights
    _, (pred_c, interp_c, flow_c, warped_c), (prev, nxt, sff) = once(256)
    gpu = sff_pipeline.build_models(device)
    for k in cpu:
        gpu[k].load_state_dict(cpu[k].state_dict())
    with torch.no_grad():
        pred_g, interp_g, flow_g, warped_g = sff_pipeline.restore_sff(gpu, prev.to(device), nxt.to(device), sff.to(device))
    torch.cuda.synchronize()
    dev = {}
    for name, g_, c_ in (("interp", interp_g, interp_c), ("flow", flow_g, flow_c), ("warped", warped_g, warped_c), ("pred", pred_g, pred_c)):
        a = g_.cpu().double().numpy(); r = c_.double().numpy()
        dev[name] = float(np.abs(a - r).max() / max(np.abs(r).max(), 1e-30))
    mse = float(((pred_g.cpu().double().numpy() - pred_c.double().numpy()) ** 2).mean())
    res["parity"] = {"psnr_db_vs_cpu_twin": round(10.0 * math.log10(1.0 / mse), 2) if mse > 0 else None,
                     "max_dev_of_range": {k: float("%.3g" % v) for k, v in dev.items()}, "tile": "1 x 256^2, recipe weights",
                     "tolerance": "1e-4 of range"}
    del gpu
    torch.cuda.empty_cache()
    return res


def metric_as_worded(args, torch, dist, device, backend, rank, world, with_cpu):
    """The metric as BASELINE.json words it -- "restored megapixels/sec (interp+fusion fwd) at 1024x1024": IFNet -> unfolding-flow
    FusionNet -> back-warp -> fusion UNet, all eval, on a batch of tiles (sff_pipeline.restore_sff; inference_singleImage.py:55-71 +
    sff_scripts_fusion/inference.py:126-153).  Its own top-level object on the JSON line (round-4 verdict: the driver's record drops
    `extra`): value, ms per step, the same step with every layer on the fp32 matrix instruction, the MFMA roofline, and -- rank 0 at
    N = 1 -- the CPU baseline of the same chain."""
    import steps as S_
    B, S = args.batch, args.size
    fw = S_.SFFRestoreForward(device, batch=B, size=S)
    sec = run_entry(torch, dist, device, backend, fw.step, k=10, w=2, prewarm=0.5)
    ms_fp32 = fp32_mfma_only_ms(torch, dist, device, backend, fw.step, k=3, w=1, prewarm=0.3)
    obj = {"workload": "SFF restoration forward (IFNet, flow FusionNet, warp, UNet; eval), %d x %d^2 per GPU" % (B, S),
           "value": round(world * B * S * S / 1e6 / sec, 2), "unit": "restored megapixels/s", "ms_per_step": round(sec * 1e3, 3),
           "ms_fp32_mfma": ms_fp32, "dtype": "f32 tensors; 3x3 products as two fp16 pieces (2^-22 per product), fp32 accumulate",
           "roofline": conv_roofline(fw.flop_per_step() / sec / 1e12, fw.flop_per_step(), inference_share=1.0, full=True)}
    del fw
    torch.cuda.empty_cache()
    if with_cpu:
        try:
            obj["cpu_baseline"] = cpu_baseline_worded(torch, device, S)
        except Exception as exc:       # noqa: BLE001  (a baseline must never cost the line)
            obj["cpu_baseline"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:200])}
    return obj


def run_extras(args, torch, dist, device, backend, rank, world, lib, which, out):
    """Entries of the `extra` list (every rank runs them; rank 0 reports).  Each: untimed warm-up, K timed steps between
    barriers, max over ranks.  Entries are kept SHORT (the whole JSON line must fit the driver's stdout tail): `workload` names the
    step in a few words (DESIGN.md 5 has the long form with the reference's file:line); convolution-bound entries say which ids
    hipnn's ALGO_AUTO ran in `roofline.conv` (f16x3 = two fp16 pieces, x6 = three bf16 pieces; fp32 tensors and accumulation
    throughout, `dtype` f32) and carry `ms_fp32_mfma`, the same entry with every layer on the fp32 matrix instruction."""
    import steps as S_
    ksteps = max(5, args.steps // 2)
    B, S = args.batch, args.size

    def run(fn, k=ksteps, w=3, prewarm=0.5):
        return run_entry(torch, dist, device, backend, fn, k, w, prewarm)

    def guarded(name, fn):
        """An extra must never cost the headline line: a failure is recorded in its entry (every rank runs the same code, so a
        failure is symmetric and no rank is left waiting in a collective)."""
        try:
            fn()
        except Exception as exc:       # noqa: BLE001
            out.append({"name": name, "error": "%s: %s" % (type(exc).__name__, str(exc)[:200])})
            torch.cuda.empty_cache()

    def hbm(nbytes, sec, kernel, launch_ms=None, launches=None):
        gbs = nbytes / sec / 1e9
        r = {"bound": "hbm", "kernel": kernel.replace("sepconv_", ""), "achieved": round(gbs), "frac": round(gbs / HBM_PEAK_GBS, 4), "bytes": nbytes}
        if launch_ms is not None:
            r["launch_ms"] = round(launch_ms, 4)
        if launches is not None and launches != 1:
            r["launches"] = launches
        return r

    def fp32_mfma_only(fn, **kw):
        return fp32_mfma_only_ms(torch, dist, device, backend, fn, **kw)

    def bf16x6_train(fn, **kw):
        return bf16x6_train_ms(torch, dist, device, backend, fn, **kw)

    def apply256():
        for Bs in (8, 64):
            a = argparse.Namespace(rgb=False, unfused=False, replicated=False, nchw=False, bf16coef=False)
            wl = ApplyWorkload(a, Bs, 256, device, rank)
            with torch.no_grad():
                sec = run(wl.step, k=max(20, args.steps), w=5, prewarm=0.3)
            out.append({"name": "apply_256", "workload": "fused apply, gray planes, %d x 256^2 per GPU" % Bs,
                        "value": round(world * Bs * 256 * 256 / 1e6 / sec, 1), "unit": "megapixels/s", "ms_per_step": round(sec * 1e3, 4),
                        "scaling": "weak", "dtype": "f32", "roofline": hbm(wl.alg_bytes(lib), sec, wl.kernel_label())})
            del wl
            torch.cuda.empty_cache()

    def apply_spellings():
        """The headline step's other spellings at the headline size, per-launch time from HIP events inside the timed region: three
        INDEPENDENT channels per frame (the op as kernel.cu:25-52 defines it), the reference-API spelling (padding outside the timed
        region, 2 SeparableConvolution.apply + add + mean) and the fused apply on NCHW coefficient tensors."""
        for name, flags, what in (
                ("apply_rgb_1024", dict(rgb=True), "fused apply, 3 independent channels per frame"),
                ("sepconv_forward_op_1024", dict(unfused=True), "reference API (2 op calls + add + mean), gray x3"),
                ("sepconv_forward_op_rgb_1024", dict(unfused=True, rgb=True), "reference API, 3 independent channels"),
                ("apply_nchw_1024", dict(nchw=True), "fused apply, gray planes, NCHW coefficients"),
                ("apply_bf16coef_1024", dict(bf16coef=True), "fused apply, gray planes, bf16 NCHW coefficients (bf16 byte model)")):
            a = argparse.Namespace(rgb=False, unfused=False, replicated=False, nchw=False, bf16coef=False)
            for k_, v_ in flags.items():
                setattr(a, k_, v_)
            wl = ApplyWorkload(a, B, S, device, rank)
            k = max(20, args.steps // 4)
            n_ev = wl.launches_per_step * k
            ev = ([torch.cuda.Event(enable_timing=True) for _ in range(n_ev)], [torch.cuda.Event(enable_timing=True) for _ in range(n_ev)])
            with torch.no_grad():
                dt = timed(torch, dist, wl.step, lambda i: wl.step(ev, i), k, 3, 0.3)
            sec = max_over_ranks(torch, dist, dt, device, backend) / k
            launch_ms = sum(x.elapsed_time(y) for x, y in zip(*ev)) / n_ev
            out.append({"name": name, "workload": "%s, %d x %d^2 per GPU" % (what, B, S),
                        "value": round(world * B * S * S / 1e6 / sec, 1), "unit": "megapixels/s", "ms_per_step": round(sec * 1e3, 4),
                        "roofline": hbm(wl.alg_bytes(lib), launch_ms * 1e-3, wl.kernel_label(), launch_ms, wl.launches_per_step)})
            del wl, ev
            torch.cuda.empty_cache()

    def sepconv_backward():
        """The op's gradient (kernel.cu:77-150,152-206: gradVertical + gradHorizontal, grad_input untouched) through the reference's
        entry point, on x3-replicated grayscale frames (what every training caller feeds: device-side channel comparison + dispatch
        inside the call) and on three independent channels.  Bytes = sstem_sepconv_backward_bytes (7,056.5 MB at C2);
        time = HIP events around every call inside the timed region."""
        import libs.sepconv._ext.cunnex as cunnex
        nbytes = int(lib.sstem_sepconv_backward_bytes(B, 3, S, S))
        for name, rgb in (("sepconv_backward_op_1024", False), ("sepconv_backward_op_rgb_1024", True)):
            g_ = torch.Generator(device=device); g_.manual_seed(777 + rank)
            inp = torch.rand(B, 3 if rgb else 1, S + 50, S + 50, device=device, generator=g_).expand(B, 3, S + 50, S + 50).contiguous()
            ver = torch.softmax(torch.randn(B, 51, S, S, device=device, generator=g_), 1)
            hor = torch.softmax(torch.randn(B, 51, S, S, device=device, generator=g_), 1)
            gout = torch.randn(B, 3, S, S, device=device, generator=g_)
            gv, gh = torch.empty_like(ver), torch.empty_like(hor)
            k = max(20, args.steps // 4)
            ev = ([torch.cuda.Event(enable_timing=True) for _ in range(k)], [torch.cuda.Event(enable_timing=True) for _ in range(k)])

            def call(i=None):
                if i is not None:
                    ev[0][i].record()
                cunnex.SeparableConvolution_cuda_backward(gout, inp, ver, hor, None, gv, gh)
                if i is not None:
                    ev[1][i].record()
            dt = timed(torch, dist, call, call, k, 3, 0.3)
            sec = max_over_ranks(torch, dist, dt, device, backend) / k
            launch_ms = sum(x.elapsed_time(y) for x, y in zip(*ev)) / k
            out.append({"name": name, "workload": "sepconv backward op (gV + gH), %s, %d x 3 x %d^2" % ("3 independent channels" if rgb else "gray x3", B, S),
                        "value": round(world * B * S * S / 1e6 / sec, 1), "unit": "megapixels/s", "ms_per_step": round(sec * 1e3, 4),
                        "roofline": hbm(nbytes, launch_ms * 1e-3, "rgb_gradh_stream + rgb_stream<1>" if rgb else "gray_gradv + gray_gradh", launch_ms, 2)})
            del inp, ver, hor, gout, gv, gh, ev
            torch.cuda.empty_cache()

    def ifnet_forward():
        fw = S_.IFNetForward(device, batch=B, size=S)
        sec = run(fw.step, k=10, w=2, prewarm=0.5)
        ms_fp32 = fp32_mfma_only(fw.step, k=3, w=1, prewarm=0.3)
        out.append({"name": "ifnet_forward", "workload": "SFF IFNet forward end to end on gray frame pairs, %d x %d^2 per GPU" % (B, S),
                    "value": round(world * B * S * S / 1e6 / sec, 2), "unit": "megapixels/s",
                    "ms_per_step": round(sec * 1e3, 3), "ms_fp32_mfma": ms_fp32,
                    "roofline": conv_roofline(fw.flop_per_step() / sec / 1e12, fw.flop_per_step(), inference_share=1.0)})
        del fw
        torch.cuda.empty_cache()

    def fusion_entry(global_batch, name, what, graph=None):
        # at 4 samples per GPU and below the step is ~290 launches in under 5 ms -- more than one Python thread issues in that time on
        # some hosts: forward + backward are replayed from a HIP graph there, as a small-batch rank would run it (if the capture fails
        # the entry runs eager and says so)
        if graph is None:
            graph = args.fusion_graph or global_batch // world <= 4
        # built (and its weights broadcast) ONCE; the capture has no collective in it and its outcome is agreed over the ranks inside
        # the step object (steps._TrainStep._finish_init): every rank replays or every rank runs eager
        st = S_.FusionStep(device, global_batch=global_batch, size=256, graph=graph)
        note = None
        if graph and not st.graphed:
            note = "graph capture failed (%s), every rank runs eager" % st.graph_error
            graph = False
        sec = run(st.step, k=max(10, args.steps), w=3, prewarm=0.7)
        ms_fp32 = fp32_mfma_only(st.step, k=10, w=2, prewarm=0.3) if not graph else None
        ms_x6 = bf16x6_train(st.step, k=10, w=2, prewarm=0.3) if not graph else None
        ar_ms = st.time_allreduce()
        flop, batch, loss_value = st.flop_per_step(), st.batch, float(st.loss.item())
        share = st.FLOW_FWD_FLOP_PER_SAMPLE / (st.FLOW_FWD_FLOP_PER_SAMPLE + 3 * st.UNET_FWD_FLOP_PER_SAMPLE)
        bucket_mb = round(st.bucket_bytes[0] / 1e6, 2)
        # the same step with the frozen flow net + back-warp of the NEXT batch on a second stream (steps.FusionStep(prefetch_flow=True):
        # the same launches per batch, the same weight trajectory bit for bit, tests/test_fullsize_gpu.py); always graph-replayed
        del st
        torch.cuda.empty_cache()
        ms_prefetch = None
        st = S_.FusionStep(device, global_batch=global_batch, size=256, graph=True, prefetch_flow=True)
        if st.graphed:
            ms_prefetch = round(run(st.step, k=max(10, args.steps), w=3, prewarm=0.3) * 1e3, 3)
        e = {"name": name, "workload": "SFF fusion step, %s" % (what % {"b": batch, "gb": global_batch, "w": world}),
             "value": round(global_batch / sec, 1), "unit": "samples/s", "ms_per_step": round(sec * 1e3, 3), "graph_replay": bool(graph),
             "ms_fp32_mfma": ms_fp32, "ms_bf16x6": ms_x6, "ms_flow_prefetch": ms_prefetch, "scaling": "strong", "loss": round(loss_value, 6),
             "roofline": conv_roofline(flop / sec / 1e12, flop, inference_share=share)}
        if world > 1:
            e.update({"allreduce_ms": round(ar_ms, 4), "grad_bucket_mb": bucket_mb, "collective": ("rccl" if backend == "nccl" else backend) + " all_reduce"})
        if note:
            e["note"] = note
        out.append(e)
        del st
        torch.cuda.empty_cache()

    def fusion_step():
        if args.fusion_batch % world:
            raise SystemExit("--fusion-batch %d does not split over %d ranks" % (args.fusion_batch, world))
        fusion_entry(args.fusion_batch, "fusion_training_step", "global batch %(gb)d x 256^2, %(w)d rank(s)")
        if world == 1 and args.fusion_batch == 16:
            # what ONE of 8 GPUs would run under the strong scaling north_star scores (>= 6x at 8 GPUs): 2 samples per GPU, no collective
            fusion_entry(2, "fusion_training_step_per_gpu_share_at_8_gpus",
                         "%(b)d x 256^2 (one GPU's share at 8 GPUs)", graph=True)

    def ifnet_step():
        """BASELINE config 5 (SFF interpolation training, 8 per GPU at 256x256, gradient all-reduce, Adam): with fp32 tensors under
        ALGO_AUTO, and under the opt-in bf16-operand convolution id the config names ("bf16 activations, fp32 sepconv accumulate")."""
        import hipnn.functional as HF
        for label, algo, dtype in (("ifnet_training_step", None, "f32"),
                                   ("ifnet_training_step_bf16_operands", HF.ALGO_MFMA_BF16, "bf16 conv operands, fp32 tensors")):
            prev = HF.get_algorithm()
            if algo is not None:
                HF.set_algorithm(algo)
            try:
                # the bf16 step is ~650 launches in ~6 ms: forward + backward replayed from a HIP graph (one Python thread is at its limit there)
                st = S_.IFNetStep(device, global_batch=8 * world, size=256, graph=algo is not None)
                sec = run(st.step, k=max(10, min(args.steps, 30)), w=3, prewarm=0.7)
                ms_x6 = bf16x6_train(st.step, k=10, w=2, prewarm=0.3) if (algo is None and not st.graphed) else None
                ar_ms = st.time_allreduce()
                tf = st.flop_per_step() / sec / 1e12
                if algo is not None:
                    roof = {"bound": "mfma", "conv": "bf16", "achieved": round(tf, 1), "peak": MFMA_BF16_PEAK_TF, "frac": round(tf / MFMA_BF16_PEAK_TF, 4)}
                else:
                    roof = conv_roofline(tf, st.flop_per_step())
                e = {"name": label, "workload": "SFF IFNet training step, 8 x 256^2 per GPU, %d rank(s)" % world,
                     "value": round(8 * world / sec, 1), "unit": "samples/s", "ms_per_step": round(sec * 1e3, 3), "graph_replay": bool(st.graphed),
                     "dtype": dtype, "loss": round(float(st.loss.item()), 6), "roofline": roof}
                if ms_x6 is not None:
                    e["ms_bf16x6"] = ms_x6
                if world > 1:
                    e.update({"allreduce_ms": round(ar_ms, 4), "grad_bucket_mb": round(st.bucket_bytes[0] / 1e6, 2)})
                out.append(e)
                del st
            finally:
                HF.set_algorithm(prev)
                torch.cuda.empty_cache()

    def sp_joint_step():
        """SURVEY 8-a13 "also report": the SP joint step (sp_scripts_train/main_fusion.py:178-257: IFNet x2, UNet x2, FusionNet x2, six
        L1 losses, one backward through all three nets -- the only step that runs the sepconv gradient kernels together with the
        U-Nets --, three bucket all-reduces started inside the backward pass, three Adams) at GLOBAL batch 16, 256x256, split over
        the ranks.  Convolution flops per sample at 256^2 (IFNet 286 G at 512^2 / 4 per pass, UNet and FusionNet 319 G at
        512^2 / 4 per pass, two passes each, x3 for forward + both gradients)."""
        gb = 16
        if gb % world:
            raise SystemExit("SP joint step: global batch 16 does not split over %d ranks" % world)
        st = S_.SPJointStep(device, global_batch=gb, size=256)
        sec = run(st.step, k=max(5, min(args.steps // 4, 10)), w=2, prewarm=0.5)
        ms_x6 = bf16x6_train(st.step, k=3, w=1, prewarm=0.2) if not st.graphed else None
        ar_ms = st.time_allreduce()
        # the same step with ONE evaluation of the interpolation net instead of the reference's two identical ones (both channels taken
        # from it: same losses, gradients up to the order of one addition per parameter -- steps.SPJointStep(single_vfi_pass=True))
        st1 = S_.SPJointStep(device, global_batch=gb, size=256, single_vfi_pass=True)
        ms_single = round(run(st1.step, k=max(5, min(args.steps // 4, 10)), w=2, prewarm=0.3) * 1e3, 2)
        del st1
        torch.cuda.empty_cache()
        flop = st.batch * 3.0 * 2.0 * (286e9 + 319e9 + 319e9) / 4.0
        e = {"name": "sp_joint_step", "workload": "SP joint step (3 nets x2, one backward), global batch %d x 256^2, %d rank(s)" % (gb, world),
             "value": round(gb / sec, 2), "unit": "samples/s", "ms_per_step": round(sec * 1e3, 2), "graph_replay": bool(st.graphed),
             "ms_single_interpolation_pass": ms_single, "ms_bf16x6": ms_x6, "scaling": "strong", "loss": round(float(st.loss.item()), 6),
             "roofline": conv_roofline(flop / sec / 1e12, flop)}
        if world > 1:
            e.update({"allreduce_ms": round(ar_ms, 4), "grad_bucket_mb": [round(b_ / 1e6, 1) for b_ in st.bucket_bytes]})
        if st.reducer is not None:
            e["allreduce_overlap"] = st.reducer.stats()      # says so when a pass fell back to blocking collectives
        out.append(e)
        del st
        torch.cuda.empty_cache()

    def sp_pipeline():
        """BASELINE config 4: the SP full pipeline (interp + correction + fusion, eval) on 2048x2048 tile sets, one tile set per rank
        and step (tile-sharded: independent units, no data-path collective).  25.0 TFLOP per tile set = ONE IFNet run (SURVEY 8a:
        4.58 + 2 x 5.11 + 2 x 5.11; the reference runs the identical IFNet twice, test_fusion.py:107-108)."""
        import dataparallel as DP_
        import sp_pipeline as SP
        S2 = 2048
        torch.manual_seed(555)
        models = SP.build_models(device)
        for m in models.values():
            DP_.broadcast_module(m)
        g = torch.Generator(device=device); g.manual_seed(555 + rank)
        im = [torch.rand(1, 1, S2, S2, device=device, generator=g) for _ in range(4)]
        mk = [(torch.rand(1, 1, S2, S2, device=device, generator=g) > 0.5).float() for _ in range(2)]
        ts = (im[0], im[1], mk[0], im[2], mk[1], im[3])

        def step():
            with torch.no_grad():
                return SP.restore_tile_set(models, *ts)
        sec = run(step, k=10, w=1, prewarm=0.3)
        ms_fp32 = fp32_mfma_only(step, k=2, w=1, prewarm=0.2)
        out.append({"name": "sp_pipeline_2048", "workload": "SP full pipeline (eval), one 2048^2 tile set per GPU, %d rank(s)" % world,
                    "value": round(2 * world * S2 * S2 / 1e6 / sec, 2), "unit": "restored megapixels/s",
                    "ms_per_step": round(sec * 1e3, 2), "ms_fp32_mfma": ms_fp32,
                    "roofline": conv_roofline(25.0e12 / sec / 1e12, 25.0e12, inference_share=1.0)})
        del models
        torch.cuda.empty_cache()

    for name, fn in (("apply256", apply256), ("apply_spellings", apply_spellings), ("sepconv_backward", sepconv_backward),
                     ("ifnet_forward", ifnet_forward), ("fusion_step", fusion_step), ("ifnet_step", ifnet_step),
                     ("sp_joint_step", sp_joint_step), ("sp_pipeline", sp_pipeline)):
        if name in which:
            guarded(name, fn)
    return out


def main():
    args = parse()
    if os.environ.get("SSTEM_BENCH_TRACE"):          # developer aid: dump every thread's Python stack after N seconds (hang hunting)
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["SSTEM_BENCH_TRACE"]), repeat=False, file=sys.stderr)
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        sys.exit(spawn_ranks(args))                       # nothing below runs in the parent
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(env_world or "1")
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))

    # HBM traffic of the headline launch from PMC counters, collected now: nothing in this process has touched the GPU yet
    measured_traffic = (None, "live passes switched off")
    if world == 1 and not args.no_live_traffic:
        if any(k.startswith(("ROCP_", "ROCPROF")) for k in os.environ):
            measured_traffic = (None, "this run is itself under a profiler")         # no profiler inside a profiler
        else:
            measured_traffic = live_traffic(args)

    import torch
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
        if dist.get_world_size() != args.gpus:
            raise SystemExit("process group has %d ranks, --gpus %d" % (dist.get_world_size(), args.gpus))
        # every rank must really be there (and RCCL up) before anything is timed
        probe = torch.ones(1, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(probe)
        if int(probe.item()) != args.gpus:
            raise SystemExit("all-reduce over the process group saw %d ranks, --gpus %d" % (int(probe.item()), args.gpus))

    import libs.sepconv._ext.cunnex as cunnex
    cunnex.set_algorithm(args.algo)
    lib = cunnex.load_library()

    B, S = args.batch, args.size
    wl = ApplyWorkload(args, B, S, device, rank)
    n_ev = wl.launches_per_step * args.steps
    ev = ([torch.cuda.Event(enable_timing=True) for _ in range(n_ev)], [torch.cuda.Event(enable_timing=True) for _ in range(n_ev)])

    with torch.no_grad():
        # Untimed pre-warm, before the W warm-up steps of the contract: an MI355X that has been idle needs a few hundred ms under load
        # to reach the clocks it then holds (measured: the same 20 timed steps 2.5 % slower with 5 warm-up steps = 7 ms than with
        # 500; steps of other benchmarks timed within 100 ms of idle ran 30-50 % slow).  Nothing of it is timed.
        dt = timed(torch, dist, wl.step, lambda k: wl.step(ev, k), args.steps, args.warmup, args.prewarm_s)
    dt = max_over_ranks(torch, dist, dt, device, backend)
    kern_ms = sum(a.elapsed_time(b) for a, b in zip(*ev)) / n_ev

    line = None
    if rank == 0:
        mp_per_step = world * B * S * S / 1e6
        alg_bytes = wl.alg_bytes(lib)
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        kname = wl.kernel_label()
        fused = not args.unfused
        # PMC traffic: measured by this run's own counter passes (live_traffic); else the recorded figure, only when it was collected for
        # THIS kernel on THIS workload
        traffic, traffic_source = measured_traffic
        if traffic is None:
            why_not = traffic_source
            traffic_source = None
        try:
            if traffic is not None:
                raise ValueError("measured live")
            with open(args.traffic_json) as f:
                tj = json.load(f)
            if tj.get("batch") == B and tj.get("size") == S and tj.get("fused", False) == fused \
                    and tj.get("rgb", False) == args.rgb and tj.get("kernel_label") == kname \
                    and tj.get("frame_planes", 3) == wl.planes:
                traffic = tj.get("hbm_bytes_per_launch")
                traffic_source = "NOT measured in this run (%s): %s (tools/pmc_traffic.py, separate rocprofv3 --pmc passes on the builder's box)" % (
                    why_not, os.path.relpath(args.traffic_json, REPO))
        except (OSError, ValueError):
            pass
        if args.unfused:
            spelling = "reference API, 2 op calls + add + mean"
        elif wl.planes == 1:
            spelling = "one fused launch on the two gray planes, " + ("row-segment coefficients as the IFNet heads store them" if wl.blocked else "NCHW coefficients")
        else:
            spelling = "one fused launch on x3-replicated frames, device-side dispatch in the span"
        data = "synthetic, " + ("3 independent channels" if args.rgb else ("grayscale pairs, one plane per frame" if wl.planes == 1 else "grayscale pairs x3"))
        line = {
            "metric": "restored megapixels/sec (interp+fusion fwd) at 1024x1024; PSNR vs ref",
            "value": round(mp_per_step * args.steps / dt, 3),
            "unit": "megapixels/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": data,
            "config": {"workload": "SepConv 51-tap interpolation forward (pad + 2 sepconv + add + mean; %s), batch=%d %dx%d tiles per GPU" % (spelling, B, S, S),
                       "batch_per_gpu": B, "tile": [S, S], "taps": 51, "frame_planes": wl.planes,
                       "coefficient_layout": "row-segments" if wl.blocked else "nchw",
                       "sharding": "independent tiles, no collective", "prewarm_s": args.prewarm_s,
                       # round-3 advisor: the headline reads the row-segment layout since round 3 -- rounds 1-2 timed NCHW tensors
                       "compare_with_rounds_1_2": "extra.apply_nchw_1024",
                       # what an `extra` entry does not repeat: f32 tensors and accumulation, weak scaling, HBM peak 8000 GB/s; mfma entries:
                       # TFLOP/s fp32-equivalent against `peak` = 2500 / MFMAs per product term (conv: f16x3 3, x6 6); ms_fp32_mfma = the
                       # entry with every layer on the fp32 matrix instruction
                       "extra_defaults": {"dtype": "f32", "scaling": "weak", "hbm_peak_gbs": HBM_PEAK_GBS, "mfma_unit": "TFLOP/s fp32-equivalent"}},
            "roofline": {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                         "bytes_per_launch": alg_bytes, "launch_ms": round(kern_ms, 4)},
        }
    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(S, args.rgb, wl.apply_first_of_batch)
    del wl, ev
    torch.cuda.empty_cache()

    if not args.no_metric_as_worded:
        # every rank runs the step (independent tiles, weak scaling); rank 0 reports
        try:
            maw = metric_as_worded(args, torch, dist, device, backend, rank, world, with_cpu=(world == 1 and not args.no_cpu_baseline))
        except Exception as exc:       # noqa: BLE001  (symmetric over the ranks: the same code on the same shapes)
            maw = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:200])}
            torch.cuda.empty_cache()
        if rank == 0:
            line["metric_as_worded"] = maw

    if not args.no_extra:
        which = set((args.extra_only or EXTRAS).split(","))
        # the headline line must come out even if an extra entry hangs (a collective that one rank never reaches, a capture that never
        # returns): after --extra-timeout seconds every rank leaves, rank 0 printing the line with what it has
        # Exactly ONE JSON line, whoever prints it: the lock is taken by the main thread before it prints and never released, and by
        # the watchdog before it prints and leaves; a hang reports itself in the exit code too (3), not only inside the JSON.
        import threading
        print_lock = threading.Lock()

        def bail():
            if not print_lock.acquire(blocking=False):
                return                                     # the main thread is already printing
            if rank == 0:
                line["extra"] = list(extras) + [{"name": "extras", "error": "the remaining extra entries did not finish within %d s" % args.extra_timeout}]
                print(fit_line(line), flush=True)
            os._exit(3)
        extras = []
        watchdog = threading.Timer(args.extra_timeout, bail)
        watchdog.daemon = True
        watchdog.start()
        try:
            run_extras(args, torch, dist, device, backend, rank, world, lib, which, extras)
        except Exception as exc:       # noqa: BLE001  (the headline line is printed whatever happens here)
            extras.append({"name": "extras", "error": "%s: %s" % (type(exc).__name__, str(exc)[:200])})
        if not print_lock.acquire(blocking=False):
            time.sleep(60)                                 # the watchdog fired a moment ago: it prints the line and ends the process
        watchdog.cancel()
        if rank == 0:
            line["extra"] = extras
    if rank == 0:
        print(fit_line(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
