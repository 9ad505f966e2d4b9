"""The training / inference step shapes of the reference's callers, as objects the benchmarks,
the tests and a training script share.  Each reproduces the arithmetic of one reference loop body -- forward(s), L1,
backward, gradient all-reduce (one flat RCCL all-reduce where the reference's nn.DataParallel reduce-adds on GPU 0),
Adam -- on synthetic ``torch.rand`` data; data providers, logging and validation stay with the caller.

* ``FusionStep``      sff_scripts_fusion/main_fusion.py:213-259: frozen FusionNet flow -> back-warp of the SFF channels ->
                      UNet -> L1 -> backward -> all-reduce -> Adam(lr 1e-4)                       (BASELINE config 3)
* ``IFNetStep``       sff_scripts_interp/main_ms.py:187-211: IFNet -> L1 -> backward -> all-reduce -> Adam(lr 1e-3)
                                                                                                  (BASELINE config 5)
* ``SPJointStep``     sp_scripts_train/main_fusion.py:178-257: IFNet x2, UNet x2, FusionNet x2, six L1 losses, one
                      backward, three Adams
* ``IFNetForward``    the whole interpolation forward on grayscale frame pairs (inference_singleImage.py:55-70)

``global_batch`` is split evenly over the ranks of the process group (strong scaling, what DataParallel does with a
batch); BatchNorm statistics stay per replica (DataParallel semantics).
"""
import os

import torch

import dataparallel as dp
import train_utils

_l1 = torch.nn.functional.l1_loss
_NATIVE_L1 = os.environ.get("SSTEM_NATIVE_L1", "1") != "0"


def _local_batch(global_batch):
    w = dp.world_size()
    if global_batch % w:
        raise ValueError("global batch %d does not split over %d ranks" % (global_batch, w))
    return global_batch // w


class _TrainStep:
    """forward_backward() records the local gradient into the flat bucket(s); step() = that + all-reduce + Adam.
    ``graph=True`` replays forward_backward from a captured HIP graph (train_utils.GraphedCallable): the collective and the
    optimiser launch stay outside the graph."""

    def _finish_init(self, graph, overlap=None):
        """graph: replay forward_backward from a HIP graph.  The capture contains no collective and its outcome is AGREED over the
        ranks (one MIN all-reduce of a flag): a capture can fail on one rank only (a helper thread calling the runtime at the wrong
        moment), and a rank that then ran eager next to ranks that replay would still be correct -- but a caller that rebuilt the
        step on that rank alone would issue weight broadcasts nobody else enters.  Either every rank replays or every rank runs
        eager; ``graph_error`` says why not."""
        self._fb = self.forward_backward
        self.graphed = False
        self.graph_error = None
        if graph:
            fb = None
            try:
                fb = train_utils.GraphedCallable(self.forward_backward, modules=self.modules, preserve=self._graph_preserve())
            except Exception as exc:       # noqa: BLE001
                self.graph_error = "%s: %s" % (type(exc).__name__, str(exc)[:200])
                torch.cuda.synchronize()
            if dp.all_ranks_agree(fb is not None):
                self._fb, self.graphed = fb, True
            elif self.graph_error is None:
                self.graph_error = "capture failed on another rank"
        self.allreduce_ms = None
        # overlap: each bucket's all-reduce starts inside the backward pass, as soon as that bucket's gradients are complete
        # (dataparallel.OverlappedBuckets; bit-identical to the blocking calls).  Default: several ranks, several buckets, eager body
        # (a replayed graph runs no Python in its backward: nothing could be started early).
        if overlap is None:
            overlap = dp.world_size() > 1 and len(self.buckets) > 1 and not self.graphed
        self.reducer = dp.OverlappedBuckets(self.buckets) if overlap else None

    def _graph_preserve(self):
        """Tensors the step body rotates or overwrites that a capture's eager warm-up runs must leave as they found them."""
        return ()

    def _loss_backward(self, pred, target):
        """L1 criterion + the start of the backward pass: ONE native launch gives the loss and d loss / d pred
        (train_utils.L1MeanLoss), autograd starts at the network's output.  SSTEM_NATIVE_L1=0: torch's l1_loss + backward (A/B runs)."""
        if _NATIVE_L1:
            self.loss, g = self._l1(pred, target)
            pred.backward(g)
        else:
            self.loss = _l1(pred, target)
            self.loss.backward()

    def step(self):
        if self.reducer is not None:
            self.reducer.begin()
            self._fb()
            self.reducer.finish()
            for op in self.opts:
                op.step()
            return
        self._fb()
        for bk, op in zip(self.buckets, self.opts):
            bk.allreduce_mean()
            op.step()

    def time_allreduce(self, iters=10):
        """Mean wall time of the gradient all-reduce(s) of one step on this process group (ms); 0 on a single rank."""
        if dp.world_size() == 1:
            return 0.0
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        for bk in self.buckets:
            bk.allreduce_mean()
        torch.cuda.synchronize(); dp.barrier()
        e0.record()
        for _ in range(iters):
            for bk in self.buckets:
                bk.allreduce_mean()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters

    @property
    def bucket_bytes(self):
        return [bk.nbytes for bk in self.buckets]


class FusionStep(_TrainStep):
    # conv flops of one sample (SURVEY 8a a11 at 256x256): trained UNet forward 279.6 G / 16, frozen FusionNet 855.2 G / 16
    UNET_FWD_FLOP_PER_SAMPLE = 279.6e9 / 16
    FLOW_FWD_FLOP_PER_SAMPLE = 855.2e9 / 16

    def __init__(self, device, global_batch=16, size=256, lr=1e-4, seed=555, graph=False, flow=None, net=None, prefetch_flow=False):
        """flow / net: prebuilt modules (e.g. the pretrained flow predictor a training script loads, main_fusion.py:176-189) instead of
        the seeded random ones; they are moved to `device`, put in eval / train mode and broadcast from rank 0 like those.  Weights
        loaded into ``self.flow`` AFTER construction are picked up too: a captured graph notices and captures again (its warm-up runs
        leave the batch buffers and the trained net's BatchNorm statistics as they found them).  With ``prefetch_flow`` the flow of
        the batch in hand was computed during the previous step, so new flow weights take effect from the batch after it.

        prefetch_flow: the frozen flow predictor and the back-warp of the NEXT batch run on a second stream while the trained net does
        its forward / backward on the current one (whose warped input the previous call prepared) -- the flow net does not depend on
        anything the step updates, so every batch gets exactly the launches, in the order per batch, of the sequential step and the
        weights follow the same trajectory bit for bit (tests/test_steps_gpu.py); what changes is that the two chains, each a string
        of small launches at 2 samples per GPU, share the chip.  Protocol: ``prime(x, target)`` sets the first batch of a run (the constructor primes with its synthetic one),
        ``load_next(x, target)`` hands over the batch AFTER the one ``step()`` is about to train on."""
        from model.model_fusionnet import FusionNet
        from model.model_unet import UNet
        from utils.image_warp_torch import SpatialTransformation
        torch.manual_seed(seed)
        self.flow = (flow if flow is not None else FusionNet(6, 2, 32)).eval().to(device)
        self.net = (net if net is not None else UNet(6, 1)).train().to(device)
        dp.broadcast_module(self.flow); dp.broadcast_module(self.net)
        self.modules = [self.flow, self.net]
        self.flat = train_utils.FlatParams(self.net.parameters())
        self.buckets = [dp.FlatGradBucket(self.net.parameters())]
        self.opts = [train_utils.FlatAdam(self.flat.flat, self.buckets[0].flat, lr=lr, betas=(0.9, 0.999), eps=1e-8)]
        self.batch = _local_batch(global_batch)
        g = torch.Generator(device=device); g.manual_seed(seed + 1000 * dp.rank())
        self.x = torch.rand(self.batch, 6, size, size, device=device, generator=g)
        self.target = torch.rand(self.batch, 1, size, size, device=device, generator=g)
        self.inp = self.x.clone()          # the batch as the trained net sees it: channels 0-2 overwritten by the warped frames
        self.x3 = self.x[:, :3].contiguous()
        self.warp = SpatialTransformation(use_gpu=True)
        self.loss = None
        self._l1 = train_utils.L1MeanLoss(device)
        self.prefetch_flow = bool(prefetch_flow)
        if self.prefetch_flow:
            self.x_next, self.x3_next = self.x.clone(), self.x3.clone()
            self.target_next = self.target.clone()
            self.inp_next = self.x.clone()
            self._flow_stream = torch.cuda.Stream(device)
            self._flow_and_warp(self.x, self.x3, self.inp)            # the first batch: nothing to overlap it with
        self._finish_init(graph)

    def _graph_preserve(self):
        # prefetch_flow: every pass ends with "the next batch becomes the current one" -- three warm-up passes would drop the primed
        # batch and train the following one twice (round-3 advisor finding)
        return (self.inp, self.target, self.inp_next, self.target_next) if self.prefetch_flow else (self.inp,)

    def flop_per_step(self):
        """Convolution flops of one step on this rank: frozen flow forward + 3x the trained net's forward (fwd, dgrad, wgrad)."""
        return self.batch * (self.FLOW_FWD_FLOP_PER_SAMPLE + 3 * self.UNET_FWD_FLOP_PER_SAMPLE)

    def _flow_and_warp(self, x, x3, inp):
        with torch.no_grad():             # frozen flow predictor + back-warp of the SFF channels (main_fusion.py:227-235)
            pred_flow = self.flow(x)
            inp[:, :3] = self.warp(x3, pred_flow.permute(0, 2, 3, 1))                 # input[:, :3] = warped_sff (:235)

    def load(self, x, target):
        """The batch the next sequential ``step()`` trains on ([B,6,H,W] and [B,1,H,W], any device)."""
        if self.prefetch_flow:
            raise RuntimeError("load belongs to the sequential step; with prefetch_flow=True hand batches over with load_next")
        self.x.copy_(x); self.x3.copy_(self.x[:, :3]); self.target.copy_(target)
        self.inp[:, 3:] = self.x[:, 3:]

    def prime(self, x, target):
        """prefetch_flow: make (x, target) the batch the next ``step()`` trains on -- its flow and back-warp run here, not overlapped
        (the first batch of a run; the constructor primes with its synthetic batch)."""
        if not self.prefetch_flow:
            raise RuntimeError("prime belongs to prefetch_flow=True; a sequential step takes load(x, target)")
        self.x.copy_(x); self.x3.copy_(self.x[:, :3]); self.target.copy_(target)
        self.inp[:, 3:] = self.x[:, 3:]
        self._flow_and_warp(self.x, self.x3, self.inp)

    def load_next(self, x, target):
        """prefetch_flow: the batch after the one the next ``step()`` trains on ([B,6,H,W] and [B,1,H,W], any device)."""
        if not self.prefetch_flow:
            raise RuntimeError("load_next belongs to prefetch_flow=True; a sequential step reads self.x / self.target")
        self.x_next.copy_(x); self.x3_next.copy_(self.x_next[:, :3]); self.target_next.copy_(target)
        self.inp_next[:, 3:] = self.x_next[:, 3:]

    def forward_backward(self):
        if self.prefetch_flow:
            main, side = torch.cuda.current_stream(), self._flow_stream
            side.wait_stream(main)
            with torch.cuda.stream(side):
                self._flow_and_warp(self.x_next, self.x3_next, self.inp_next)
            self.buckets[0].zero()
            self._loss_backward(self.net(self.inp), self.target)
            main.wait_stream(side)
            self.inp.copy_(self.inp_next); self.target.copy_(self.target_next)       # the next batch becomes the current one
            return
        x = self.x
        with torch.no_grad():             # frozen flow predictor + back-warp of the SFF channels (main_fusion.py:227-235)
            pred_flow = self.flow(x)
            self.inp[:, :3] = self.warp(self.x3, pred_flow.permute(0, 2, 3, 1))      # input[:, :3] = warped_sff (:235)
        self.buckets[0].zero()
        self._loss_backward(self.net(self.inp), self.target)


class IFNetStep(_TrainStep):
    FWD_FLOP_PER_SAMPLE = 45.7e9        # SFF IFNet at 256x256

    def __init__(self, device, global_batch=8, size=256, lr=1e-3, seed=555, graph=False, net=None):
        from model.model_interp import IFNet
        torch.manual_seed(seed)
        self.net = (net if net is not None else IFNet(51)).train().to(device)
        dp.broadcast_module(self.net)
        self.modules = [self.net]
        self.flat = train_utils.FlatParams(self.net.parameters())
        self.buckets = [dp.FlatGradBucket(self.net.parameters())]
        self.opts = [train_utils.FlatAdam(self.flat.flat, self.buckets[0].flat, lr=lr, betas=(0.9, 0.999), eps=1e-8)]
        b = self.batch = _local_batch(global_batch)
        g = torch.Generator(device=device); g.manual_seed(seed + 1000 * dp.rank())
        f = torch.rand(b, 2, size, size, device=device, generator=g)      # two grayscale frames, each replicated x3
        self.x = torch.cat((f[:, :1].expand(b, 3, size, size), f[:, 1:].expand(b, 3, size, size)), 1).contiguous()
        self.target = torch.rand(b, 1, size, size, device=device, generator=g)
        self.loss = None
        self._l1 = train_utils.L1MeanLoss(device)
        self._finish_init(graph)

    def flop_per_step(self):
        return 3 * self.FWD_FLOP_PER_SAMPLE * self.batch * (self.x.shape[2] / 256.0) ** 2

    def forward_backward(self):
        self.buckets[0].zero()
        self._loss_backward(self.net(self.x), self.target)


class SPJointStep(_TrainStep):
    def __init__(self, device, global_batch=16, size=256, seed=555, graph=False, overlap=None, single_vfi_pass=False):
        """single_vfi_pass: the reference runs the interpolation net TWICE on the same input and takes channel 0 of the first pass and
        channel 1 of the second (sp_scripts_train/main_fusion.py:213-214) -- two evaluations of one function on one argument.  True
        evaluates it once and takes both channels: the same losses bit for bit, the same gradients up to the order of one addition per
        parameter (the two passes' contributions are summed by autograd there, here they arrive as one backward pass with both channels'
        gradients): tests/test_steps_gpu.py.  Default: the reference's literal dataflow."""
        self.single_vfi_pass = bool(single_vfi_pass)
        import networks
        torch.manual_seed(seed)
        self.vfi = networks.IFNet().train().to(device)
        self.den = networks.UNet(1, 1).train().to(device)
        self.fus = networks.FusionNet(1, 1).train().to(device)
        self.modules = [self.vfi, self.den, self.fus]
        self.buckets, self.opts, self.flats = [], [], []
        for m, lr in ((self.vfi, 1e-4 * 1e-20), (self.den, 1e-4 * 1e-6), (self.fus, 1e-4)):     # config/train_fusion.yaml:13,15 lr scales
            dp.broadcast_module(m)
            flat = train_utils.FlatParams(m.parameters())
            bk = dp.FlatGradBucket(m.parameters())
            self.flats.append(flat); self.buckets.append(bk)
            self.opts.append(train_utils.FlatAdam(flat.flat, bk.flat, lr=lr))
        b = self.batch = _local_batch(global_batch)
        g = torch.Generator(device=device); g.manual_seed(seed + 1000 * dp.rank())
        # img_1, img_2, img_2_degra, img_3, img_3_degra, img_4 and the two degradation masks
        self.im = [torch.rand(b, 1, size, size, device=device, generator=g) for _ in range(6)]
        self.mk = [(torch.rand(b, 1, size, size, device=device, generator=g) > 0.5).float() for _ in range(2)]
        self.loss = None
        self._l1s = [train_utils.L1MeanLoss(device) for _ in range(6)]
        self._finish_init(graph, overlap)

    def forward_backward(self):
        im, mk = self.im, self.mk
        for bk in self.buckets:
            bk.zero()
        inputs_vfi = torch.cat((im[0], im[0], im[0], im[5], im[5], im[5]), 1)
        if self.single_vfi_pass:
            both = self.vfi(inputs_vfi)
            vfi_pred1, vfi_pred2 = torch.unsqueeze(both[:, 0], 1), torch.unsqueeze(both[:, 1], 1)
        else:
            vfi_pred1 = torch.unsqueeze(self.vfi(inputs_vfi)[:, 0], 1)
            vfi_pred2 = torch.unsqueeze(self.vfi(inputs_vfi)[:, 1], 1)
        d1 = self.den(im[2]); d2 = self.den(im[4])
        pred1 = self.fus(vfi_pred1 * (1 - mk[0]), d1 * mk[0])
        pred2 = self.fus(vfi_pred2 * (1 - mk[1]), d2 * mk[1])
        if _NATIVE_L1:
            # six criteria, one launch each (loss + gradient); the backward pass starts at the six network outputs with those gradients:
            # the derivative of the reference's sum of six losses (main_fusion.py:237-249) hands every term a one
            outs = (vfi_pred1, d1, pred1, vfi_pred2, d2, pred2)
            tgts = (im[1], im[1], im[1], im[3], im[3], im[3])
            ls, gs = zip(*(l1(o, t) for l1, o, t in zip(self._l1s, outs, tgts)))
            self.loss = (ls[0] + ls[1] + ls[2]) + (ls[3] + ls[4] + ls[5])
            torch.autograd.backward(outs, gs)
            return
        self.loss = (_l1(vfi_pred1, im[1]) + _l1(d1, im[1]) + _l1(pred1, im[1])) + (_l1(vfi_pred2, im[3]) + _l1(d2, im[3]) + _l1(pred2, im[3]))
        self.loss.backward()


class IFNetForward:
    """Whole SFF interpolation forward on grayscale frame pairs (per rank: `batch` pairs; independent tiles, no collective)."""
    FWD_FLOP_PER_SAMPLE_256 = 45.7e9

    def __init__(self, device, batch=8, size=1024, seed=555):
        from model.model_interp import IFNet
        torch.manual_seed(seed)
        self.net = IFNet(51).eval().to(device)
        dp.broadcast_module(self.net)
        g = torch.Generator(device=device); g.manual_seed(seed + 1000 * dp.rank())
        self.f1 = torch.rand(batch, 1, size, size, device=device, generator=g)
        self.f2 = torch.rand(batch, 1, size, size, device=device, generator=g)
        self.batch, self.size = batch, size

    def flop_per_step(self):
        return self.FWD_FLOP_PER_SAMPLE_256 * self.batch * (self.size / 256.0) ** 2

    @torch.no_grad()
    def step(self):
        return self.net.interpolate_gray(self.f1, self.f2)


class SFFRestoreForward:
    """The headline metric's literal wording, "interp + fusion fwd": the whole SFF restoration forward on a batch of tiles
    (sff_pipeline.restore_sff: IFNet on the two neighbouring sections -> unfolding-flow FusionNet -> back-warp -> fusion UNet, all
    eval; sff_scripts_interp/inference_singleImage.py:55-71 + sff_scripts_fusion/inference.py:126-153).  Per rank: `batch`
    independent tiles, no collective."""
    IFNET_FLOP_256 = 45.7e9
    FLOW_FLOP_256 = 855.2e9 / 16            # a11: frozen FusionNet, per 256 x 256 sample
    UNET_FLOP_256 = 279.6e9 / 16            # a11: fusion UNet forward, per 256 x 256 sample

    def __init__(self, device, batch=8, size=1024, seed=555):
        import sff_pipeline
        torch.manual_seed(seed)
        self.models = sff_pipeline.build_models(device)
        for m in self.models.values():
            dp.broadcast_module(m)
        g = torch.Generator(device=device); g.manual_seed(seed + 1000 * dp.rank())
        self.prev, self.nxt, self.sff = (torch.rand(batch, 1, size, size, device=device, generator=g) for _ in range(3))
        self.batch, self.size = batch, size
        self._restore = sff_pipeline.restore_sff

    def flop_per_step(self):
        return (self.IFNET_FLOP_256 + self.FLOW_FLOP_256 + self.UNET_FLOP_256) * self.batch * (self.size / 256.0) ** 2

    @torch.no_grad()
    def step(self):
        return self._restore(self.models, self.prev, self.nxt, self.sff)[0]
