"""Training-harness pieces that sit directly on the hot path's step: the Adam update over
the flat parameter/gradient buffers, the reference's polynomial learning-rate schedule and its checkpoint
dictionary layout.  The data providers, loggers and validation loops of the reference stay out of scope.
"""
import gc
import math
import os

import torch

import sstem_native


def poly_lr(iters, base_lr, end_lr, warmup_iters, decay_iters, power):
    """``calculate_lr`` of sff_scripts_interp/main_ms.py:127-135 (same branches, same formula)."""
    if iters < warmup_iters:
        return (base_lr - end_lr) * pow(float(iters) / warmup_iters, power) + end_lr
    if iters < decay_iters:
        return (base_lr - end_lr) * pow(1 - float(iters - warmup_iters) / decay_iters, power) + end_lr
    return end_lr


class FlatParams:
    """Re-homes a module's trainable fp32 parameters as views into ONE contiguous buffer (values kept)."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev = self.params[0].device
        if any(p.dtype != torch.float32 or p.device != dev for p in self.params):
            raise ValueError("FlatParams needs fp32 parameters on one device")
        self.flat = torch.empty(sum(p.numel() for p in self.params), dtype=torch.float32, device=dev)
        off = 0
        with torch.no_grad():
            for p in self.params:
                n = p.numel()
                self.flat[off:off + n].copy_(p.data.reshape(-1))
                p.data = self.flat[off:off + n].view_as(p)
                off += n
        # whoever writes the flat buffer behind torch's back (FlatAdam's native launch) has to tell autograd: the version counters
        # of the parameters key the packed-weight and BatchNorm-fold caches of hipnn
        self.flat._sstem_param_views = self.params

    def mark_modified(self):
        for p in self.params:
            torch.autograd.graph.increment_version(p)


class FlatAdam:
    """torch.optim.Adam semantics (betas, eps, L2 weight decay, bias correction; no amsgrad) as ONE native launch
    over flat buffers: ``flat_param`` from ``FlatParams``, ``flat_grad`` from ``dataparallel.FlatGradBucket``."""

    def __init__(self, flat_param, flat_grad, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if flat_param.shape != flat_grad.shape or flat_param.dtype != torch.float32 or not flat_param.is_cuda:
            raise ValueError("FlatAdam needs matching fp32 GPU buffers")
        self.p, self.g = flat_param, flat_grad
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.exp_avg = torch.zeros_like(flat_param)
        self.exp_avg_sq = torch.zeros_like(flat_param)
        self.steps = 0

    def step(self, lr=None):
        if lr is not None:
            self.lr = lr
        self.steps += 1
        from hipnn import functional as _hf
        _hf.join_side_streams()        # weight-gradient launches still on hipnn's side stream add into self.g
        _hf.flush_deferred_wgrad()     # ... and so do reduce jobs left to the grouped launch (a no-op after a normal backward())
        lib = sstem_native.load_library()
        with torch.cuda.device(self.p.device):
            rc = lib.sstem_adam_step_f32(self.p.data_ptr(), self.g.data_ptr(), self.exp_avg.data_ptr(),
                                         self.exp_avg_sq.data_ptr(), self.p.numel(), float(self.lr), float(self.betas[0]),
                                         float(self.betas[1]), float(self.eps), float(self.weight_decay), self.steps,
                                         torch.cuda.current_stream().cuda_stream)
        sstem_native.check(rc, "sstem_adam_step_f32")
        # the launch wrote the parameters through a raw pointer: bump their version counters (caches keyed on them -- packed conv
        # weights, folded BatchNorm -- would otherwise serve the values from before the step)
        views = getattr(self.p, "_sstem_param_views", ())
        for q in views:
            torch.autograd.graph.increment_version(q)
        # ... and re-pack the 3x3 layers' MFMA weight layouts from the new values with ONE launch (hipnn keeps the pair workspaces
        # on the Parameters; without this every layer packs its own weights in the next forward)
        _hf.repack_after_update(views)

    def state_dict(self):
        return {"steps": self.steps, "lr": self.lr, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq}

    def load_state_dict(self, sd):
        self.steps, self.lr = int(sd["steps"]), float(sd["lr"])
        self.exp_avg.copy_(sd["exp_avg"]); self.exp_avg_sq.copy_(sd["exp_avg_sq"])


class L1MeanLoss:
    """``loss, grad = L1MeanLoss(device)(pred, target)``: nn.L1Loss()(pred, target) (the reference's criterion: sff_scripts_fusion/
    main_fusion.py:252, sff_scripts_interp/main_ms.py:205, sp_scripts_train/main_fusion.py:237-249) AND d loss / d pred, from ONE native
    launch (include/sstem_io.h, sstem_l1_mean_forward_grad_f32) -- the step then starts its backward pass at the network's output,
    ``pred.backward(grad)`` (or ``torch.autograd.backward([...], [...])`` for several losses whose sum is differentiated: the
    gradient of a sum hands every term a one), where torch's own pair is ~10 launches.  Same value up to the order of the fp32
    sum (fixed here, run to run), the same gradient bit for bit (sign(pred - target) / n, zero where they are equal).
    One object per call site: it owns the launch's small workspace (created here, outside any graph capture)."""

    def __init__(self, device):
        import sstem_native
        self._lib = sstem_native.load_library()
        self._check = sstem_native.check
        self._ws = torch.zeros(int(self._lib.sstem_l1_workspace_floats()), dtype=torch.float32, device=device)

    def __call__(self, pred, target):
        if not (pred.is_cuda and target.is_cuda):
            raise NotImplementedError("L1MeanLoss is GPU-only")
        if pred.dtype != torch.float32 or target.dtype != torch.float32 or pred.shape != target.shape:
            raise TypeError("L1MeanLoss: float32 tensors of one shape")
        p = pred.detach().contiguous()
        t = target.detach().contiguous()
        loss = torch.empty((), dtype=torch.float32, device=p.device)
        grad = torch.empty_like(p)
        with torch.cuda.device(p.device):
            rc = self._lib.sstem_l1_mean_forward_grad_f32(p.data_ptr(), t.data_ptr(), p.numel(), loss.data_ptr(), grad.data_ptr(),
                                                          self._ws.data_ptr(), torch.cuda.current_stream().cuda_stream)
        self._check(rc, "sstem_l1_mean_forward_grad_f32")
        return loss, grad.view(pred.shape)


class GraphedCallable:
    """``fn()`` -- a step body that works on persistent tensors (inputs, the flat gradient bucket, module buffers) --
    captured ONCE into a HIP graph and replayed by ``__call__``.

    The per-GPU share of a data-parallel step is small (BASELINE config 3 at 8 GPUs: 2 samples) and host-bound when its
    ~500 launches are issued one by one; a replay is one submission.  What stays OUTSIDE the graph by design: the gradient
    all-reduce (RCCL) and the optimiser launch (the caller runs them after the replay), so the captured body never
    changes the values its own packed-weight caches were built from.

    * warm-up: ``warmup`` eager runs on a side stream first (allocator, lazy attribute settings, packed-weight caches of
      frozen modules, the pair workspaces of the trained ones, the sepconv flag slots) -- nothing lazy is left to happen under capture.
      The warm-up runs leave NO trace in the training state (round-3 advisor finding): the running statistics and batch counters of
      every train-mode BatchNorm of ``modules`` and every tensor in ``preserve`` (state the body rotates or overwrites: a step's
      current / next batch buffers) are saved before the first warm-up run and put back after the last -- at construction (a
      checkpoint-resumed net keeps its statistics) and at every later re-capture (the batch that was handed over is still the one
      the next replay trains on);
    * tensors ``fn`` creates live in the graph's private pool and keep their addresses: whatever ``fn`` stores on its
      owner (e.g. ``self.loss``) stays readable after every replay;
    * train-mode BatchNorm launches update ``running_mean / running_var`` through raw pointers and tell autograd with
      ``increment_version`` on the host -- a replay does not run that host code, so it is repeated here for every
      BatchNorm buffer of ``modules`` (the eval-mode fold cache of hipnn is keyed on those counters).
    Replay equals the eager call bit for bit (same kernels, same order, same addresses): tests/test_steps_gpu.py.
    """

    def __init__(self, fn, modules=(), warmup=3, preserve=()):
        if not torch.cuda.is_available():
            raise NotImplementedError("GraphedCallable needs a GPU")
        self.fn = fn
        self.modules = list(modules)
        self.warmup = warmup
        self.preserve = list(preserve)
        self.captures = 0
        self.replays = 0
        self._capture()

    def _capture(self):
        from hipnn import functional as _hf
        # train-mode BatchNorm layers only: an eval-mode layer's buffers are not written by the body, and its fold cache (below) is
        # keyed on their version counters
        self._bn_buffers = []
        keep = list(self.preserve)
        for root in self.modules:
            for m in root.modules():
                if isinstance(m, torch.nn.modules.batchnorm._BatchNorm) and m.track_running_stats and m.training:
                    self._bn_buffers += [m.running_mean, m.running_var]
                    keep += [m.running_mean, m.running_var] + ([m.num_batches_tracked] if m.num_batches_tracked is not None else [])
        torch.cuda.synchronize()
        saved = [(t, t.detach().clone()) for t in keep]       # what the warm-up runs must not leave a trace in
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        # Weight packing stays OUTSIDE the graph: the layers find their pair workspaces (kept on the Parameters, hipnn.functional) packed by
        # the warm-up runs, the captured body launches no pack kernels, FlatAdam.step re-packs all of them with one launch after its
        # update, and __call__ re-packs, before replaying, any layer whose weights something else has changed since (version counter /
        # address).  (Recording every layer's own pack launch cost the bf16 IFNet step 46 launches per replay: 6.8 against 6.3 ms eager.)
        self._pack_params = [p for root in self.modules for p in root.parameters()]
        prev, prev_pin = _hf._pack_always, _hf._pin_slots
        _hf._pin_slots = True                                # the workspaces the captured launches read stay where they are
        _hf.capture_generation += 1                          # amax words handed out under this capture come from pools of its own
        try:
            with torch.cuda.stream(side):
                for _ in range(self.warmup):
                    self.fn()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            with torch.no_grad():
                for t, c in saved:
                    t.copy_(c)
            torch.cuda.synchronize()
            del saved
            self.graph = torch.cuda.CUDAGraph()
            # thread_local: helper threads of the process (RCCL's proxies, torch's process-group watchdog) keep calling the HIP
            # runtime while this thread captures; in the default "global" mode any such call invalidates the capture
            _hf._touch_log = touched = []
            # no cyclic garbage collection while the capture is open: a collection that happens to run inside it may destroy streams,
            # events or older graphs of the process (other step objects a caller dropped) -- runtime calls a capture does not survive
            # (seen once as "Fatal Python error: Aborted ... Garbage-collecting" under the test suite)
            gc_was_on = gc.isenabled()
            gc.collect()
            gc.disable()
            try:
                with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                    self.fn()
            finally:
                if gc_was_on:
                    gc.enable()
        finally:
            _hf._pack_always, _hf._pin_slots = prev, prev_pin
            _hf._touch_log = None
        # Frozen / inference modules keep their packed weights and their folded eval-mode BatchNorm on the MODULE (hipnn: _sstem_packs,
        # _sstem_fold), filled by the warm-up runs: the captured launches read those buffers, and a later change of the weights or the
        # running statistics (a checkpoint loaded after the capture) would make hipnn allocate NEW ones that the graph never sees.
        # Remember what the captured caches were built from (tensor, version counter, address); __call__ compares and captures again
        # when any of it has moved.
        seen, self._frozen = set(), []
        for group in touched:             # exactly the caches the captured body consulted (hipnn logs them while _touch_log is set)
            for t in group:
                if id(t) not in seen:
                    seen.add(id(t))
                    self._frozen.append((t, t._version, t.data_ptr()))
        self.captures += 1

    def frozen_caches_stale(self):
        return any(t._version != v or t.data_ptr() != a for t, v, a in self._frozen)

    def __call__(self):
        from hipnn import functional as _hf
        if self.frozen_caches_stale():
            # weights of a frozen module or statistics of an eval-mode BatchNorm changed after the capture: the eager warm-up runs of a
            # new capture rebuild the module-level caches, and the new graph reads the new buffers
            torch.cuda.synchronize()
            self.graph = None
            self._capture()
        _hf.refresh_stale_pack_slots(self._pack_params)
        self.graph.replay()
        self.replays += 1
        for b in self._bn_buffers:
            torch.autograd.graph.increment_version(b)


def save_checkpoint(model, iters, save_path, data_parallel_prefix=False, optimizer=None):
    """``{'current_iter', 'valid_result': None, 'model_weights'}`` as ``model-%06d.ckpt``
    (sff_scripts_interp/main_ms.py:282-285).  ``data_parallel_prefix`` reproduces the ``module.`` key prefix a
    DataParallel-wrapped reference model writes (and that inference_singleImage.py:42-47 strips)."""
    sd = model.state_dict()
    if data_parallel_prefix:
        sd = {"module." + k: v for k, v in sd.items()}
    states = {"current_iter": iters, "valid_result": None, "model_weights": sd}
    if optimizer is not None:
        states["optimizer_weights"] = optimizer.state_dict()      # SP scripts: main_interp.py:193-196
    path = os.path.join(save_path, "model-%06d.ckpt" % iters)
    torch.save(states, path)
    return path


def load_checkpoint(model, path, strip_module_prefix=None):
    """Loads 'model_weights'; strips a leading ``module.`` when present (or when told to, like the reference's
    unconditional k[7:])."""
    ckpt = torch.load(path, map_location="cpu")
    sd = ckpt["model_weights"]
    if strip_module_prefix is None:
        strip_module_prefix = all(k.startswith("module.") for k in sd)
    if strip_module_prefix:
        sd = {k[7:]: v for k, v in sd.items()}
    model.load_state_dict(sd)
    return ckpt.get("current_iter", 0)
