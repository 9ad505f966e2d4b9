"""One-process-per-GPU data parallelism for the restoration networks (RCCL over xGMI on MI355X).

What the reference does with a single-process ``nn.DataParallel`` (``sff_scripts_interp/main_ms.py:97-103``,
``sff_scripts_fusion/main_fusion.py:102-108``, ``sp_scripts_train/main_fusion.py:51-60``): every step it
re-broadcasts all parameters from GPU 0, scatters the batch, gathers outputs on GPU 0 and reduce-adds the
gradients there.  Here each rank owns its replica and its slice of the batch / its tiles:

* ``broadcast_module``      weights (and buffers) from rank 0 ONCE, as one flat bucket per dtype;
* ``FlatGradBucket``        all gradients live as views into one flat fp32 buffer -> ONE all-reduce per step
                            (86.6 MB for the SFF IFNet, 6.8 MB for the SFF UNet), then a scale by 1/world;
* ``shard_indices``         independent tiles / image pairs are dealt round-robin, no data-path collective;
* BatchNorm statistics stay per replica (DataParallel semantics); ``broadcast_module(..., buffers=True)``
  can re-align running stats from rank 0 when a checkpoint is written.

Backend: ``nccl`` (= RCCL) when CUDA/HIP devices are present, ``gloo`` otherwise (CPU tests).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torch.distributed.run sets them).
    Returns (rank, world_size, device).  A single process (WORLD_SIZE unset or 1) needs no group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_cuda = torch.cuda.is_available()
    device = torch.device("cuda", local_rank) if use_cuda else torch.device("cpu")
    if use_cuda:
        torch.cuda.set_device(device)
    if world > 1 and not dist.is_initialized():
        backend = backend or ("nccl" if use_cuda else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kwargs = {"device_id": device} if (use_cuda and backend == "nccl") else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kwargs)
    return rank, world, device


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_initialized() else 0


def shard_indices(n_items, rank, world):
    """Round-robin ownership of independent items (tiles, image pairs): rank r gets r, r+world, ..."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world of %d" % (rank, world))
    return list(range(rank, n_items, world))


def _gloo_with_gpu_tensor(t):
    # rehearsal of the multi-rank path on a one-GPU box (bench.py, SSTEM_BENCH_BACKEND=gloo): gloo moves host buffers
    return t.is_cuda and dist.get_backend() == "gloo"


def _all_reduce_sum(t):
    if _gloo_with_gpu_tensor(t):
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)


def _broadcast(t, src):
    if _gloo_with_gpu_tensor(t):
        h = t.cpu()
        dist.broadcast(h, src=src)
        t.copy_(h)
    else:
        dist.broadcast(t, src=src)


def _flat_groups(tensors):
    groups = {}
    for t in tensors:
        groups.setdefault((t.dtype, t.device), []).append(t)
    return groups


@torch.no_grad()
def broadcast_module(module, src=0, buffers=True, force=False):
    """Make every rank's parameters (and buffers) equal to rank ``src``'s: one broadcast per (dtype, device).
    force: run the collective even on a process group of ONE rank (the RCCL readiness test on a one-GPU box)."""
    if world_size() == 1 and not (force and dist.is_initialized()):
        return
    # the parameters / buffers themselves (not .data): copy_ then bumps their version counters, which key hipnn's packed-weight
    # and folded-BatchNorm caches -- a forward run before the broadcast must not leave stale packs behind on the non-source ranks
    tensors = list(module.parameters())
    if buffers:
        tensors += list(module.buffers())
    for (_, _), group in _flat_groups(tensors).items():
        flat = torch.cat([t.detach().reshape(-1) for t in group])
        _broadcast(flat, src)
        off = 0
        for t in group:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t))
            off += n


class FlatGradBucket:
    """All gradients of ``params`` as views into one contiguous buffer; ``allreduce_mean`` = one collective.

    The optimizer keeps working on ``p.grad`` (the views).  ``zero()`` replaces ``optimizer.zero_grad()``
    (which would drop the views when ``set_to_none`` is on)."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dtype, device = self.params[0].dtype, self.params[0].device
        for p in self.params:
            if p.dtype != dtype or p.device != device:
                raise ValueError("FlatGradBucket needs one dtype and one device")
        self.flat = torch.zeros(sum(p.numel() for p in self.params), dtype=dtype, device=device)
        off = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            # hipnn's native weight / bias / BatchNorm gradient launches may ADD into these views directly (accumulate flag of
            # the *_ex entry points) instead of handing autograd a tensor to add: one launch per parameter and step less
            p._sstem_grad_sink = True
            off += n

    @property
    def nbytes(self):
        return self.flat.numel() * self.flat.element_size()

    def zero(self):
        _join_side_streams(drop_deferred=True)   # a backward that raised may have left weight-gradient launches on the side stream un-joined
        self.flat.zero_()

    def check_views(self):
        """True while every p.grad is still a view of the flat buffer (autograd accumulates in place)."""
        base = self.flat.untyped_storage().data_ptr()
        return all(p.grad is not None and p.grad.untyped_storage().data_ptr() == base for p in self.params)

    def allreduce_mean(self, force=False):
        """force: run the collective even on a process group of ONE rank (the RCCL readiness test on a one-GPU box)."""
        _join_side_streams()       # weight-gradient launches that hipnn put on its side stream (normally joined at the end of backward())
        w = world_size()
        if w == 1 and not (force and dist.is_initialized()):
            return
        _all_reduce_sum(self.flat)
        self.flat.div_(w)


class OverlappedBuckets:
    """The gradient all-reduces of several buckets started DURING the backward pass: a bucket's collective is issued (asynchronously,
    on the process group's own stream) the moment the pass has delivered the last gradient of that bucket, while the networks further
    up the graph are still being differentiated -- the SP joint step's three buckets (92.5 + 69.1 + 69.1 MB,
    sp_scripts_train/main_fusion.py:51-60,255-257: the fusion net's gradients are complete before the correction and the
    interpolation nets are entered) instead of three blocking calls behind the whole pass.  Same collectives on the same values in
    the same order per bucket: bit-identical to ``FlatGradBucket.allreduce_mean`` (tests/test_dataparallel_cpu.py, world 2).

    How "the last gradient" is known: every parameter reports each delivery into its bucket -- autograd's own accumulation through a
    post-accumulate hook, hipnn's gradient sinks (native launches that add into the bucket directly) through ``_sstem_grad_notify``.
    The FIRST backward pass only counts (and reduces blocking at ``finish``, buckets in index order): how many deliveries a bucket
    receives per pass is a property of the step's graph (a gradient sink delivers once per launch, i.e. twice for a network called
    twice, autograd's own accumulation once per parameter and pass; dead parameters never do).  That pass also fixes the ORDER in which
    the buckets complete; counts and order are compared over the ranks once (``all_ranks_agree``) -- replicas that disagree never
    overlap (``mode`` says so), because ranks that issue the same collectives in different orders hang.

    From the second pass on a bucket fires when its count is reached AND every bucket before it in the learned order has fired, so
    every rank issues the collectives in one order whatever happens to its graph.  A graph that changes afterwards:
      * fewer deliveries than learned (a head gone dead, a parameter frozen): the bucket does not fire early; ``finish()`` fires what
        is left in the learned order -- correct, not overlapped -- and the next pass counts again (the order is kept);
      * MORE deliveries than learned (a parameter unfrozen, a network called once more): a delivery would land in a buffer whose
        collective is already reading and writing it.  That is detected at the delivery -- ``RuntimeError`` out of ``backward()``,
        the in-flight collectives are waited for, the counts are forgotten -- the step's gradients are invalid and the caller
        repeats the step (which counts again, blocking).  Round-3 advisor finding: this used to be silent.
        The change must be the SAME on every rank (the replicas run one program on equal shapes, so a parameter unfrozen or a network
        called once more is): the detecting rank has issued only part of the pass's collectives when it raises, and ranks that did
        not see the extra delivery issue all of them -- a rank-local change (a data-dependent branch that differs between replicas)
        leaves the process group out of step, and the repeated step hangs.  Such a step has no business under this class: use
        ``FlatGradBucket.allreduce_mean`` behind the pass (round-4 advisor finding: stated, not repaired).
    ``blocking_passes`` counts the passes after the first that could not start every collective early; ``stats()`` reports it.
    Usage per step:   reducer.begin(); loss.backward(); reducer.finish(); optimiser steps"""

    def __init__(self, buckets):
        self.buckets = list(buckets)
        n = len(self.buckets)
        self.expected = None                      # deliveries per bucket and pass, learned (totals; per parameter in _pexpected)
        self._pexpected = None
        self._pcounts = [dict() for _ in range(n)]
        self._remaining = [0] * n                 # parameters of the bucket that have not had all their learned deliveries yet
        self.order = None                         # the order in which the collectives are issued, learned once, never changed
        self.mode = "calibrating"                 # -> "overlapped" | "blocking: <why>"
        self._counts = [0] * n
        self._last = [0] * n                      # sequence number of each bucket's latest delivery (calibration: completion order)
        self._seq = 0
        self._fired = [False] * n
        self._work = [None] * n
        self._next = 0                            # position in `order` of the next bucket to fire
        self.fired_early = 0                      # collectives of the last pass that started before finish()
        self.passes = 0
        self.blocking_passes = 0
        self._active = False
        self._force = False
        for i, bk in enumerate(self.buckets):
            for p in bk.params:
                p.register_post_accumulate_grad_hook(self._hook(i))
                p._sstem_grad_notify = self._hook(i)

    def _hook(self, i):
        def delivered(p):
            if not self._active:
                return
            self._counts[i] += 1
            self._seq += 1
            self._last[i] = self._seq
            pc = self._pcounts[i]
            c = pc[id(p)] = pc.get(id(p), 0) + 1
            if self.expected is None:
                return                            # a counting pass: everything fires at finish()
            e = self._pexpected[i].get(id(p), 0)
            if self._fired[i] or c > e:           # counted PER PARAMETER: a head going dead while another goes live keeps the total
                self._abort(i)
            if c == e:
                self._remaining[i] -= 1
            self._fire_ready()
        return delivered

    def _fire_ready(self):
        while self._next < len(self.order):
            i = self.order[self._next]
            if self._remaining[i] != 0:
                return
            self._fire(i)
            self.fired_early += 1
            self._next += 1

    def _abort(self, i):
        """A delivery into a bucket that has (or should have) started its collective: this pass cannot be repaired."""
        self._active = False
        for w in self._work:
            if w is not None:
                w.wait()
        self._work = [None] * len(self.buckets)
        expected, self.expected, self._pexpected = self.expected, None, None
        raise RuntimeError("OverlappedBuckets: bucket %d received delivery %d of a pass it was learned to receive %d in -- the step's "
                           "graph changed (a parameter unfrozen, a network called once more) and the bucket's all-reduce had already "
                           "been started on incomplete gradients.  This step's gradients are invalid: zero the buckets and repeat the "
                           "step (the next pass counts again)." % (i, self._counts[i], expected[i]))

    def begin(self, force=False):
        """force: run the collectives even on a process group of one rank (RCCL readiness on a one-GPU box)."""
        self._force = force
        n = len(self.buckets)
        self._counts = [0] * n
        self._pcounts = [dict() for _ in range(n)]
        self._remaining = [len(d) for d in self._pexpected] if self._pexpected is not None else [0] * n
        self._last = [0] * n
        self._seq = 0
        self._fired = [False] * n
        self._work = [None] * n
        self._next = 0
        self.fired_early = 0
        self._active = True

    def _fire(self, i):
        self._fired[i] = True
        _join_side_streams()                      # weight-gradient launches still on hipnn's side stream add into this bucket
        if world_size() == 1 and not (self._force and dist.is_initialized()):
            return
        flat = self.buckets[i].flat
        if _gloo_with_gpu_tensor(flat):           # the one-GPU rehearsal: gloo moves host buffers, nothing to overlap
            _all_reduce_sum(flat)
        else:
            self._work[i] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)

    def _learn(self):
        """End of a counting pass: keep the counts; the first one also fixes the order and asks the other ranks."""
        self.expected = list(self._counts)
        self._pexpected = [dict(d) for d in self._pcounts]
        if self.order is not None:
            return
        n = len(self.buckets)
        self.order = sorted(range(n), key=lambda i: (self._last[i], i))
        self.mode = "overlapped"
        if world_size() > 1:
            # every rank must have learned the same counts and the same order; compared through rank 0's
            mine = torch.tensor(self.expected + self.order, dtype=torch.int64)
            ref = mine.clone()
            if dist.get_backend() == "nccl" and torch.cuda.is_available():
                ref = ref.cuda()
            dist.broadcast(ref, src=0)
            if not all_ranks_agree(bool((ref.cpu() == mine).all())):
                self.mode = "blocking: the ranks learned different delivery counts / orders"
                self.order = list(range(n))

    def finish(self):
        self._active = False
        counting = self.expected is None
        complete = (not counting) and all(r == 0 for r in self._remaining)
        # what has not fired: in index order on the very first pass (no order yet: every rank is in its first pass together), in the
        # learned order afterwards -- one order on every rank
        for i in (self.order if self.order is not None else range(len(self.buckets))):
            if not self._fired[i]:
                self._fire(i)
        if counting:
            self._learn()
            if self.mode.startswith("blocking"):
                self.expected = self._pexpected = None     # never fire early: every pass stays a counting pass
        elif not complete:
            self.expected = self._pexpected = None         # fewer deliveries than learned: count again next pass
        self.passes += 1
        if self.passes > 1 and self.fired_early < len(self.buckets):
            self.blocking_passes += 1
        w = world_size()
        for i, bk in enumerate(self.buckets):
            if self._work[i] is not None:
                self._work[i].wait()              # GPU: the current stream waits for the collective; CPU: blocks
            if w > 1:
                bk.flat.div_(w)

    def stats(self):
        return {"mode": self.mode, "passes": self.passes, "passes_not_fully_overlapped_after_the_first": self.blocking_passes,
                "fired_early_last_pass": self.fired_early, "buckets": len(self.buckets)}


def _join_side_streams(drop_deferred=False):
    """Weight-gradient work hipnn still has outstanding: launches on its side stream are joined, reduce jobs it deferred to one grouped
    launch (hipnn.functional.flush_deferred_wgrad, normally run by the autograd engine at the end of backward()) are issued -- or,
    with drop_deferred (the buckets are about to be zeroed: whatever a backward pass that raised left behind is void), forgotten."""
    import sys
    hf = sys.modules.get("hipnn.functional")
    if hf is not None:
        hf.join_side_streams()
        if drop_deferred:
            hf.drop_deferred_wgrad()
        else:
            hf.flush_deferred_wgrad()


def all_ranks_agree(ok):
    """True when `ok` is true on EVERY rank (one MIN all-reduce of a flag; every rank must call it)."""
    if world_size() == 1:
        return bool(ok)
    use_cuda = torch.cuda.is_available() and dist.get_backend() == "nccl"
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cuda" if use_cuda else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.item()))


def barrier():
    if dist.is_initialized():
        dist.barrier()


def shutdown():
    if dist.is_initialized():
        dist.destroy_process_group()
