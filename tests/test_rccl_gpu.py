"""RCCL readiness on a one-GPU box (round-2 verdict: "RCCL has never executed").  The pool's boxes have one GPU, so the only process
group RCCL can form here has ONE rank: the weight broadcast and the flat gradient all-reduce of ``dataparallel`` are forced through
it (``force=True``; with world_size 1 they are no-ops otherwise) on the real 86.6 MB SFF-IFNet bucket -- communicator creation, the
collectives' launches on torch's stream, stream ordering with the kernels around them and the results (sum over one rank, divided by
one: unchanged bit for bit) are the real thing; the xGMI transport is what a one-rank group cannot exercise.  Runs in a child
process: a process group in the pytest process would leak into the other tests."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

CHILD = r"""
import json, os, sys, time
sys.path.insert(0, os.path.join(%(repo)r, "sstem-restoration_amd")); sys.path.insert(0, %(repo)r)
os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="%(port)d")
import torch, torch.distributed as dist
import dataparallel as dp
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl" and dp.world_size() == 1
from model.model_interp import IFNet
torch.manual_seed(5)
net = IFNet(51).to(dev)
res = {}
before = [p.detach().clone() for p in net.parameters()]
v0 = [p._version for p in net.parameters()]
dp.broadcast_module(net, force=True)                       # one RCCL broadcast of the flat 86.6 MB parameter buffer
torch.cuda.synchronize()
res["broadcast_unchanged"] = all(torch.equal(a, b) for a, b in zip(before, net.parameters()))
res["versions_bumped"] = all(p._version > v for p, v in zip(net.parameters(), v0))
bucket = dp.FlatGradBucket(net.parameters())
g = torch.Generator(device=dev); g.manual_seed(6)
bucket.flat.copy_(torch.randn(bucket.flat.shape, device=dev, generator=g))
want = bucket.flat.clone()
# ordering with the kernels around it: a kernel that writes the bucket right before, one that reads it right after
bucket.flat.mul_(2.0)
bucket.allreduce_mean(force=True)                          # RCCL all-reduce(sum) + the 1/world scale
after = bucket.flat * 0.5
torch.cuda.synchronize()
res["allreduce_exact"] = bool(torch.equal(after, want))
res["bucket_mb"] = bucket.nbytes / 1e6
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    bucket.allreduce_mean(force=True)
e1.record(); torch.cuda.synchronize()
res["allreduce_ms_world1"] = e0.elapsed_time(e1) / 10
res["allreduce_repeat_exact"] = bool(torch.equal(bucket.flat, want * 2.0))
t = torch.tensor([3.0], device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX); res["max"] = t.item()
res["agree"] = dp.all_ranks_agree(True) and not dp.all_ranks_agree(False)
# the SP joint step's three buckets: all-reduces started inside the backward pass (dataparallel.OverlappedBuckets) == the three
# blocking calls behind it, through RCCL, bit for bit; the fusion net's bucket (at least) starts before the pass has ended
import steps
a = steps.SPJointStep(dev, global_batch=2, size=64, overlap=False)
b = steps.SPJointStep(dev, global_batch=2, size=64, overlap=True)
early = []
for it in range(3):
    a._fb()
    for bk in a.buckets:
        bk.allreduce_mean(force=True)
    b.reducer.begin(force=True); b._fb(); b.reducer.finish()
    early.append(b.reducer.fired_early)
    torch.cuda.synchronize()
    res.setdefault("overlap_equal", []).append(all(torch.equal(p.flat, q.flat) for p, q in zip(a.buckets, b.buckets)))
    for op in a.opts + b.opts:
        op.step()
res["overlap_early"] = early
res["overlap_expected"] = b.reducer.expected
dist.barrier(); dist.destroy_process_group()
print("RCCL_RESULT " + json.dumps(res))
"""


def test_rccl_world_of_one_broadcast_and_flat_bucket_allreduce(repo_root):
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", CHILD % {"repo": repo_root, "port": port}], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RCCL_RESULT ")][-1]
    res = json.loads(line[len("RCCL_RESULT "):])
    assert res["broadcast_unchanged"] and res["versions_bumped"]
    assert res["allreduce_exact"] and res["allreduce_repeat_exact"]
    assert res["max"] == 3.0 and res["agree"]
    assert 86.0 < res["bucket_mb"] < 87.5                    # 21,660,468 fp32 parameters
    assert all(res["overlap_equal"]) and res["overlap_early"][0] == 0 and all(e >= 1 for e in res["overlap_early"][1:]), res
    out_dir = os.path.join(repo_root, "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "rccl_world1.json"), "w") as f:
            json.dump(res, f)
