"""world_size-2 gloo tests (CPU) of the multi-GPU host logic: weight broadcast, flat gradient bucket
all-reduce, tile sharding.  The GPU path uses the same code with the nccl (= RCCL) backend."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

import dataparallel as dp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _net(seed):
    torch.manual_seed(seed)
    return nn.Sequential(nn.Conv2d(2, 4, 3, padding=1), nn.BatchNorm2d(4), nn.ReLU(), nn.Conv2d(4, 1, 3, padding=1))


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, w, dev = dp.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and dev.type == "cpu"
    res = {}
    # 1. broadcast: ranks start from different seeds, end identical to rank 0
    net = _net(100 + rank)
    dp.broadcast_module(net)
    ref = _net(100)
    res["bcast_equal"] = all(torch.equal(a, b) for a, b in zip(net.state_dict().values(), ref.state_dict().values()))
    # 2. one data-parallel step == the same step on the concatenated batch in one process
    torch.manual_seed(7)
    full_x = torch.randn(4, 2, 6, 6); full_t = torch.randn(4, 1, 6, 6)
    mine = dp.shard_indices(4, rank, world)
    net.eval()                               # (train-mode BN statistics are per replica by design)
    bucket = dp.FlatGradBucket(net.parameters())
    opt = torch.optim.Adam(net.parameters(), lr=1e-2)
    bucket.zero()
    loss = nn.functional.l1_loss(net(full_x[mine]), full_t[mine])
    loss.backward()
    res["views_ok"] = bucket.check_views()
    bucket.allreduce_mean()
    opt.step()
    single = _net(100).eval()
    sopt = torch.optim.Adam(single.parameters(), lr=1e-2)
    sloss = nn.functional.l1_loss(single(full_x), full_t)
    sloss.backward()
    sopt.step()
    res["step_err"] = max((a - b).abs().max().item() for a, b in zip(net.state_dict().values(), single.state_dict().values()))
    res["bucket_bytes"] = bucket.nbytes
    # 3. max-over-ranks reduction as bench.py does for its timing
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    res["max"] = t.item()
    dp.barrier()
    out[rank] = res
    dp.shutdown()


def test_two_rank_gloo_broadcast_allreduce_and_step():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    for rank in range(world):
        r = out[rank]
        assert r["bcast_equal"]
        assert r["views_ok"]
        assert r["step_err"] < 1e-6          # mean of the two half-batch gradients == full-batch gradient
        assert r["max"] == 2.0
        assert r["bucket_bytes"] == 4 * sum(p.numel() for p in _net(0).parameters())


def test_shard_indices_cover_every_tile_once():
    for n in (0, 1, 7, 8, 64):
        for world in (1, 2, 4, 8):
            got = sorted(i for r in range(world) for i in dp.shard_indices(n, r, world))
            assert got == list(range(n))
    with pytest.raises(ValueError):
        dp.shard_indices(4, 2, 2)


def test_single_process_is_a_no_op():
    net = _net(1)
    dp.broadcast_module(net)
    b = dp.FlatGradBucket(net.parameters())
    b.allreduce_mean()
    assert dp.world_size() == 1 and b.check_views()
