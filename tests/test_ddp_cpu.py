"""world_size-2 gloo tests (CPU) of the data-parallel pieces: contiguous-chunk sharding with wrap padding
(pythia/datasets/samplers.py:42-60) and the bucketed, overlapped gradient all-reduce (vitxt_gqa_amd/ddp.py):
rank-averaged gradients must equal the single-process gradient of the mean loss over the global batch."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vitxt_gqa_amd.ddp import GradBuckets, shard_range


def test_shard_range_matches_distributed_sampler_chunks():
    assert shard_range(10, 0, 4) == [0, 1, 2] and shard_range(10, 3, 4) == [9, 0, 1]      # padded by wrapping
    assert sorted(sum((shard_range(64, r, 8) for r in range(8)), [])) == list(range(64))
    for n, w in ((7, 2), (512, 8), (5, 8)):
        per = (n + w - 1) // w
        assert all(len(shard_range(n, r, w)) == per for r in range(w))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model():
    torch.manual_seed(0)
    m = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.Tanh(), torch.nn.Linear(32, 32), torch.nn.Tanh(),
                            torch.nn.Linear(32, 4))
    m[2].bias.requires_grad_(False)            # a frozen ("dead") parameter must simply be skipped
    return m


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(1)
    x = torch.randn(8, 16)
    y = torch.randn(8, 4)
    m = _model()
    buckets = GradBuckets(m.parameters(), bucket_bytes=2048)       # several small buckets -> several collectives
    assert len(buckets.buckets) > 2
    idx = shard_range(8, rank, world)
    for step in range(2):                                          # second step checks reset()
        buckets.reset()
        loss = ((m(x[idx]) - y[idx]) ** 2).mean()
        loss.backward()
        buckets.finish()
    out[rank] = [p.grad.clone() for p in m.parameters() if p.requires_grad]
    dist.destroy_process_group()


def test_bucketed_allreduce_equals_global_batch_gradient():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    torch.manual_seed(1)
    x = torch.randn(8, 16)
    y = torch.randn(8, 4)
    m = _model()
    ((m(x) - y) ** 2).mean().backward()
    ref = [p.grad for p in m.parameters() if p.requires_grad]
    for r in range(world):
        assert len(out[r]) == len(ref)
        for a, b in zip(out[r], ref):
            assert torch.allclose(a, b, atol=1e-6), (a - b).abs().max()


def test_bucket_views_must_survive():
    m = _model()
    b = GradBuckets(m.parameters())
    m[0].weight.grad = None
    with pytest.raises(RuntimeError):
        b.reset()


def _worker_unused(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = _model()
    buckets = GradBuckets(m.named_parameters(), bucket_bytes=2048)
    buckets.reset()
    m[0](torch.randn(4, 16)).sum().backward()          # a partial backward: the later layers get no gradient
    try:
        buckets.finish()
        out[rank] = "no error"
    except RuntimeError as e:
        out[rank] = str(e)
    dist.destroy_process_group()


def test_finish_refuses_buckets_that_were_never_reduced():
    """A requires_grad parameter without a gradient leaves its bucket un-reduced: finish() must say so, not return."""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_unused, args=(world, _free_port(), out), nprocs=world, join=True)
    for r in range(world):
        assert "never all-reduced" in out[r] and "4.weight" in out[r], out[r]
