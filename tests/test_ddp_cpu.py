"""world_size-2 gloo tests (CPU) of the data-parallel pieces: contiguous-chunk sharding with wrap padding
(pythia/datasets/samplers.py:42-60) and the bucketed, overlapped gradient all-reduce (vitxt_gqa_amd/ddp.py):
rank-averaged gradients must equal the single-process gradient of the mean loss over the global batch."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vitxt_gqa_amd.ddp import DistributedSampler, GradBuckets, reduce_dict, shard_range


def test_shard_range_matches_distributed_sampler_chunks():
    assert shard_range(10, 0, 4) == [0, 1, 2] and shard_range(10, 3, 4) == [9, 0, 1]      # padded by wrapping
    assert sorted(sum((shard_range(64, r, 8) for r in range(8)), [])) == list(range(64))
    for n, w in ((7, 2), (512, 8), (5, 8)):
        per = (n + w - 1) // w
        assert all(len(shard_range(n, r, w)) == per for r in range(w))


def test_sampler_matches_torch_distributed_sampler_every_epoch():
    """The reference's sampler is ``torch.utils.data.distributed.DistributedSampler`` of its era, copied
    (pythia/datasets/samplers.py:1-3): epoch-seeded randperm, wrap padding, CONTIGUOUS chunk per rank.  Today's torch class
    strides the ranks (indices[rank::world]) instead, so the pin is the restated arithmetic of samplers.py:42-60 plus the
    properties both share: every rank sees the same permutation, the chunks tile it, set_epoch reshuffles reproducibly."""
    for n, w in ((10, 4), (64, 8), (7, 2), (5, 8), (513, 8)):
        for epoch in (0, 1, 13):
            g = torch.Generator()
            g.manual_seed(epoch)
            perm = torch.randperm(n, generator=g).tolist()
            per = (n + w - 1) // w
            padded = perm + perm[: per * w - n]
            chunks = []
            for r in range(w):
                sm = DistributedSampler(n, num_replicas=w, rank=r, shuffle=True)
                sm.set_epoch(epoch)
                got = list(iter(sm))
                assert got == padded[r * per:(r + 1) * per] and len(sm) == per
                chunks += got
            assert chunks == padded and set(chunks) == set(range(n))
        sm = DistributedSampler(list(range(n)), num_replicas=w, rank=w - 1, shuffle=False)
        assert list(iter(sm)) == shard_range(n, w - 1, w)
    a = DistributedSampler(100, 2, 0)
    e0 = list(a)
    a.set_epoch(1)
    e1 = list(a)
    a.set_epoch(0)
    assert e0 != e1 and list(a) == e0


def test_sampler_equals_the_reference_class():
    """Against the reference's own class where its checkout is present (this container; not on the GPU box)."""
    ref = "/root/reference/pythia/datasets/samplers.py"
    if not os.path.exists(ref):
        pytest.skip("reference checkout not present")
    import importlib.util
    spec = importlib.util.spec_from_file_location("_ref_samplers", ref)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for n, w, shuffle in ((10, 4, True), (64, 8, True), (7, 2, False), (513, 8, True)):
        for epoch in (0, 3):
            for r in range(w):
                a = mod.DistributedSampler(list(range(n)), num_replicas=w, rank=r, shuffle=shuffle)
                b = DistributedSampler(n, num_replicas=w, rank=r, shuffle=shuffle)
                a.set_epoch(epoch)
                b.set_epoch(epoch)
                assert list(a) == list(b) and len(a) == len(b)


def _reduce_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = {"train/vtextgqa/pos_bce_loss": torch.tensor([1.0 + rank]), "train/vtextgqa/InfoNCE": torch.tensor(10.0 * (rank + 1)),
         "a_metric": torch.tensor(0.25 * rank)}
    r = reduce_dict(d)
    out[rank] = {k: float(v) for k, v in r.items()}
    dist.destroy_process_group()


def test_reduce_dict_is_one_mean_reduce_to_rank_0():
    """distributed_utils.py:91-110: stacked in sorted-key order, ONE reduce, rank 0 divides by the world size."""
    world = 2
    out = mp.Manager().dict()
    mp.spawn(_reduce_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert out[0] == {"a_metric": 0.125, "train/vtextgqa/InfoNCE": 15.0, "train/vtextgqa/pos_bce_loss": 1.5}
    assert list(out[0]) == sorted(out[0])
    d = {"x": torch.tensor(2.0)}
    assert reduce_dict(d) is d                      # no process group: returned untouched (world_size < 2 branch)


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


# ------------------------------------------------------------------------------------------------------------------
# World size 8 (BASELINE.json configs[3]: 8 ranks, one per GPU) rehearsed over gloo on the CPU: the sampler, the buckets and
# the logging exchange with EIGHT ranks - bucket layout and launch order identical on every rank, finish() semantics, and
# the one documented deviation of the data-parallel mean (pos_bce_loss divides by the LOCAL mask count,
# pythia/modules/losses.py:341-342) shown as a number.
def _worker8(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    n_items = 29                                                   # not a multiple of 8: three ranks see wrapped items
    torch.manual_seed(1)
    x, y = torch.randn(n_items, 16), torch.randn(n_items, 4)
    m = _model()
    order = []
    buckets = GradBuckets(m.named_parameters(), bucket_bytes=2048)
    hooks = [p.register_post_accumulate_grad_hook(lambda p_, n=n: order.append(n)) for n, p in m.named_parameters() if p.requires_grad]
    sampler = DistributedSampler(n_items, num_replicas=world, rank=rank, shuffle=True)
    sampler.set_epoch(5)
    idx = torch.tensor(list(sampler))
    for step in range(2):
        buckets.reset()
        loss = ((m(x[idx]) - y[idx]) ** 2).mean()
        loss.backward()
        launched_before_finish = buckets.launched
        buckets.finish()
    logged = reduce_dict({"loss": loss.detach(), "rank": torch.tensor(float(rank))})
    # uneven loss masks: every rank holds 4 answers of 12 decoding steps; rank r masks out its last r steps of every answer.
    # The reference's pos_bce_loss = sum(masked losses) / max(sum(mask), 1) per RANK; the DDP mean of those is not the global mean
    mask = torch.ones(4, 12)
    if rank:
        mask[:, -rank:] = 0
    torch.manual_seed(100 + rank)
    per_elem = torch.rand(4, 12)
    local = (per_elem * mask).sum() / mask.sum().clamp(min=1)
    stats = torch.stack([local, (per_elem * mask).sum(), mask.sum()])
    gathered = [torch.zeros(3) for _ in range(world)]
    dist.all_gather(gathered, stats)
    out[rank] = dict(grads=[p.grad.clone() for p in m.parameters() if p.requires_grad], idx=idx.tolist(),
                     layout=[(flat.numel(), [tuple(p.shape) for p in ps]) for flat, ps in buckets.buckets], order=order,
                     launched=launched_before_finish, n_buckets=len(buckets.buckets),
                     logged={k: float(v) for k, v in logged.items()}, gathered=torch.stack(gathered))
    for h in hooks:
        h.remove()
    dist.destroy_process_group()


def test_world_size_8_sampler_buckets_and_logging_exchange():
    world, n_items = 8, 29
    out = mp.Manager().dict()
    mp.spawn(_worker8, args=(world, _free_port(), out), nprocs=world, join=True)
    res = [out[r] for r in range(world)]
    # sampler: the ranks' chunks tile the epoch's padded permutation (samplers.py:42-60)
    g = torch.Generator()
    g.manual_seed(5)
    perm = torch.randperm(n_items, generator=g).tolist()
    per = (n_items + world - 1) // world
    padded = perm + perm[: per * world - n_items]
    assert sum((r["idx"] for r in res), []) == padded and set(padded) == set(range(n_items))
    # buckets: identical layout and identical gradient-ready order on every rank (the collectives pair up by launch order),
    # every bucket launched from a hook BEFORE finish() (overlap with backward), twice (two steps)
    for r in res[1:]:
        assert r["layout"] == res[0]["layout"] and r["order"] == res[0]["order"]
    assert res[0]["n_buckets"] > 2 and all(r["launched"] == 2 * r["n_buckets"] for r in res)
    # gradients: the rank-average equals the single-process gradient of the mean over the PADDED global batch
    torch.manual_seed(1)
    x, y = torch.randn(n_items, 16), torch.randn(n_items, 4)
    m = _model()
    pi = torch.tensor(padded)
    ((m(x[pi]) - y[pi]) ** 2).mean().backward()
    ref = [p.grad for p in m.parameters() if p.requires_grad]
    for r in res:
        for a, b in zip(r["grads"], ref):
            assert torch.allclose(a, b, atol=1e-6), (a - b).abs().max()
    # logging exchange: rank 0 holds the mean over 8 ranks (distributed_utils.py:91-110; what a non-destination rank's buffer holds
    # after dist.reduce is backend-defined - as in the reference, only the main process logs)
    assert abs(res[0]["logged"]["rank"] - 3.5) < 1e-6 and list(res[0]["logged"]) == ["loss", "rank"]
    # the documented deviation, as a number: mean over ranks of (local masked mean) vs the global masked mean
    gt = res[0]["gathered"]
    ddp_mean = gt[:, 0].mean().item()
    global_mean = (gt[:, 1].sum() / gt[:, 2].sum()).item()
    assert gt[:, 2].tolist() == [48.0 - 4 * r for r in range(world)]
    assert abs(ddp_mean - global_mean) > 1e-4                      # they DO differ with uneven masks ...
    assert abs(ddp_mean - global_mean) < 0.05 * global_mean        # ... by a few per cent here (and by nothing for equal masks)
    print("pos_bce local-mean deviation at world 8, mask counts 48..20: ddp %.6f vs global %.6f (%.2f %%)"
          % (ddp_mean, global_mean, 100 * (ddp_mean / global_mean - 1)))


def _report_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    # rank 1 is the straggler: slower wall time, a longer exposed all-reduce wait, and a hand-off timeout in its status word
    local = {"elapsed_s": 2.0 + 0.5 * rank, "finish_wait_gpu_ms_per_step": 1.25 + 3.0 * rank, "finish_wait_host_ms_per_step": 0.5,
             "handoff_status": rank, "fused_launches": 22, "attn_bwd_ms_per_step": 240.0 + rank, "attn_fwd_ms_per_step": None,
             "peak_mem_gb": 210.0}
    rep = bench.gather_rank_report(local)
    out[rank] = rep
    dist.destroy_process_group()


def test_bench_rank_report_gathers_every_rank_over_gloo():
    """the N > 1 bench line must say WHY a run scaled badly - per-rank step times (min / max / rank of max), the exposed
    wait in GradBuckets.finish(), the hand-off status per rank.  bench.gather_rank_report over a two-rank gloo group: every rank gets
    the same report, lists indexed by rank, the slow rank named, the status words OR-ed."""
    world = 2
    out = mp.Manager().dict()
    mp.spawn(_report_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    r0, r1 = out[0], out[1]
    assert r0["per_rank"] == r1["per_rank"]
    assert r0["ranks"] == 2 and r0["per_rank"]["elapsed_s"] == [2.0, 2.5]
    assert r0["elapsed_min_s"] == 2.0 and r0["elapsed_max_s"] == 2.5 and r0["rank_of_max"] == 1
    assert abs(r0["spread_pct"] - 20.0) < 1e-9
    assert r0["per_rank"]["finish_wait_gpu_ms_per_step"] == [1.25, 4.25] and r0["exposed_allreduce_wait_ms_per_step_max"] == 4.25
    assert r0["per_rank"]["handoff_status"] == [0.0, 1.0] and r0["handoff_status_or"] == 1
    assert r0["per_rank"]["attn_bwd_ms_per_step"] == [240.0, 241.0]
    assert r0["per_rank"]["attn_fwd_ms_per_step"] == [None, None]                                # a field a rank did not report: null
    import json
    json.loads(json.dumps(r0), parse_constant=lambda c: (_ for _ in ()).throw(ValueError("non-finite constant %s in the line" % c)))
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    solo = bench.gather_rank_report({"elapsed_s": 1.0, "finish_wait_gpu_ms_per_step": 0.0, "handoff_status": 0})
    assert solo["ranks"] == 1 and solo["rank_of_max"] == 0 and solo["handoff_status_or"] == 0   # no process group: the single rank
    comm = bench.comm_environment()
    assert set(comm) == {"rccl_version", "env", "channels"} and "pinned" in comm["channels"]
