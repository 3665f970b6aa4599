"""Data parallelism for the T2S train step: one process per GPU, questions sharded across ranks, ONE
exchange step per iteration = gradient all-reduce (mean) over RCCL/xGMI, bucketed and overlapped
with backward.

Replaces the reference's ``nn.DataParallel`` default / opt-in ``DistributedDataParallel(...,
find_unused_parameters=True)`` (``pythia/trainers/base_trainer.py:51-71,121-137``): the 58 parameters
that never receive a gradient are frozen at build time, so no unused-parameter
scan is needed and every bucket's readiness is known statically.

Gradients live as views into a few large flat fp32 buckets (few, large collectives: a ring all-reduce
over point-to-point xGMI links is per-link bound, so per-collective latency is what to amortise).
A bucket is all-reduced (async, on RCCL's own stream) as soon as the last of its gradients has been
accumulated; ``finish()`` waits for the outstanding collectives before clipping / the optimizer.
Bucket size: the T2S model has ~84 M live parameters (334 MB of fp32 gradients at V=5000); with 256 MB
buckets the larger one only becomes ready when backward ends and its whole all-reduce is exposed; 48 MB
buckets give 7 collectives of which all but the last (the TextBert embedding table, whose gradient is the last one
backward produces) run under the remaining backward kernels.
"""
import torch
import torch.distributed as dist


def shard_range(n_items, rank, world_size):
    """Contiguous-chunk sharding of ``DistributedSampler`` (``pythia/datasets/samplers.py:42-60``):
    pad to a multiple of world_size by wrapping, rank r takes [r*n, (r+1)*n)."""
    per = (n_items + world_size - 1) // world_size
    idx = list(range(n_items))
    idx += idx[: per * world_size - n_items]
    return idx[rank * per:(rank + 1) * per]


class DistributedSampler:
    """The reference's sampler (``pythia/datasets/samplers.py:10-66``, wired by ``multi_dataset.py:285-304``): every epoch
    all ranks draw the SAME permutation of the dataset from a generator seeded with the epoch number (``shuffle=True``; the
    identity order otherwise), pad it to a multiple of the world size by wrapping around to its own head, and rank r takes the
    contiguous chunk [r n, (r + 1) n).  ``set_epoch`` (the trainer's ``seed_sampler``) re-seeds the next iteration.
    ``dataset`` may be anything with a length, or the length itself."""

    def __init__(self, dataset, num_replicas=None, rank=None, shuffle=True):
        if num_replicas is None:
            num_replicas = dist.get_world_size() if dist.is_initialized() else 1
        if rank is None:
            rank = dist.get_rank() if dist.is_initialized() else 0
        if not 0 <= rank < num_replicas:
            raise ValueError("rank %d outside a world of %d" % (rank, num_replicas))
        self.n = int(dataset) if isinstance(dataset, int) else len(dataset)
        self.num_replicas, self.rank, self.shuffle, self.epoch = num_replicas, rank, shuffle, 0
        self.num_samples = (self.n + num_replicas - 1) // num_replicas
        self.total_size = self.num_samples * num_replicas

    def indices(self):
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.epoch)
            order = torch.randperm(self.n, generator=g)
        else:
            order = torch.arange(self.n)
        order = torch.cat([order, order[: self.total_size - self.n]])
        return order[self.rank * self.num_samples:(self.rank + 1) * self.num_samples]

    def __iter__(self):
        return iter(self.indices().tolist())

    def __len__(self):
        return self.num_samples

    def set_epoch(self, epoch):
        self.epoch = int(epoch)


def reduce_dict(dictionary, group=None):
    """The scalar exchange of the logging path (``pythia/utils/distributed_utils.py:91-110``, called on the losses / metrics of
    a report): ONE reduce of the stacked values to rank 0, which divides by the world size; what the other ranks' returned values
    hold is whatever the backend's reduce leaves in a non-destination buffer (as in the reference, only the main process logs).
    Keys are sorted so that every rank stacks in the same order."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world < 2 or len(dictionary) == 0:
        return dictionary
    with torch.no_grad():
        keys = sorted(dictionary)
        values = torch.stack([dictionary[k].detach().reshape(()).float() for k in keys], dim=0)
        dist.reduce(values, dst=0, group=group)
        if dist.get_rank(group) == 0:
            values /= world
        return {k: v for k, v in zip(keys, values)}


class GradBuckets:
    def __init__(self, params, bucket_bytes=48 << 20, group=None, average=True, names=None, single_rank_collectives=False):
        """``params``: parameters, or (name, parameter) pairs as from ``named_parameters()`` (names only serve error messages).
        ``single_rank_collectives``: launch the bucket all-reduces even in a one-rank group (an identity, but the whole
        RCCL path - communicator, collective kernels on the bucket memory, stream hand-over - really runs: the rehearsal a
        1-GPU box allows)."""
        params = list(params)
        self._names = {}
        if params and isinstance(params[0], tuple):
            self._names = {id(p): n for n, p in params}
            params = [p for _, p in params]
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.collective = self.world > 1 or (single_rank_collectives and dist.is_initialized())
        self.launched = 0          # collectives launched since construction
        self.launch_events = None  # set to a list to record (bucket index, CUDA event on the compute stream) at every launch (tools/bucket_timing.py)
        self.average = average
        params = [p for p in params if p.requires_grad]
        # gradients become ready roughly in reverse construction order
        params = list(reversed(params))
        self.buckets = []          # (flat tensor, [params])
        cur, cur_n = [], 0
        cap = max(1, bucket_bytes // 4)
        for p in params:
            if cur and cur_n + p.numel() > cap:
                self._close(cur)
                cur, cur_n = [], 0
            cur.append(p)
            cur_n += p.numel()
        if cur:
            self._close(cur)
        self._pending = [0] * len(self.buckets)
        self._handles = []
        self._hooks = []
        for bi, (_, ps) in enumerate(self.buckets):
            for p in ps:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(bi)))
        self.reset()

    def _close(self, ps):
        flat = torch.zeros(sum(p.numel() for p in ps), dtype=torch.float32, device=ps[0].device)
        off = 0
        for p in ps:
            p.grad = flat[off:off + p.numel()].view_as(p)
            off += p.numel()
        self.buckets.append((flat, ps))

    def _make_hook(self, bi):
        def hook(_p):
            self._pending[bi] -= 1
            if self._pending[bi] == 0 and self.collective:
                flat = self.buckets[bi][0]
                if self.average and self.world > 1:
                    flat.div_(self.world)
                self.launched += 1
                if self.launch_events is not None and flat.is_cuda:
                    ev = torch.cuda.Event(enable_timing=True)
                    ev.record()
                    self.launch_events.append((bi, ev))
                self._handles.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        return hook

    def reset(self):
        """Zero the buckets (keeps .grad views alive; use instead of optimizer.zero_grad())."""
        for bi, (flat, ps) in enumerate(self.buckets):
            flat.zero_()
            self._pending[bi] = len(ps)
            for p in ps:                      # restore views if something replaced .grad
                if p.grad is None or p.grad.data_ptr() < flat.data_ptr() or p.grad.data_ptr() >= flat.data_ptr() + flat.numel() * 4:
                    raise RuntimeError("a bucketed .grad view was replaced; call GradBuckets.reset() instead of zero_grad(set_to_none=True)")
        self._handles = []

    def finish(self):
        """Wait for the outstanding collectives.  Every bucket must have been launched: a requires_grad parameter that got
        no gradient this iteration (a new unused parameter, a partial backward, an eval-style branch) would leave its bucket
        un-reduced and the ranks would step on different gradients - refuse instead of diverging silently (the reference
        pays for the same guarantee with ``find_unused_parameters=True``, base_trainer.py:134-137)."""
        if self.collective and any(self._pending):
            missing = [i for i, c in enumerate(self._pending) if c]
            names = []
            for bi in missing:
                names += [self._names.get(id(p), "<%s>" % "x".join(map(str, p.shape))) for p in self.buckets[bi][1]]
            for h in self._handles:
                h.wait()
            self._handles = []
            raise RuntimeError("GradBuckets.finish(): %d of %d gradient buckets were never all-reduced: %d parameter(s) of them "
                               "produced no gradient this iteration (candidates: %s)" % (len(missing), len(self.buckets),
                                                                                         sum(self._pending), ", ".join(names[:8])))
        # a hand-off wait that timed out on ONE rank puts NaN rows into the summed gradients of EVERY rank: the sticky status word
        # (ops.fused_status_tensor) is exchanged too (MAX), so that every rank's optimizer gates the step off and raises.  MAX because
        # word 1 counts launches; the two status bits travel as 0 / 1 flags of their own (words 2, 3: MAX = OR for flags), so a timeout on
        # one rank and a placement violation on another both arrive everywhere (ops.decode_status_words)
        flat0 = self.buckets[0][0] if self.buckets else None
        if self.collective and flat0 is not None and flat0.is_cuda and self.world > 1:
            from . import ops
            self._handles.append(dist.all_reduce(ops.fused_status_tensor(flat0.device), op=dist.ReduceOp.MAX, group=self.group, async_op=True))
        for h in self._handles:
            h.wait()
        self._handles = []

    def remove(self):
        for h in self._hooks:
            h.remove()
