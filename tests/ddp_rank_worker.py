"""One data-parallel rank of the T2S train step on a golden fixture (started as a FRESH process by tests/test_ddp_gpu.py;
never forked from a process that touched the GPU).  Each rank takes its contiguous shard of the fixture's questions
(DistributedSampler chunks, pythia/datasets/samplers.py:42-60), runs forward + both losses + backward through
``ddp.GradBuckets`` (the bucketed gradient all-reduce that replaces DistributedDataParallel(find_unused_parameters=True),
base_trainer.py:128-137) and rank 0 writes the rank-averaged gradients.

    RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment;
    argv: <fixture> <backend: gloo|nccl> <one_gpu: 0|1> <out.pt>
"""
import os
import sys

import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def main():
    case, backend, one_gpu, out_path = sys.argv[1], sys.argv[2], sys.argv[3] == "1", sys.argv[4]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = 0 if one_gpu else rank
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group(backend)
    from golden_util import Fixture
    from vitxt_gqa_amd.ddp import GradBuckets, shard_range
    from vitxt_gqa_amd.testing import build_model_for_fixture, to_device
    fx = Fixture(case)
    model = build_model_for_fixture(fx, torch.float32).to(dev).train()
    idx = torch.tensor(shard_range(fx.B, rank, world))
    s = to_device({k: v[idx] for k, v in fx.batch().items()}, dev)
    s.grounding_noise = (fx["E1"][idx], fx["E2"][idx])
    s.grounding_masks = {k: v[idx] for k, v in fx.masks().items()}
    # several buckets -> several collectives; a one-rank group (the RCCL rehearsal of a 1-GPU box) launches them all the same
    buckets = GradBuckets(model.named_parameters(), bucket_bytes=8 << 20, single_rank_collectives=True)
    for _ in range(2):                                                             # the second pass checks reset()
        buckets.reset()
        out = model(s)
        loss = sum(l.mean() for l in out["losses"].values())
        loss.backward()
        buckets.finish()
    lt = loss.detach().clone()
    dist.all_reduce(lt)
    if rank == 0:
        torch.save({"grads": {n: p.grad.detach().cpu() for n, p in model.named_parameters() if p.grad is not None},
                    "mean_loss": lt.item() / world, "n_buckets": len(buckets.buckets), "world": dist.get_world_size(),
                    "backend": dist.get_backend(), "collectives": buckets.launched}, out_path)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
