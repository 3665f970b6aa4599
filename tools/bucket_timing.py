#!/usr/bin/env python3
"""Where in the backward the gradient all-reduces are launched: one B = 64, 100 x 100 train step per measurement with ``GradBuckets`` on a
one-rank RCCL group (all a 1-GPU box allows: the library elides the one-rank collective kernels - profiles/r04_single_rank_rccl_overlap.json -
so the kernel timeline cannot show overlap).  What CAN be measured on one card is the launch POINT of every bucket on the compute stream:
a CUDA event recorded when the bucket's last gradient has been accumulated (the moment ``dist.all_reduce(async_op=True)`` is issued) against
events at the start and the end of backward.  The backward compute that remains behind a launch is the window its all-reduce has to hide in
(ring all-reduce of a 48 MB bucket over 7 xGMI links at ~50 GB/s effective per direction: ~2 ms).  usage: bucket_timing.py [steps]"""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29544")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    from vitxt_gqa_amd import training_config
    from vitxt_gqa_amd.ddp import GradBuckets
    from vitxt_gqa_amd.optim import build_optimizer, clip_and_step
    from vitxt_gqa_amd.synth import make_batch, make_noise
    from vitxt_gqa_amd.testing import make_model, to_device
    B, F, P, V = 64, 100, 100, 5000
    model = make_model(F, P, V, seed=0, dtype=torch.bfloat16, dropout=0.1).to("cuda").train()
    cfg = training_config()
    opt = build_optimizer(model, cfg)
    buckets = GradBuckets(model.named_parameters(), single_rank_collectives=True)
    batch = to_device(make_batch(B, F, P, V=V, seed=100), "cuda")
    batch.grounding_noise = tuple(t.cuda() for t in make_noise(B, F, P, seed=100))
    out_steps = []
    for it in range(steps):
        out = model(batch)
        loss = sum(l.mean() for l in out["losses"].values())
        buckets.reset()
        buckets.launch_events = []
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        loss.backward()
        e1.record()
        buckets.finish()
        clip_and_step(model, opt, cfg)
        torch.cuda.synchronize()
        total = e0.elapsed_time(e1)
        rec = [{"bucket": bi, "mb": buckets.buckets[bi][0].numel() * 4 / 2 ** 20, "launched_at_ms": e0.elapsed_time(ev),
                "backward_left_ms": ev.elapsed_time(e1)} for bi, ev in buckets.launch_events]
        out_steps.append({"backward_ms": total, "buckets": rec})
    last = out_steps[-1]
    hidden = sum(1 for r in last["buckets"] if r["backward_left_ms"] > 2.0)
    print(json.dumps({"config": "B=64, 100 x 100, bf16, dropout 0.1, one-rank RCCL group (launch points only)", "n_buckets": len(buckets.buckets),
                      "steps": out_steps[1:], "buckets_with_more_than_2ms_of_backward_behind_them": hidden}, indent=1))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
