#!/usr/bin/env python3
"""torch.profiler view of one train step (B=64 default): device time per aten op and input shape, to attribute the small
fill / add / copy kernels of the rocprof kernel stats to the host code that issues them.
usage: python tools/op_profile.py [B F P]"""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vitxt_gqa_amd import training_config  # noqa: E402
from vitxt_gqa_amd.ddp import GradBuckets  # noqa: E402
from vitxt_gqa_amd.optim import build_optimizer, clip_and_step  # noqa: E402
from vitxt_gqa_amd.synth import make_batch, make_noise  # noqa: E402
from vitxt_gqa_amd.testing import make_model, to_device  # noqa: E402

a = sys.argv[1:]
B, F, P = (int(a[0]), int(a[1]), int(a[2])) if len(a) >= 3 else (64, 100, 100)
dev = torch.device("cuda", 0)
model = make_model(F, P, 5000, seed=0, dtype=torch.bfloat16, dropout=0.1).to(dev).train()
cfg = training_config()
opt = build_optimizer(model, cfg)
buckets = GradBuckets(model.parameters())
batch = to_device(make_batch(B, F, P, V=5000, seed=100), dev)
batch.grounding_noise = tuple(t.to(dev) for t in make_noise(B, F, P, seed=100))


def step():
    out = model(batch)
    loss = sum(l.mean() for l in out["losses"].values())
    buckets.reset()
    loss.backward()
    buckets.finish()
    clip_and_step(model, opt, cfg)


step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=70, max_name_column_width=40,
                                                         max_shapes_column_width=70))

# where the big copies / adds / fills come from: python stacks of the ops on [B, ~N, 768]-sized operands
big = {}
for e in prof.events():
    if e.name in ("aten::copy_", "aten::add_", "aten::add", "aten::fill_", "aten::cat", "aten::_to_copy", "aten::mul", "aten::tanh", "aten::sum", "aten::zero_",
                  "aten::index_select", "aten::index_add_", "aten::gather", "aten::where", "aten::tanh_backward", "aten::native_dropout", "aten::masked_fill_",
                  "aten::div_", "aten::mul_", "aten::clone", "aten::contiguous", "aten::zeros", "aten::zeros_like", "aten::sub", "aten::neg") and e.device_time_total > 100:
        st = [f for f in (e.stack or []) if ".py" in f and "torch/" not in f][:4]
        key = (e.name, str(e.input_shapes)[:60], " <- ".join(s.split("/")[-1] for s in st))
        t = big.setdefault(key, [0, 0.0])
        t[0] += 1
        t[1] += e.device_time_total
print("\nbig glue ops by call site (count, device ms):")
for k, (n, t) in sorted(big.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%3d %8.2f ms  %-12s %-60s %s" % (n, t / 1e3, k[0], k[1], k[2]))
