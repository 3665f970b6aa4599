#!/usr/bin/env python3
"""Loss / gradient-norm curve of the benchmark configuration on a FIXED synthetic batch (B=64, 100 frames x 100 OCR, bf16
operands, dropout 0.1, the reference's recipe: Adam 1e-4, warm-up factor 0.2 -> 1 over 1000 iterations, clip 0.25): evidence
that the shipped train step (fused five-product backward, shared-prefix passes, own clip + Adam) optimises at full size.
usage: python tools/train_curve.py [steps B]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vitxt_gqa_amd import training_config  # noqa: E402
from vitxt_gqa_amd.optim import build_optimizer, lr_lambda_update, train_step  # noqa: E402
from vitxt_gqa_amd.synth import make_batch, make_noise  # noqa: E402
from vitxt_gqa_amd.testing import make_model, to_device  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 16
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
F, P, V = 100, 100, 5000
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = make_model(F, P, V, seed=0, dtype=torch.bfloat16, dropout=0.1).to(dev).train()
cfg = training_config()
opt = build_optimizer(model, cfg)
sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda it: lr_lambda_update(it, cfg))
batch = to_device(make_batch(B, F, P, V=V, seed=100), dev)
batch.grounding_noise = tuple(t.to(dev) for t in make_noise(B, F, P, seed=100))
# every fused-backward launch ORs its status word (bit 0: a bounded hand-off spin timed out) into the device's sticky word and counts itself
from vitxt_gqa_amd import ops  # noqa: E402
launches = timeouts = 0
print("step  loss          grad_norm     lr            ms")
for i in range(steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loss, norm, _ = train_step(model, opt, sched, batch, cfg)
    torch.cuda.synchronize()
    print("%4d  %-12.4f  %-12.4f  %-12.3e  %.1f" % (i, loss.item(), norm.item(), opt.param_groups[0]["lr"], 1e3 * (time.perf_counter() - t0)), flush=True)
    assert torch.isfinite(loss) and torch.isfinite(norm)
    launches = ops.fused_launches_seen(dev)
    timeouts += ops.fused_handoff_status(dev) & 1
print("fused attention-backward launches: %d (dQ %s), hand-off spins that timed out: %d" % (launches, "hand-off" if (ops.ATTN_BWD_DQ_MODE & 0xff) == 1 else "atomics", timeouts))
