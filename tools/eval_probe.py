#!/usr/bin/env python3
"""Times eval-mode greedy decoding: prefix-reuse decoder vs the reference's recompute-everything loop.
usage: python tools/eval_probe.py [B F P]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vitxt_gqa_amd.synth import make_batch, make_noise  # noqa: E402
from vitxt_gqa_amd.testing import make_model, to_device  # noqa: E402

B, F, P = [int(a) for a in sys.argv[1:4]] if len(sys.argv) > 3 else (16, 100, 100)
dev = "cuda:0"
model = make_model(F, P, 5000, dtype=torch.bfloat16).to(dev).eval()
batch = to_device(make_batch(B, F, P, V=5000, seed=1, full_targets=False), dev)
batch.grounding_noise = tuple(t.to(dev) for t in make_noise(B, F, P, 1))
for cached in (True, False):
    model.decode_with_prefix_cache = cached
    with torch.no_grad():
        model.forward(batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = model.forward(batch)
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("B=%d %dx%d  %s: %.1f ms per batch = %.1f samples/s  (pos argmax checksum %d)"
          % (B, F, P, "prefix-cache decode" if cached else "reference loop     ", 1e3 * dt, B / dt,
             int(out["pos_scores"].argmax(-1).sum().item())))
