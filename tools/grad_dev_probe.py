#!/usr/bin/env python3
"""Per-parameter gradient-norm deviation of the bf16 mode from the REFERENCE's fp32 gradients on a full-length fixture, next to the
reference's own deviation under torch.autocast(bfloat16) (tests/golden/bf16_floor.json).   python tools/grad_dev_probe.py [fixture]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_util import Fixture  # noqa: E402
from vitxt_gqa_amd import functional as FN  # noqa: E402
from vitxt_gqa_amd.testing import build_model_for_fixture, to_device  # noqa: E402

case = sys.argv[1] if len(sys.argv) > 1 else "full_peaky_b2_f100_p100"
fx = Fixture(case)
floor = json.load(open(os.path.join(ROOT, "tests", "golden", "bf16_floor.json")))[case]["autocast_bf16"]["grad"]["rel"]
model = build_model_for_fixture(fx, torch.bfloat16).to("cuda:0").train()
s = to_device(fx.batch(), "cuda:0")
s.grounding_noise = (fx["E1"], fx["E2"])
s.grounding_masks = fx.masks()
out = model(s)
loss = sum(v.mean() for v in out["losses"].values())
loss.backward()
params = dict(model.named_parameters())
total = fx["grad_total_norm"].item()
rows, sq = [], 0.0
for n, r in zip(fx.meta["grad_names"], fx["grad_norms"].tolist()):
    g = params[n].grad.double().norm().item()
    sq += g * g
    rows.append((abs(g - r) / (r + 1e-6 * total), n, g, r, floor[n]))
print("%s  FOLD_QSCALE=%s  loss %.4f (reference %.4f)  total norm %.2f vs %.2f (%.4f %%)" % (case, FN.FOLD_QSCALE, loss.item(), fx["loss_total"].item(), sq ** 0.5, total,
                                                                                         100 * abs(sq ** 0.5 - total) / total))
live = [x for x in rows if not x[1].endswith("attention.self.key.bias")]
print("parameters beyond 3 %%: %d (reference under autocast: %d); median deviation %.3f %% (autocast %.3f %%)" % (
    sum(x[0] >= 0.03 for x in live), sum(x[4] >= 0.03 for x in live), 100 * sorted(x[0] for x in live)[len(live) // 2], 100 * sorted(x[4] for x in live)[len(live) // 2]))
for rel, n, g, r, f in sorted(live, reverse=True)[:25]:
    print("  %7.3f %%  (autocast %7.3f %%)  %-70s %12.5g vs %12.5g" % (100 * rel, 100 * f, n, g, r))
