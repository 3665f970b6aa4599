#!/usr/bin/env python3
"""ELEMENT-WISE distance of the bf16 mode's gradients from this build's own fp32 parity mode (which matches the reference's fp32 gradients to
1e-5 at this length: tests/test_fulllength_reference_gpu.py) on a full-length fixture - every parameter, every element, not norms: the global
relative L2 distance of the whole gradient vector, its cosine, and the per-parameter relative L2 distances.  Run once per setting of
T2S_FOLD_QSCALE (read at import).   python tools/grad_vector_probe.py [fixture]"""
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


def grads(dtype):
    model = build_model_for_fixture(fx, dtype).to("cuda:0").train()
    s = to_device(fx.batch(), "cuda:0")
    s.grounding_noise = (fx["E1"], fx["E2"])
    s.grounding_masks = fx.masks()
    out = model(s)
    loss = sum(v.mean() for v in out["losses"].values())
    loss.backward()
    g = {n: p.grad.detach().double().cpu() for n, p in model.named_parameters() if p.grad is not None}
    del model, out, loss
    torch.cuda.empty_cache()
    return g


ref = grads(torch.float32)
got = grads(torch.bfloat16)
num = sum(((got[n] - ref[n]) ** 2).sum() for n in ref)
den = sum((ref[n] ** 2).sum() for n in ref)
dot = sum((got[n] * ref[n]).sum() for n in ref)
gn = sum((got[n] ** 2).sum() for n in ref)
print("%s  FOLD_QSCALE=%s: whole gradient vector, bf16 mode vs this build's fp32 mode: relative L2 distance %.4f, cosine %.6f, norm ratio %.5f" % (
    case, FN.FOLD_QSCALE, (num / den).sqrt().item(), (dot / (den.sqrt() * gn.sqrt())).item(), (gn / den).sqrt().item()))
rows = sorted(((((got[n] - ref[n]).norm() / (ref[n].norm() + 1e-9 * den.sqrt())).item(), n, ref[n].norm().item()) for n in ref
               if not n.endswith("attention.self.key.bias")), reverse=True)
import statistics  # noqa: E402
rel = [r[0] for r in rows]
print("   per-parameter relative L2 distance: median %.4f, mean %.4f, max %.4f; parameters beyond 0.10: %d, beyond 0.20: %d of %d" % (
    statistics.median(rel), sum(rel) / len(rel), rel[0], sum(r > 0.10 for r in rel), sum(r > 0.20 for r in rel), len(rel)))
for r, n, rn in rows[:12]:
    print("   %7.4f  %-70s |g| %.5g" % (r, n, rn))
for key in ("mmt.encoder.layer.0.attention.self.key.weight", "mmt.encoder.layer.0.attention.self.query.weight", "mmt.encoder.layer.0.attention.self.value.weight",
            "TransLayer.encoder.layer.0.attention.self.key.weight", "text_bert.encoder.layer.0.attention.self.query.weight"):
    for r, n, rn in rows:
        if n == key:
            print("   [%s] %.4f" % (n, r))
