#!/usr/bin/env python3
"""Error budget of the bf16 operand mode against the REFERENCE's fp32 logits on a reference-generated fixture:
which stage owns the logit error?  The teacher-forced training forward (masks and noise injected, dropout 0) is run several times;
each run moves ONE more stage of the product's bf16 path to fp32 arithmetic, cumulatively:

    bf16            the product as it ships (bf16 MFMA operands, fp32 accumulate / residual stream / logits)
    +softmax32      the attention kernel in exact fp32 on the SAME bf16-rounded Q / K / V values: removes the second rounding of
                    Q (x scale log2 e), the bf16 rounding of P ahead of P.V, exp2 instead of exp
    +qkv32          additionally the fused QKV projection with an fp32 OUTPUT (bf16 inputs): Q / K / V themselves are not rounded
    +ptr32          additionally the pointer head's keys in fp32 and its dot products exact (t2s_ptr_scores exact_fp32)
    fp32            the parity mode (every operand fp32): what is left is the fp32 kernels' own error

Reported per run: max |logit - reference| over the vocabulary logits and over the pointer logits of the three passes, the worst
decoder-output (mmt_dec) deviation, and the argmax flips.  A stage owns what its switch removes.

    python tools/error_budget.py [fixture ...]          (default: both full-length fixtures)
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from golden_util import Fixture  # noqa: E402
from vitxt_gqa_amd import functional as FN  # noqa: E402
from vitxt_gqa_amd import ops  # noqa: E402
from vitxt_gqa_amd.testing import build_model_for_fixture, to_device  # noqa: E402

DEV = "cuda:0"
F32, BF16 = torch.float32, torch.bfloat16


class Switches:
    """Monkey-patches of the product's host functions for ONE forward (restored on exit)."""

    def __init__(self, softmax32=False, qkv32=False, ptr32=False):
        self.softmax32, self.qkv32, self.ptr32 = softmax32, qkv32, ptr32

    def __enter__(self):
        self.saved = (ops.attn_fwd, FN._mm_bias, ops.ptr_scores)
        attn_fwd, mm_bias, ptr_scores = self.saved
        sw = self

        def attn_fwd32(qkv, keys, *a, kv=None, **k):
            if not sw.softmax32:
                return attn_fwd(qkv, keys, *a, kv=kv, **k)
            out, lse = attn_fwd(qkv.float(), keys, *a, kv=(kv.float() if kv is not None else None), **k)      # the exact-fp32 kernel
            return out.to(BF16), lse

        def mm_bias32(x2, w, b):
            if sw.qkv32 and x2.dtype == BF16 and w.shape[0] == 3 * FN.HID:
                return torch.addmm(b.float(), x2.float(), w.float().t())          # fp32 output: Q / K / V keep their fp32 values
            return mm_bias(x2, w, b)

        def ptr_scores32(q, k, mask01, out, col0, exact_fp32=False):
            if sw.ptr32:
                return ptr_scores(q, k.float().contiguous(), mask01, out, col0, exact_fp32=True)
            return ptr_scores(q, k, mask01, out, col0, exact_fp32=exact_fp32)

        ops.attn_fwd, FN._mm_bias, ops.ptr_scores = attn_fwd32, mm_bias32, ptr_scores32
        return self

    def __exit__(self, *a):
        ops.attn_fwd, FN._mm_bias, ops.ptr_scores = self.saved


def run(fx, dtype, **sw):
    model = build_model_for_fixture(fx, dtype).to(DEV).train()
    model.keep_intermediates = True
    s = to_device(fx.batch(), DEV)
    s.grounding_noise = (fx["E1"], fx["E2"])
    s.grounding_masks = fx.masks()
    prune = FN.PRUNE_KV_MAX_KEYS
    if sw.get("qkv32"):
        FN.PRUNE_KV_MAX_KEYS = -1          # (the pruned K | V projection is a second call site of the same GEMM: keep one form)
    try:
        with torch.no_grad(), Switches(**sw):
            out = model.forward(s)
    finally:
        FN.PRUNE_KV_MAX_KEYS = prune
    res = {}
    V = fx.V
    for k in ("ref_scores", "pos_scores", "neg_scores"):
        err = (out[k].float().cpu() - fx[k]).abs()
        res[k] = (err[..., :V].max().item(), err[..., V:].max().item(), int((out[k].float().cpu().argmax(-1) != fx[k].argmax(-1)).sum()),
                  err.pow(2).mean().sqrt().item())
    f = model._last_fwd
    res["mmt_dec"] = max((f[p + "_mmt_dec"].float().cpu() - fx[p + "_mmt_dec"]).abs().max().item() for p in ("ref", "pos", "neg"))
    res["qtv_ocr"] = (f["ocr_mmt_in"].float().cpu()[:, ::fx.meta["row_stride"]] - fx["ocr_in"]).abs().max().item()
    del model
    torch.cuda.empty_cache()
    return res


def main():
    cases = sys.argv[1:] or ["full_b1_f100_p100", "full_peaky_b2_f100_p100"]
    stages = [("bf16 (product)", BF16, {}),
              ("+softmax32", BF16, dict(softmax32=True)),
              ("+qkv32", BF16, dict(softmax32=True, qkv32=True)),
              ("+ptr32", BF16, dict(softmax32=True, qkv32=True, ptr32=True)),
              ("ptr32 only", BF16, dict(ptr32=True)),
              ("fp32 (parity mode)", F32, {})]
    for case in cases:
        fx = Fixture(case)
        st = fx.meta.get("attention_stats") or {}
        print("== %s  (B = %d, attn_gain %g%s)" % (case, fx.B, fx.meta["attn_gain"],
              "; reference attention entropy %s nats" % ", ".join("%s %.2f" % (k, v["entropy_mean"]) for k, v in st.items()) if st else ""))
        print("%-20s | %-23s | %-23s | %-23s | mmt_dec  | qtv ocr  | flips | RMS ref / pos / neg" % ("stage", "ref  vocab / pointer", "pos  vocab / pointer", "neg  vocab / pointer"))
        for name, dt, sw in stages:
            r = run(fx, dt, **sw)
            cells = " | ".join("%.3e / %.3e  " % r[k][:2] for k in ("ref_scores", "pos_scores", "neg_scores"))
            print("%-20s | %s | %.2e | %.2e | %d | %s" % (name, cells, r["mmt_dec"], r["qtv_ocr"], sum(r[k][2] for k in ("ref_scores", "pos_scores", "neg_scores")),
                                                          " / ".join("%.2e" % r[k][3] for k in ("ref_scores", "pos_scores", "neg_scores"))), flush=True)


if __name__ == "__main__":
    main()
