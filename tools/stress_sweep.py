#!/usr/bin/env python3
"""BASELINE.json configs[4] (300 frames x 200 OCR: L = 60 332 rows per question, attention = 93 % of the FLOPs): the attention kernels
alone at that length, every geometry the build has, interleaved in ONE process:
  forward   the shipped two-waves-per-SIMD kernel (64-key tiles); (round 3-4 also ran the one-wave-per-SIMD kernel with 128-key tiles here:
            tools/ablate/attn_fwd_pw_bf16.hip, slower, no longer in the product library)
  backward  two-kernel (128-key blocks, 7 products) vs fused 384-key blocks with the dQ hand-off vs fused with fp32 atomics
over B in {1, 2, 4}, dropout 0.1 and 0.  usage: stress_sweep.py [L1 keep]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vitxt_gqa_amd import ops  # noqa: E402

L1 = int(sys.argv[1]) if len(sys.argv) > 1 else 20 + 300 + 300 * 200
keep = float(sys.argv[2]) if len(sys.argv) > 2 else 0.7
n_dec = 12
L = L1 + n_dec
dev = "cuda:0"


def med(v):
    v = sorted(v)
    return v[len(v) // 2]


for B in (1, 2, 4):
    torch.manual_seed(0)
    qkv = torch.randn(B, L, 2304, device=dev, dtype=torch.bfloat16)
    dout = torch.randn(B, L, 768, device=dev, dtype=torch.bfloat16)
    valid = torch.rand(B, L1, device=dev) < keep
    valid[:, 0] = True
    keys = ops.compact_keys(valid, n_dec=n_dec, dec_row0=L1)
    nk = float(keys.cnt.float().mean().item()) + n_dec
    for dp in (0.1, 0.0):
        kw = dict(drop_p=dp, drop_seed=7) if dp else {}
        fw = {"fwd shipped (64-key tiles, 2 waves/SIMD)": "0"}
        bw = {"bwd two-kernel (128-key blocks)": dict(fused=False), "bwd fused 384 keys, hand-off": dict(fused=True, dq_mode=1),
              "bwd fused 384 keys, atomics": dict(fused=True, dq_mode=0)}
        t = {k: [] for k in list(fw) + list(bw)}
        os.environ["T2S_ATTN_FWD_PW"] = "0"
        out, lse = ops.attn_fwd(qkv, keys, **kw)
        for _ in range(4):
            for name, env in fw.items():
                os.environ["T2S_ATTN_FWD_PW"] = env
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                o2, _ = ops.attn_fwd(qkv, keys, **kw)
                b.record()
                torch.cuda.synchronize()
                t[name].append(a.elapsed_time(b))
            os.environ["T2S_ATTN_FWD_PW"] = "0"
            for name, f in bw.items():
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                g = ops.attn_bwd(qkv, out, dout, lse, keys, **f, **kw)
                b.record()
                torch.cuda.synchronize()
                t[name].append(a.elapsed_time(b))
                del g
        fl_f = 4.0 * B * 12 * L * nk * 64
        print("B=%d L=%d keys=%.0f dropout %.1f" % (B, L, nk, dp))
        for name, v in t.items():
            m = med(v[1:])
            fl = fl_f if name.startswith("fwd") else 2.5 * fl_f
            print("   %-44s %9.2f ms   %6.0f TFLOP/s (algorithmic)  frac %.3f" % (name, m, fl / m / 1e9, fl / m / 1e9 / 2500))
    del qkv, dout, out, lse
    torch.cuda.empty_cache()
print("hand-off status word:", ops.fused_handoff_status())
