#!/usr/bin/env python3
"""Times the attention kernels (fwd, bwd) in isolation at a given shape with HIP events.
usage: python tools/attn_probe.py [B L keep n_dec iters drop_p]"""
import sys
import time

import torch

import os  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vitxt_gqa_amd import ops  # noqa: E402


def main():
    a = sys.argv[1:]
    B = int(a[0]) if a else 8
    L1 = int(a[1]) if len(a) > 1 else 10120
    keep = float(a[2]) if len(a) > 2 else 1.0
    n_dec = int(a[3]) if len(a) > 3 else 12
    iters = int(a[4]) if len(a) > 4 else 5
    dp = float(a[5]) if len(a) > 5 else 0.0
    kw = dict(drop_p=dp, drop_seed=1234) if dp > 0 else {}
    L = L1 + n_dec
    dev = "cuda:0"
    torch.manual_seed(0)
    qkv = torch.randn(B, L, 2304, device=dev, dtype=torch.bfloat16)
    dout = torch.randn(B, L, 768, device=dev, dtype=torch.bfloat16)
    valid = torch.rand(B, L1, device=dev) < keep
    valid[:, 0] = True
    keys = ops.compact_keys(valid, n_dec=n_dec, dec_row0=L1)
    nk = float(keys.cnt.float().mean().item()) + n_dec
    out, lse = ops.attn_fwd(qkv, keys, **kw)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    ev[0].record()
    for _ in range(iters):
        out, lse = ops.attn_fwd(qkv, keys, **kw)
    ev[1].record()
    torch.cuda.synchronize()
    # backward: the two-kernel (7-product) and the fused (5-product, no dropout) forms, interleaved rounds in ONE process
    # (fused: dQ across key blocks by the ordered hand-off, dq_mode 1, and by fp32 atomics + cast, dq_mode 0)
    # "split": full and edge key blocks as two launches (rounds 2-3, T2S_FB_SPLIT_EDGE=1, read per call); default: one launch
    HO = 0x201 if os.environ.get("T2S_FB_HANDOFF_SCOPE", "xcd") == "agent" else 1      # (write-through running sums: the round-4 form)
    forms = [("two-kernel", dict(fused=False), "0"), ("fused/handoff", dict(fused=True, dq_mode=HO), "0"), ("fused/atomic", dict(fused=True, dq_mode=0), "0"),
             ("fused/handoff split", dict(fused=True, dq_mode=HO), "1"), ("fused/atomic split", dict(fused=True, dq_mode=0), "1")]
    if os.environ.get("T2S_PROBE_FORMS") == "shipped":          # PMC passes: only the forms the product runs
        forms = forms[:2]
    tb_all = {n: [] for n, _, _ in forms}
    for n, f, env in forms:
        os.environ["T2S_FB_SPLIT_EDGE"] = env
        ops.attn_bwd(qkv, out, dout, lse, keys, **f, **kw)
    torch.cuda.synchronize()
    for _ in range(iters):
        for n, f, env in forms:
            os.environ["T2S_FB_SPLIT_EDGE"] = env
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            ops.attn_bwd(qkv, out, dout, lse, keys, **f, **kw)
            b.record()
            torch.cuda.synchronize()
            tb_all[n].append(a.elapsed_time(b))
    ev[2].record()
    ev[3].record()
    torch.cuda.synchronize()
    os.environ["T2S_FB_SPLIT_EDGE"] = "0"
    print("   hand-off status word: %d" % ops.fused_handoff_status())
    tf = ev[0].elapsed_time(ev[1]) / iters
    tb = sorted(tb_all["two-kernel"])[len(tb_all["two-kernel"]) // 2]
    dense = 4.0 * B * 12 * L * L * 64
    execd = 4.0 * B * 12 * L * nk * 64
    print("drop=%.2f " % dp, end="")
    print("B=%d L=%d keys=%.0f  fwd %.3f ms: %.1f TF/s dense-equivalent, %.1f TF/s executed | bwd %.3f ms: %.1f TF/s (2.5x fwd flops) executed %.1f"
          % (B, L, nk, tf, dense / tf / 1e9, execd / tf / 1e9, tb, 2.5 * dense / tb / 1e9, 2.5 * execd / tb / 1e9))
    for n, ts in tb_all.items():
        ts = sorted(ts)
        print("   bwd %-20s median %.3f ms  min %.3f ms  -> %.1f TF/s on the algorithmic 5 products (executed keys)" % (n, ts[len(ts) // 2], ts[0], 2.5 * execd / ts[len(ts) // 2] / 1e9))


if __name__ == "__main__":
    main()
