#!/usr/bin/env python3
"""Does the in-kernel attention dropout train like i.i.d. dropout?  (VERDICT r4 #5.)

The attention kernels draw their keep mask as a stateless hash: keep(b, h, q, p) = f(rowkey16(q, key window) * colkey16(p, row window))
(csrc/attn_common.h) - cheap enough to regenerate in the backward, but not an i.i.d. Bernoulli draw.  Mask STATISTICS are tested
(tests/test_dropout_gpu.py); this tool compares TRAINING DYNAMICS at the cfg1 shape (B = 2, 20 frames x 30 OCR tokens, L = 652, the full
84 M-parameter model, reference recipe: Adam 1e-4 with warm-up, clip 0.25, every dropout 0.1):
  arm "kernel"   the product path (masks generated inside the HIP attention kernels)
  arm "hash"     attention through a dense torch restatement whose mask is the SAME hash, exported by t2s_attn_dropout_mask
                 (the injection point; shows the restatement trains like the product path)
  arm "iid"      the same restatement with torch Philox Bernoulli(1 - p') masks
over S seeds (the seed drives every dropout draw; weights, data order and the 4 cycled batches are the same in every run) x N steps.
Output: per 20-step window the mean and standard deviation over seeds of the window-mean loss of each arm, and the gap between "hash"
and "iid" in pooled standard deviations.   usage: python tools/dropout_dynamics.py [seeds steps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vitxt_gqa_amd import functional as FN, ops, training_config  # noqa: E402
from vitxt_gqa_amd.optim import build_optimizer, lr_lambda_update, train_step  # noqa: E402
from vitxt_gqa_amd.synth import make_batch, make_noise  # noqa: E402
from vitxt_gqa_amd.testing import make_model, to_device  # noqa: E402

DEV = torch.device("cuda", 0)
H = 12


def _dense(x, keys, scale, keep, p_eff):
    """x [B, L, 2304] fp32 -> [B, L, 768]: softmax over the key LIST of each sample (prefix keys, then decoder keys under the causal rule),
    probabilities * keep / (1 - p')."""
    B, L, _ = x.shape
    cap = keys.idx.shape[1]
    q, k, v = [t.view(B, L, H, 64).permute(0, 2, 1, 3) for t in x.split(768, dim=-1)]
    idx = keys.idx.long().clamp(0, L - 1)                                             # [B, cap] rows of the list positions
    kl = torch.gather(k, 2, idx.view(B, 1, cap, 1).expand(B, H, cap, 64))
    vl = torch.gather(v, 2, idx.view(B, 1, cap, 1).expand(B, H, cap, 64))
    s = (q @ kl.transpose(-1, -2)) * scale                                            # [B, H, L, cap]
    pos = torch.arange(cap, device=x.device).view(1, 1, cap)
    cnt = keys.cnt.view(B, 1, 1).long()
    row = torch.arange(L, device=x.device).view(1, L, 1)
    vis = (pos < cnt) | ((pos >= cnt) & (pos < cnt + keys.n_dec) & (row - keys.dec_q0 >= pos - cnt))      # [B, L, cap]
    s = s.masked_fill(~vis.unsqueeze(1), float("-inf"))
    lse = torch.logsumexp(s, -1)
    a = torch.softmax(s, -1)
    if keep is not None:
        a = a * keep / (1.0 - p_eff)
    return (a @ vl).permute(0, 2, 1, 3).reshape(B, L, 768), lse


def _mask(kind, B, L, cap, drop_p, seed):
    if drop_p <= 0:
        return None, 0.0
    p_eff = round(65536 * drop_p) / 65536.0
    if kind == "hash":
        return ops.attn_dropout_mask(B, L, cap, drop_p, seed, DEV).float(), p_eff
    g = torch.Generator(device=DEV).manual_seed(seed % (2 ** 63))
    return (torch.rand(B, H, L, cap, device=DEV, generator=g) >= p_eff).float(), p_eff


def dense_arm(kind):
    """(attn_fwd, attn_bwd) with the signatures of ops.attn_fwd / ops.attn_bwd, computed densely in fp32 with the given mask kind."""
    def fwd(qkv, keys, scale=0.125, drop_p=0.0, drop_seed=0, kv=None):
        assert kv is None
        keep, pe = _mask(kind, qkv.shape[0], qkv.shape[1], keys.idx.shape[1], drop_p, drop_seed)
        with torch.no_grad():
            out, lse = _dense(qkv.float(), keys, scale, keep, pe)
        return out.to(qkv.dtype), lse

    def bwd(qkv, out, dout, lse, keys, scale=0.125, drop_p=0.0, drop_seed=0, fused=None, kv=None, dq_mode=None):
        assert kv is None
        keep, pe = _mask(kind, qkv.shape[0], qkv.shape[1], keys.idx.shape[1], drop_p, drop_seed)
        with torch.enable_grad():
            x = qkv.float().detach().requires_grad_(True)
            o, _ = _dense(x, keys, scale, keep, pe)
            o.backward(dout.float())
        return x.grad.to(qkv.dtype)
    return fwd, bwd


def run_arm(arm, seed, steps, F=20, P=30, V=5000, B=2, n_batches=4):
    saved = (ops.attn_fwd, ops.attn_bwd, FN.PRUNE_KV_MAX_KEYS)
    try:
        if arm != "kernel":
            ops.attn_fwd, ops.attn_bwd = dense_arm(arm)
        # every arm projects K / V for all rows (no pruned key buffers): the restatement takes the fused [B, L, 2304] projection only, and
        # the mask is indexed by key-LIST position, so "kernel" and "hash" then draw the very same masks
        FN.PRUNE_KV_MAX_KEYS = -1
        model = make_model(F, P, V, seed=0, dtype=torch.bfloat16, dropout=0.1).to(DEV).train()
        cfg = training_config()
        opt = build_optimizer(model, cfg)
        sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda it: lr_lambda_update(it, cfg))
        batches = []
        for i in range(n_batches):
            b = to_device(make_batch(B, F, P, V=V, seed=100 + i), DEV)
            b.grounding_noise = tuple(t.to(DEV) for t in make_noise(B, F, P, seed=100 + i))
            batches.append(b)
        torch.manual_seed(1000 + seed)                     # every dropout seed of the run follows from here
        losses = []
        for it in range(steps):
            loss, _, _ = train_step(model, opt, sched, batches[it % n_batches], cfg)
            losses.append(loss)
        return torch.stack(losses).float().cpu()
    finally:
        ops.attn_fwd, ops.attn_bwd, FN.PRUNE_KV_MAX_KEYS = saved


def run(seeds, steps, arms=("kernel", "hash", "iid"), window=20):
    """-> {arm: tensor [seeds, steps // window] of window-mean losses}"""
    res = {}
    for arm in arms:
        curves = torch.stack([run_arm(arm, s, steps) for s in range(seeds)])
        res[arm] = curves[:, :(steps // window) * window].view(seeds, -1, window).mean(-1)
    return res


def table(res, window=20):
    arms = list(res)
    lines = ["steps        " + "   ".join("%-22s" % a for a in arms) + ("   |hash - iid| / pooled sd" if "hash" in res and "iid" in res else "")]
    nw = res[arms[0]].shape[1]
    worst = 0.0
    for w in range(nw):
        cells = ["%8.4f +- %-10.4f" % (res[a][:, w].mean().item(), res[a][:, w].std().item()) for a in arms]
        tail = ""
        if "hash" in res and "iid" in res:
            a, b = res["hash"][:, w], res["iid"][:, w]
            sd = ((a.var() + b.var()) / 2).sqrt().item()
            gap = abs(a.mean().item() - b.mean().item()) / max(sd, 1e-9)
            worst = max(worst, gap)
            tail = "   %.2f" % gap
        lines.append("%4d - %-4d  " % (w * window + 1, (w + 1) * window) + "   ".join(cells) + tail)
    return "\n".join(lines), worst


if __name__ == "__main__":
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    print("# attention-dropout training dynamics at the cfg1 shape (B = 2, 20 x 30, L = 652; V = 5000; bf16 operands; every dropout 0.1;")
    print("# Adam 1e-4 with the reference's warm-up, clip 0.25; 4 cycled batches; %d seeds x %d steps): window-mean train loss, mean +- sd over seeds" % (seeds, steps))
    res = run(seeds, steps)
    txt, worst = table(res)
    print(txt)
    print("# largest gap between the hash-mask and the i.i.d.-mask arm: %.2f pooled standard deviations of a window mean" % worst)
    k, h = res["kernel"], res["hash"]
    print("# product kernels vs their dense restatement with the exported masks: largest window gap %.4f (loss units; bf16 kernels vs an fp32 restatement)"
          % (k.mean(0) - h.mean(0)).abs().max().item())
