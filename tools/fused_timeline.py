#!/usr/bin/env python3
"""When and where every workgroup of the fused attention backward ran: builds a DIAGNOSTIC copy of the library with -DFB_TIMELINE
(each workgroup records its start / end on the 100 MHz real-time counter, its HW_ID / XCC_ID and its key block), runs the launch
and prints: the launch's span against the sum of the workgroups' own run times (= how full the 256 CUs were), how many
workgroups ran at a time, the run time of a workgroup by kind (full / edge key block) and by position in the launch, and the gaps
a CU leaves between two workgroups.  Unlike the cycle stamps (tools/fused_stamps.py) this build adds no fences to the sweep.
usage (GPU box): python tools/fused_timeline.py [B keep drop_p dq_mode]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = os.path.join(ROOT, "tools", "ablate", "_build")
os.makedirs(out, exist_ok=True)
lib = os.path.join(out, "libt2s_timeline.so")
from vitxt_gqa_amd import build as Bld  # noqa: E402
if not os.path.exists(lib) or os.environ.get("FB_REBUILD", "0") == "1":
    subprocess.check_call([Bld.HIPCC] + Bld.FLAGS + ["-DFB_TIMELINE"] + ["-o", lib] + Bld.sources())
os.environ["T2S_HIP_LIB"] = lib
os.environ["T2S_KEEP_DQ32"] = "1"
import torch  # noqa: E402
from vitxt_gqa_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
keep = float(sys.argv[2]) if len(sys.argv) > 2 else 0.7
dp = float(sys.argv[3]) if len(sys.argv) > 3 else 0.1
dq_mode = int(sys.argv[4]) if len(sys.argv) > 4 else 1
L1, nd = 10120, 12
L = L1 + nd
torch.manual_seed(0)
qkv = torch.randn(B, L, 2304, device="cuda", dtype=torch.bfloat16)
dout = torch.randn(B, L, 768, device="cuda", dtype=torch.bfloat16)
valid = torch.rand(B, L1, device="cuda") < keep
valid[:, 0] = True
keys = ops.compact_keys(valid, n_dec=nd, dec_row0=L1)
kw = dict(drop_p=dp, drop_seed=77) if dp > 0 else {}
o, lse = ops.attn_fwd(qkv, keys, **kw)
for _ in range(3):
    ops.attn_bwd(qkv, o, dout, lse, keys, fused=True, dq_mode=dq_mode, **kw)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
ops.attn_bwd(qkv, o, dout, lse, keys, fused=True, dq_mode=dq_mode, **kw)
b.record()
torch.cuda.synchronize()
ms_call = a.elapsed_time(b)
ws = ops._LAST_DQ32
tail_bytes = 16384 + 131072 * 32
tl = ws.view(torch.uint8)[-tail_bytes + 16384:].view(torch.int64).view(-1, 4).cpu()
tl = tl[tl[:, 1] > 0]
allwg = tl
if os.environ.get("FB_TL_SAVE"):                           # raw records [n, 4] int64 (start, end: 10 ns ticks; HW_ID | XCC_ID << 32; key block word) + workgroup ids
    import numpy as np
    full = ws.view(torch.uint8)[-tail_bytes + 16384:].view(torch.int64).view(-1, 4).cpu()
    ids = torch.nonzero(full[:, 1] > 0).flatten()
    np.savez_compressed(os.environ["FB_TL_SAVE"], rec=full[ids].numpy(), wg=ids.numpy())
tl = tl[tl[:, 3] >= 0]                                     # bit 63 marks a workgroup that had no key block to sweep (left at once)
r0, r1 = tl[:, 0].double(), tl[:, 1].double()
t0 = r0.min()
r0, r1 = (r0 - t0) / 100.0, (r1 - t0) / 100.0             # microseconds
dur = r1 - r0
span = r1.max().item()
hw, xcc = tl[:, 2] & 0xFFFFFFFF, (tl[:, 2] >> 32) & 0xF
cu, sh, se = (hw >> 8) & 0xF, (hw >> 12) & 0x1, (hw >> 13) & 0x7
cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
kb, edge, nkeys = tl[:, 3] & 0xFF, (tl[:, 3] >> 48) & 1, (tl[:, 3] >> 32) & 0xFFFF
ncu = len(torch.unique(cuid))
print("dq_mode %d, B=%d, dropout %.2f: op (prep + sweep) %.3f ms by events; sweep span %.3f ms from the workgroups' own clocks" % (dq_mode, B, dp, ms_call, span / 1e3))
print("workgroups that ran a sweep: %d (%d full, %d edge), on %d distinct CUs" % (len(tl), int((edge == 0).sum()), int((edge == 1).sum()), ncu))
print("sum of workgroup run times / (CUs x span) = %.3f   (1.0 = every CU busy from the first start to the last end)" % (dur.sum().item() / (ncu * span)))
for name, m in (("full", edge == 0), ("edge", edge == 1)):
    if m.any():
        d = dur[m]
        print("  %s blocks: run time median %.1f us, p10 %.1f, p90 %.1f, min %.1f, max %.1f" % (name, d.median().item(), d.quantile(0.1).item(), d.quantile(0.9).item(), d.min().item(), d.max().item()))
# concurrency over the span, 50 samples
ts = torch.linspace(0, span, 52)[1:-1].double()
conc = [(int(((r0 <= t) & (r1 > t)).sum())) for t in ts]
print("workgroups in flight at 50 points of the span: min %d, median %d, max %d; first 5 %s, last 5 %s" % (min(conc), sorted(conc)[25], max(conc), conc[:5], conc[-5:]))
# run time by start order (deciles)
order = torch.argsort(r0)
n = len(order)
print("run time by start order (deciles of the launch, median us): " + " ".join("%.0f" % dur[order[i * n // 10:(i + 1) * n // 10]].median().item() for i in range(10)))
print("run time by key block index (median us): " + " ".join("%d:%.0f" % (k, dur[kb == k].median().item()) for k in sorted(set(kb.tolist()))))
# gaps on a CU between the end of one workgroup and the start of the next
gaps = []
per_cu = []
for c in torch.unique(cuid):
    m = cuid == c
    s, e = r0[m], r1[m]
    o_ = torch.argsort(s)
    s, e = s[o_], e[o_]
    if len(s) > 1:
        gaps.append(s[1:] - e[:-1])
    per_cu.append((e - s).sum().item())
g = torch.cat(gaps)
print("gap on a CU between two consecutive workgroups: median %.1f us, p90 %.1f, max %.1f; negative (two at once on one CU): %d of %d" % (g.median().item(), g.quantile(0.9).item(), g.max().item(), int((g < 0).sum()), len(g)))
pc = torch.tensor(per_cu)
print("busy time per CU / span: min %.3f, median %.3f, max %.3f" % (pc.min().item() / span, pc.median().item() / span, pc.max().item() / span))
print("workgroups per XCD: " + " ".join("%d" % int((xcc == x).sum()) for x in range(8)))
# the workgroups that leave at once (key blocks beyond a sample's list: the grid is sized by the static bound)
dm = allwg[allwg[:, 3] < 0]
if len(dm):
    d0, d1 = (dm[:, 0].double() - t0) / 100.0, (dm[:, 1].double() - t0) / 100.0
    print("workgroups that leave at once: %d, run time median %.2f us, p90 %.2f, max %.2f; started over %.1f .. %.1f ms of the span"
          % (len(dm), (d1 - d0).median().item(), (d1 - d0).quantile(0.9).item(), (d1 - d0).max().item(), d0.min().item() / 1e3, d0.max().item() / 1e3))
    dhw, dxcc = dm[:, 2] & 0xFFFFFFFF, (dm[:, 2] >> 32) & 0xF
    dcu = ((dxcc * 8 + ((dhw >> 13) & 0x7)) * 2 + ((dhw >> 12) & 0x1)) * 16 + ((dhw >> 8) & 0xF)
    # how much of each gap between two sweeps on a CU is covered by such workgroups, and how many sit in it
    cov, cnt, big = [], [], []
    for c in torch.unique(cuid):
        m = cuid == c
        s_, e_ = r0[m], r1[m]
        o_ = torch.argsort(s_)
        s_, e_ = s_[o_], e_[o_]
        md = dcu == c
        ds, de = d0[md], d1[md]
        for i in range(len(s_) - 1):
            inside = (ds >= e_[i]) & (de <= s_[i + 1])
            gap = (s_[i + 1] - e_[i]).item()
            cnt.append(int(inside.sum()))
            cov.append((de[inside] - ds[inside]).sum().item())
            if gap > 50:
                big.append((gap, int(inside.sum()), (de[inside] - ds[inside]).sum().item()))
    cnt_t, cov_t = torch.tensor(cnt, dtype=torch.double), torch.tensor(cov)
    print("per gap between two sweeps on a CU: leave-at-once workgroups inside: mean %.2f, max %d; their run time covers %.1f %% of all gap time"
          % (cnt_t.mean().item(), int(cnt_t.max().item()), 100 * cov_t.sum().item() / max(g.clamp(min=0).sum().item(), 1e-9)))
    if big:
        bg = torch.tensor(big)
        print("gaps > 50 us: %d, holding on average %.1f such workgroups that cover %.1f %% of those gaps" % (len(big), bg[:, 1].mean().item(), 100 * bg[:, 2].sum().item() / bg[:, 0].sum().item()))
