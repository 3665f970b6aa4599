#!/usr/bin/env python3
"""When and where every workgroup of the attention FORWARD ran (a -DT2S_FWD_TIMELINE build: start and last wave's end on the 100 MHz
real-time counter, HW_ID / XCC_ID, query block, sample, head, keys): span of the launch against the workgroups' own run times (two
workgroups share a CU), run time by sample / query block, gaps, balance over the XCDs.
usage (GPU box): python tools/fwd_timeline.py [B keep drop_p]"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = os.path.join(ROOT, "tools", "ablate", "_build")
os.makedirs(out, exist_ok=True)
lib = os.path.join(out, "libt2s_fwd_timeline.so")
from vitxt_gqa_amd import build as Bld  # noqa: E402
if not os.path.exists(lib) or os.environ.get("FB_REBUILD", "0") == "1":
    # the product forward source carries no diagnostics: the timeline build links tools/ablate/attn_fwd_bf16_diag.hip in its place
    srcs = [s_ for s_ in Bld.sources() if not s_.endswith("attn_fwd_bf16.hip")] + [os.path.join(ROOT, "tools", "ablate", "attn_fwd_bf16_diag.hip")]
    subprocess.check_call([Bld.HIPCC] + Bld.FLAGS + ["-w", "-DT2S_FWD_TIMELINE", "-I" + os.path.join(ROOT, "vitxt_gqa_amd", "csrc")] + ["-o", lib] + srcs)
os.environ["T2S_HIP_LIB"] = lib
import torch  # noqa: E402
from vitxt_gqa_amd import hipext as X  # noqa: E402
from vitxt_gqa_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
keep = float(sys.argv[2]) if len(sys.argv) > 2 else 0.7
dp = float(sys.argv[3]) if len(sys.argv) > 3 else 0.1
L1, nd = 10120, 12
L = L1 + nd
torch.manual_seed(0)
qkv = torch.randn(B, L, 2304, device="cuda", dtype=torch.bfloat16)
valid = torch.rand(B, L1, device="cuda") < keep
valid[:, 0] = True
keys = ops.compact_keys(valid, n_dec=nd, dec_row0=L1)
kw = dict(drop_p=dp, drop_seed=77) if dp > 0 else {}
for _ in range(3):
    ops.attn_fwd(qkv, keys, **kw)
torch.cuda.synchronize()
cap = 262144
buf = torch.zeros(cap * 4, dtype=torch.int64, device="cuda")
fn = X.lib().t2s_dbg_fwd_timeline
fn.restype, fn.argtypes = ctypes.c_int, [ctypes.c_void_p]
assert fn(buf.data_ptr()) == 0
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
ops.attn_fwd(qkv, keys, **kw)
b.record()
torch.cuda.synchronize()
assert fn(None) == 0
tl = buf.view(-1, 4).cpu()
tl = tl[tl[:, 1] > 0]
r0, r1 = tl[:, 0].double(), tl[:, 1].double()
t0 = r0.min()
r0, r1 = (r0 - t0) / 100.0, (r1 - t0) / 100.0
dur = r1 - r0
span = r1.max().item()
hw, xcc = tl[:, 2] & 0xFFFFFFFF, (tl[:, 2] >> 32) & 0xF
cuid = ((xcc * 8 + ((hw >> 13) & 7)) * 2 + ((hw >> 12) & 1)) * 16 + ((hw >> 8) & 0xF)
qb, sb, nk = tl[:, 3] & 0xFFFF, (tl[:, 3] >> 16) & 0xFFFF, (tl[:, 3] >> 40) & 0xFFFFF
ncu = len(torch.unique(cuid))
print("forward, B=%d, dropout %.2f: %.3f ms by events; span %.3f ms from the workgroups' own clocks; %d workgroups on %d CUs" % (B, dp, a.elapsed_time(b), span / 1e3, len(tl), ncu))
print("sum of workgroup run times / (2 x CUs x span) = %.3f   (two workgroups share a CU)" % (dur.sum().item() / (2 * ncu * span)))
print("run time: median %.1f us, p10 %.1f, p90 %.1f, min %.1f, max %.1f; per listed key: median %.2f ns" % (
    dur.median().item(), dur.quantile(0.1).item(), dur.quantile(0.9).item(), dur.min().item(), dur.max().item(), (dur / nk.double()).median().item() * 1e3))
ts = torch.linspace(0, span, 52)[1:-1].double()
conc = [int(((r0 <= t) & (r1 > t)).sum()) for t in ts]
print("workgroups in flight at 50 points: min %d, median %d, max %d; first 5 %s, last 5 %s" % (min(conc), sorted(conc)[25], max(conc), conc[:5], conc[-5:]))
order = torch.argsort(r0)
n = len(order)
print("run time by start order (deciles, median us): " + " ".join("%.0f" % dur[order[i * n // 10:(i + 1) * n // 10]].median().item() for i in range(10)))
nqb = int(qb.max().item()) + 1
print("run time by query block (first, middle, last two; median us): " + " ".join("%d:%.0f" % (k, dur[qb == k].median().item()) for k in (0, 1, nqb // 2, nqb - 2, nqb - 1)))
per = []
busy = []
for c in torch.unique(cuid):
    m = cuid == c
    s, e = r0[m], r1[m]
    # time during which the CU holds 0 / 1 / 2 workgroups
    ev = sorted([(x.item(), 1) for x in s] + [(x.item(), -1) for x in e])
    lvl, last, acc = 0, 0.0, [0.0, 0.0, 0.0, 0.0]
    for t, d in ev:
        acc[min(lvl, 3)] += t - last
        last, lvl = t, lvl + d
    acc[0] += span - last
    per.append(acc)
pa = torch.tensor(per)
tot = pa.sum(1, keepdim=True)
fr = (pa / tot).mean(0)
print("time a CU holds 0 / 1 / 2 / 3+ workgroups (mean over CUs): %.3f / %.3f / %.3f / %.3f" % tuple(fr.tolist()))
print("workgroups per XCD: " + " ".join("%d" % int((xcc == x).sum()) for x in range(8)))
print("last end per XCD (ms): " + " ".join("%.2f" % (r1[xcc == x].max().item() / 1e3) for x in range(8)))
