#!/usr/bin/env python3
"""Where a query tile of the fused attention backward spends its cycles: builds a DIAGNOSTIC copy of the library with -DFB_STAMP
(s_memtime stamps around the segments of the tile loop, summed per workgroup by wave 0), runs one launch and prints the
shares.  The stamps serialise the segments (fences), so read the SHARES, not the total (cdna_hip_programming.md section 7).
usage (GPU box): [FB_SLOTS=1] [FB_ABL=n] [FB_MODE=1 (stamp the edge-block kernel)] python tools/fused_stamps.py [B keep drop_p]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = os.path.join(ROOT, "tools", "ablate", "_build")
os.makedirs(out, exist_ok=True)
from vitxt_gqa_amd import build as Bld  # noqa: E402
abl = os.environ.get("FB_ABL", "0")          # timing-only ablations: 1 = no dQ operand reads, 2 = no dQ MFMAs (results wrong)
slots = os.environ.get("FB_SLOTS", "0") == "1"   # stamp the slot classes of phase A instead of the tile's segments
mode = os.environ.get("FB_MODE", "0")
lib = os.path.join(out, "libt2s_stamp_a%s_s%d_m%s.so" % (abl, int(slots), mode))
if not os.path.exists(lib) or os.environ.get("FB_REBUILD", "0") == "1":
    subprocess.check_call([Bld.HIPCC] + Bld.FLAGS + ["-w", "-DFB_STAMP", "-DFB_TIMELINE", "-DFB_ABL=" + abl] + (["-DFB_STAMP_SLOTS"] if slots else []) + ["-DFB_STAMP_MODE=" + mode] + ["-o", lib] + Bld.sources())
os.environ["T2S_HIP_LIB"] = lib
os.environ["T2S_KEEP_DQ32"] = "1"            # ops keeps the fused backward's workspace (the stamps sit in its tail)
import torch  # noqa: E402
from vitxt_gqa_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
keep = float(sys.argv[2]) if len(sys.argv) > 2 else 0.7
L1, nd = 10120, 12
L = L1 + nd
torch.manual_seed(0)
qkv = torch.randn(B, L, 2304, device="cuda", dtype=torch.bfloat16)
dout = torch.randn(B, L, 768, device="cuda", dtype=torch.bfloat16)
valid = torch.rand(B, L1, device="cuda") < keep
valid[:, 0] = True
keys = ops.compact_keys(valid, n_dec=nd, dec_row0=L1)
dp = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
kw = dict(drop_p=dp, drop_seed=77) if dp > 0 else {}
o, lse = ops.attn_fwd(qkv, keys, **kw)
for _ in range(3):
    ops.attn_bwd(qkv, o, dout, lse, keys, fused=True, dq_mode=int(sys.argv[4]) if len(sys.argv) > 4 else None, **kw)
torch.cuda.synchronize()
# the stamps sit in the last 16 KB of the workspace (FbWork.dbg)
tail_bytes = 16384 + 131072 * 32          # built with -DFB_TIMELINE as well: the stamps are the first 16 KB of the tail
d = ops._LAST_DQ32.view(torch.uint8)[-tail_bytes:-tail_bytes + 16384].view(torch.int64).view(-1, 8)[:256].cpu()
d = d[d[:, 6] > 0]
names = ["phase A", "stage write", "barrier 1", "phase B", "atomics", "barrier 2"]
if slots:
    names = ["G1(b0)      x1", "G1+E        x5", "G2+M 1st half x5", "G2+M 2nd half x4", "G2M4+E5, G2M5 2nd", "behind A"]
tiles = d[:, 6].double()
per = d[:, :6].double() / tiles.unsqueeze(1)
print("workgroups sampled: %d, tiles per workgroup %d" % (len(d), int(tiles[0])))
tot = per.sum(1).mean().item()
for i, n in enumerate(names):
    print("  %-12s %8.0f cycles per tile  (%4.1f %%)" % (n, per[:, i].mean().item(), 100 * per[:, i].mean().item() / tot))
print("  %-12s %8.0f cycles per tile (stamped build; MFMA time of a tile: 120 x 32 = 3840)" % ("sum", tot))
print("  shader clock over the sweep (s_memtime per 100 MHz s_memrealtime tick): %.2f GHz (min %.2f, max %.2f over the workgroups)"
      % (d[:, 7].double().mean().item() / 1e4, d[:, 7].min().item() / 1e4, d[:, 7].max().item() / 1e4))
clk = d[:, 7].double().mean().item() / 1e4
print("  a tile = %.2f us at that clock; the stamped workgroup's sweep = %.0f us" % (tot / clk / 1e3, tot * tiles[0].item() / clk / 1e3))
tl = ops._LAST_DQ32.view(torch.uint8)[-tail_bytes + 16384:].view(torch.int64).view(-1, 4).cpu()
tl = tl[(tl[:, 1] > 0) & (tl[:, 3] >= 0)]
dur = (tl[:, 1] - tl[:, 0]).double() / 100.0
print("  workgroup run times of the same launch by the 100 MHz counter: median %.0f us (full blocks %.0f, edge blocks %.0f)"
      % (dur.median().item(), dur[((tl[:, 3] >> 48) & 1) == 0].median().item(), dur[((tl[:, 3] >> 48) & 1) == 1].median().item()))
