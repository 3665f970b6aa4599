#!/usr/bin/env python3
"""Where a query tile of the INTERLEAVED fused attention backward (round 5 variant, tools/ablate/attn_bwd_fused_bf16_ilv256.hip: 256 keys per
workgroup, the dQ product of tile t - 1 inside phase A of tile t; measured 7-8 % SLOWER than the shipped 384-key kernel) spends its cycles.
Builds a diagnostic library whose fused-backward source is that variant patched with s_memtime stamps (tools/ablate/make_fb_diag.py), runs
one launch and prints the shares.  The stamps serialise the segments (fences + an
lgkmcnt(0) each), so read the SHARES, not the total.   usage (GPU box): python tools/fused_stamps2.py [B keep drop_p]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = os.path.join(ROOT, "tools", "ablate", "_build")
os.makedirs(out, exist_ok=True)
from vitxt_gqa_amd import build as Bld  # noqa: E402
lib = os.path.join(out, "libt2s_stamp2.so")
diag = os.path.join(out, "attn_bwd_fused_bf16_stamp.hip")
WHICH = os.environ.get("FB_SRC", "ilv256")       # "ilv256": the 256-key variant; "ilv384": the 384-key interleaved variant; "product"; or a .hip path
if os.environ.get("FB_LIB"):                     # a stamped library built beforehand (tools/ablate/build_fb_libs.sh): no compile on the GPU box
    lib = os.environ["FB_LIB"]
    WHICH = os.environ.get("FB_SRC", "product")
else:
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "ablate", "make_fb_diag.py"), WHICH, diag])
    srcs = [s for s in Bld.sources() if not s.endswith("attn_bwd_fused_bf16.hip")] + [diag]
    subprocess.check_call([Bld.HIPCC] + Bld.FLAGS + ["-w", "-o", lib] + srcs)
os.environ["T2S_HIP_LIB"] = lib
os.environ["T2S_KEEP_DQ32"] = "1"
import torch  # noqa: E402
from vitxt_gqa_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
keep = float(sys.argv[2]) if len(sys.argv) > 2 else 0.7
dp = float(sys.argv[3]) if len(sys.argv) > 3 else 0.1
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
    ops.attn_bwd(qkv, o, dout, lse, keys, fused=True, **kw)
torch.cuda.synchronize()
d = ops._LAST_DQ32.view(torch.uint8)[-16384:].view(torch.int64).view(-1, 8)[:256].cpu()
d = d[d[:, 6] > 0]
names = (["slot 0: G1(b0) + 16 dQ MFMAs", "slots 1..3, dQ hand-off", "slots 4..7"] if WHICH == "ilv256" else
         ["slot 0: G1(b0) + 12 dQ MFMAs", "slot 1: G1(b1) + E(b0) + 12 dQ MFMAs, hand-off", "slots 2..11"]) + [ "K^T preload, vmcnt(0), flag wait", "stage write, sum loads, barrier", "preload behind the barrier"]
mfmas = 80 if WHICH == "ilv256" else 120
tiles = d[:, 6].double()
per = d[:, :6].double() / tiles.unsqueeze(1)
print("dropout %.2f, B = %d: workgroups sampled %d, tiles per workgroup %d" % (dp, B, len(d), int(tiles[0])))
tot = per.sum(1).mean().item()
for i, n in enumerate(names):
    print("  %-36s %8.0f cycles per tile  (%4.1f %%)" % (n, per[:, i].mean().item(), 100 * per[:, i].mean().item() / tot))
print("  %-36s %8.0f cycles per tile (stamped build; MFMA time of a tile: %d x 32 = %d)" % ("sum", tot, mfmas, mfmas * 32))
clk = d[:, 7].double().mean().item() / 1e4
print("  shader clock over the sweep: %.2f GHz; a tile = %.2f us" % (clk, tot / clk / 1e3))
