#!/usr/bin/env python3
"""Where a steady-state tile of the bf16 attention FORWARD spends its cycles.  Builds a diagnostic library whose forward
source is the product source patched with s_memtime stamps (tools/ablate/make_fwd_diag.py), runs one launch and prints the shares.  Two
waves share a SIMD in this kernel: a wave's stage time includes the cycles its partner holds the issue port or the matrix pipe, so the
figures are WALL cycles of one wave, and 2 x 32 MFMAs x 32 cycles = 2 048 is the matrix pipe's time per tile PAIR (one tile of each of
the SIMD's two waves).   usage (GPU box): python tools/fwd_stamps.py [B keep drop_p]"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = os.path.join(ROOT, "tools", "ablate", "_build")
os.makedirs(out, exist_ok=True)
from vitxt_gqa_amd import build as Bld  # noqa: E402
lib = os.path.join(out, "libt2s_fwd_stamp.so")
diag = os.path.join(out, "attn_fwd_bf16_stamp.hip")
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "ablate", "make_fwd_diag.py"), diag])
srcs = [s for s in Bld.sources() if not s.endswith("attn_fwd_bf16.hip")] + [diag]
subprocess.check_call([Bld.HIPCC] + Bld.FLAGS + ["-w", "-o", lib] + srcs)
os.environ["T2S_HIP_LIB"] = lib
import torch  # noqa: E402
from vitxt_gqa_amd import hipext as X, ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
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
buf = torch.zeros(256 * 8, dtype=torch.int64, device="cuda")
fn = X.lib().t2s_dbg_fwd_stamps
fn.restype, fn.argtypes = ctypes.c_int, [ctypes.c_void_p]
assert fn(buf.data_ptr()) == 0
for _ in range(3):
    o, lse = ops.attn_fwd(qkv, keys, **kw)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
o, lse = ops.attn_fwd(qkv, keys, **kw)
e1.record()
torch.cuda.synchronize()
d = buf.view(-1, 8).cpu()
d = d[d[:, 6] > 0]
names = ["top: next tile's loads issued, row keys", "stage A: S(kb 0)", "stage B: S(kb 1) + softmax(0)", "stage C: PV(0) + softmax(1)", "stage D: PV(1)", "row sums, stage write, barrier"]
tiles = d[:, 6].double()
per = d[:, :6].double() / tiles.unsqueeze(1)
tot = per.sum(1).mean().item()
print("forward, dropout %.2f, B = %d, L = %d, %d%% keys: workgroups sampled %d, steady-state tiles per workgroup %d (stamped launch %.2f ms)"
      % (dp, B, L, int(keep * 100), len(d), int(tiles.mean().item()), e0.elapsed_time(e1)))
for i, n in enumerate(names):
    print("  %-42s %7.0f cycles per tile  (%4.1f %%)" % (n, per[:, i].mean().item(), 100 * per[:, i].mean().item() / tot))
print("  %-42s %7.0f cycles per tile of ONE wave; the SIMD's two waves share the matrix pipe: 2 x 32 MFMAs x 32 = 2 048 cycles per tile pair" % ("sum", tot))
clk = d[:, 7].double().mean().item() / 1e4
print("  shader clock over the loop: %.2f GHz; MFMA share of the pipe if the two waves' tiles alternate perfectly: %.2f" % (clk, 2048.0 / tot))
