import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from vitxt_gqa_amd import ops
DEV = "cuda:0"
B, L1, n_dec = 1, int(sys.argv[1]) if len(sys.argv) > 1 else 515, 12
dp = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
g = torch.Generator().manual_seed(1)
L = L1 + n_dec
x = (torch.randn(B, L, 2304, generator=g) * 1.0).to(DEV).to(torch.bfloat16)
dout = torch.randn(B, L, 768, generator=g).to(DEV).to(torch.bfloat16)
keys = ops.compact_keys(torch.ones(B, L1, dtype=torch.bool, device=DEV), n_dec=n_dec, dec_row0=L1)
kw = dict(drop_p=dp, drop_seed=7) if dp else {}
out, lse = ops.attn_fwd(x, keys, **kw)
two = ops.attn_bwd(x, out, dout, lse, keys, fused=False, **kw).float()
got = ops.attn_bwd(x, out, dout, lse, keys, fused=True, **kw).float()
torch.cuda.synchronize()
for name, sl in (("dq", slice(0, 768)), ("dk", slice(768, 1536)), ("dv", slice(1536, 2304))):
    a, b = got[0, :, sl], two[0, :, sl]
    nan_rows = torch.isnan(a).any(-1).nonzero().flatten().tolist()
    d = (a - b).abs().nan_to_num(99.0)
    bad = (d > 0.05 * b.abs().max()).any(-1).nonzero().flatten().tolist()
    print(name, "nan rows:", len(nan_rows), nan_rows[:12], "| bad rows:", len(bad), bad[:20], "| max diff %.3e of %.3e" % (d.max(), b.abs().max()))
    if name == "dq" and bad:
        r = bad[0]
        badc = (d[r] > 0.05 * b.abs().max()).nonzero().flatten().tolist()
        print("   row", r, "bad cols", len(badc), badc[:40])
import collections
for name, sl in (("dk", slice(768, 832)), ("dv", slice(1536, 1600))):      # head 0
    a, b = got[0, :384, sl], two[0, :384, sl]
    d = ((a - b).abs().nan_to_num(99.0) > 0.05 * b.abs().max())
    print(name, "head 0: bad by 32-key block:", [int(d[i * 32:(i + 1) * 32].sum()) for i in range(12)])
    print(name, "head 0: bad by dim column:", [int(d[:, c].sum()) for c in range(64)])
