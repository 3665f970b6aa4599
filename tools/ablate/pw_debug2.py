import sys, os, torch, collections
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from vitxt_gqa_amd import ops
DEV = "cuda:0"
g = torch.Generator().manual_seed(300)
B, L1, n_dec = 1, int(sys.argv[1]) if len(sys.argv) > 1 else 503, 12
L = L1 + n_dec
qkv = torch.randn(B, L, 2304, generator=g).to(DEV).to(torch.bfloat16)
v = torch.zeros(B, L, 768, device=DEV)
per = (L + 63) // 64
for k in range(L):
    v[:, k, (k // per) % 64] = 1.0                  # head 0: one-hot bucket of the key position
qkv[..., 1536:1600] = v[..., :64].to(torch.bfloat16)
keys = ops.compact_keys(torch.ones(B, L1, dtype=torch.bool, device=DEV), n_dec=n_dec, dec_row0=L1)
out, lse = ops.attn_fwd(qkv, keys, drop_p=0.0, drop_seed=1)
q, k, vv = [t.view(B, L, 12, 64).permute(0, 2, 1, 3).double() for t in qkv.split(768, dim=-1)]
s = (q @ k.transpose(-1, -2)) * 0.125
vis = torch.ones(L, L, dtype=torch.bool, device=DEV)
vis[:, L1:] = False
vis[L1:, L1:] = torch.tril(torch.ones(n_dec, n_dec, dtype=torch.bool, device=DEV))
s = s.masked_fill(~vis, float("-inf"))
a = torch.softmax(s, -1)
ref = (a @ vv).permute(0, 2, 1, 3).reshape(B, L, 768)
o = out.double()
print("nan count", int(torch.isnan(o).sum()), "of", o.numel())
nanrows = torch.isnan(o).any(-1)[0].nonzero().flatten().tolist()
print("rows with nan:", nanrows[:40], "..." if len(nanrows) > 40 else "")
nancols = torch.isnan(o).any(1)[0].nonzero().flatten().tolist()
print("cols with nan:", nancols[:80])
d = (o - ref)[0, :, :64].abs().nan_to_num(9.0)
bad = (d > 0.004).nonzero()
print("bad entries head 0:", bad.shape[0], " keys per bucket", per)
rows = collections.Counter((int(r) // 32) for r, c in bad.tolist())
cols = collections.Counter(int(c) for r, c in bad.tolist())
print("by query block of 32:", sorted(rows.items()))
print("by key bucket:", sorted(cols.items()))
print("max err all heads", float((o - ref).abs().nan_to_num(9.0).max()))
