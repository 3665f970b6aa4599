import torch, sys
sys.path.insert(0, "/root/repo")
from vitxt_gqa_amd import ops
DEV = "cuda"
def run(B, L1, n_dec, keep, drop_p, dq_mode):
    g = torch.Generator(device="cpu").manual_seed(B * 1000 + L1 + 1)
    L = L1 + n_dec
    x = (torch.randn(B, L, 2304, generator=g) * 1.5).to(DEV).to(torch.bfloat16)
    valid = (torch.rand(B, L1, generator=g) < keep).to(DEV)
    valid[:, 0] = True
    dout = torch.randn(B, L, 768, generator=g).to(DEV).to(torch.bfloat16)
    keys = ops.compact_keys(valid, n_dec=n_dec, dec_row0=L1)
    kw = dict(drop_p=drop_p, drop_seed=4242) if drop_p else {}
    out, lse = ops.attn_fwd(x, keys, **kw)
    two = ops.attn_bwd(x, out, dout, lse, keys, fused=False, **kw)
    for rep in range(3):
        got = ops.attn_bwd(x, out, dout, lse, keys, fused=True, dq_mode=dq_mode, **kw)
        d = (got.float() - two.float()).abs()
        names = ("dq", "dk", "dv")
        msg = []
        for i, n in enumerate(names):
            e = d[..., 768 * i:768 * (i + 1)]
            bad = (e > 0.05 * two.float().abs().max()).nonzero()
            msg.append("%s max %.3g bad %d" % (n, e.max().item(), len(bad)))
            if len(bad) and rep == 0:
                rows = bad[:, 1].unique()
                heads = (bad[:, 2] // 64).unique()
                msg.append("   rows %s..%s (%d distinct) tiles %s heads %s" % (rows.min().item(), rows.max().item(), len(rows), (rows // 64).unique().tolist()[:20], heads.tolist()))
        print("B%d L1=%d drop=%.1f mode=%d rep %d: %s  nan=%d keys=%s" % (B, L1, drop_p, dq_mode, rep, " | ".join(msg), torch.isnan(got.float()).sum().item(), keys.cnt.tolist()))
for cfg in [(2, 1000, 12, 0.7, 0.1, 0), (2, 1000, 12, 0.7, 0.0, 0), (2, 1000, 12, 0.7, 0.1, 1), (2, 2000, 12, 0.9, 0.1, 0), (2, 2000, 12, 0.9, 0.0, 0), (1, 5000, 12, 1.0, 0.1, 0)]:
    run(*cfg)
