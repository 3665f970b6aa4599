"""For each attention-backward call of one training step: how many query rows arrive with an exactly-zero dO row, and how that
compares with the rows that are not keys.  usage: python tools/ablate/live_rows_probe.py [B F P]"""
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from vitxt_gqa_amd import ops, training_config  # noqa: E402
from vitxt_gqa_amd.synth import make_batch, make_noise  # noqa: E402
from vitxt_gqa_amd.testing import make_model, to_device  # noqa: E402

B, F, P = [int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (4, 20, 20))]
dev = "cuda:0"
model = make_model(F, P, 5000, seed=0, dtype=torch.bfloat16, dropout=0.1).to(dev)
model.train(True)
batch = to_device(make_batch(B, F, P, V=5000, seed=100), dev)
batch.grounding_noise = tuple(t.to(dev) for t in make_noise(B, F, P, seed=100))
orig = ops.attn_bwd


def spy(qkv, out, dout, lse, keys, *a, **kw):
    Bq, L, _ = dout.shape
    zero_rows = (dout == 0).all(-1)
    nan = int(torch.isnan(dout).sum())
    listed = torch.zeros(Bq, L, dtype=torch.bool, device=dout.device)
    for b in range(Bq):
        n = int(keys.cnt[b]) + keys.n_dec
        listed[b, keys.idx[b, :n].long()] = True
    print("attn_bwd L=%d cap_hint=%d: zero-dO rows %.3f | rows that are not keys %.3f | zero among non-keys %.3f | zero among keys %.3f | nan %d | max|dO| %.3e"
          % (L, keys.cap_hint, zero_rows.float().mean(), (~listed).float().mean(), zero_rows[~listed].float().mean() if (~listed).any() else -1,
             zero_rows[listed].float().mean(), nan, dout.float().abs().max()))
    nz = dout[~listed].float().abs()
    if nz.numel():
        print("     non-key rows: max|dO| %.3e, median row max %.3e" % (nz.max(), nz.max(-1).values.median()))
    return orig(qkv, out, dout, lse, keys, *a, **kw)


ops.attn_bwd = spy
import vitxt_gqa_amd.functional as FN  # noqa: E402
if hasattr(FN, "ops"):
    FN.ops.attn_bwd = spy
out = model(batch)
loss = sum(l.mean() for l in out["losses"].values())
loss.backward()
torch.cuda.synchronize()
print("loss", float(loss))
