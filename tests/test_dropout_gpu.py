"""Fused hidden-state dropout (BertSelfOutput / BertOutput dropout inside the residual+LayerNorm kernels).
The mask is a stateless hash of (seed, element index); t2s_dropout_mask exports it so that the forward, the backward
and the fused BERT layer can be checked against a plain torch restatement with the SAME mask."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def test_mask_statistics_and_determinism():
    _need_gpu()
    from vitxt_gqa_amd import ops
    n = 1000 * 768
    m1 = ops.dropout_mask(n, 0.1, 1234567890123, DEV)
    m2 = ops.dropout_mask(n, 0.1, 1234567890123, DEV)
    m3 = ops.dropout_mask(n, 0.1, 1234567890124, DEV)
    assert torch.equal(m1, m2) and not torch.equal(m1, m3)
    keep = m1.float().mean().item()
    assert abs(keep - 0.9) < 3e-3
    # no structure along rows / columns
    mm = m1.view(1000, 768).float()
    assert (mm.mean(0) - 0.9).abs().max().item() < 0.05 and (mm.mean(1) - 0.9).abs().max().item() < 0.06
    assert ops.dropout_mask(n, 0.0, 5, DEV).all()


@pytest.mark.parametrize("xdt", [torch.float32, torch.bfloat16])
def test_layernorm_with_dropout_matches_masked_reference(xdt):
    _need_gpu()
    from vitxt_gqa_amd import ops
    g = torch.Generator().manual_seed(0)
    rows, p, seed = 257, 0.25, 987654321
    x = torch.randn(rows, 768, generator=g).to(DEV).to(xdt)
    r = torch.randn(rows, 768, generator=g).to(DEV)
    gam = (1 + 0.1 * torch.randn(768, generator=g)).to(DEV)
    bet = (0.1 * torch.randn(768, generator=g)).to(DEV)
    dy = torch.randn(rows, 768, generator=g).to(DEV)
    keep = ops.dropout_mask(rows * 768, p, seed, DEV).view(rows, 768).double()
    xr, rr = x.double().requires_grad_(True), r.double().requires_grad_(True)
    z = xr * keep / (1 - p) + rr
    ref = (z - z.mean(-1, keepdim=True)) / torch.sqrt(z.var(-1, unbiased=False, keepdim=True) + 1e-12) * gam.double() + bet.double()
    y, _, zz, st = ops.add_layernorm_fwd(x.clone(), r, gam, bet, inplace_z=False, stream_dtype=torch.float32, drop_p=p, drop_seed=seed)
    assert (y.double() - ref).abs().max().item() < 2e-5
    dz, dzx, dg, db = ops.add_layernorm_bwd(dy, zz, st, gam, out_dtype=torch.float32, drop_p=p, drop_seed=seed)
    gx, gr = torch.autograd.grad(ref, (xr, rr), dy.double())
    assert (dz.double() - gr).abs().max().item() < 1e-4          # residual input: undropped
    assert (dzx.double() - gx).abs().max().item() < 1e-4         # branch input: dz * keep / (1 - p)
    assert (dzx[keep == 0] == 0).all()


def test_bert_layer_with_hidden_dropout():
    """Fused layer fwd + bwd with dropout == textbook BERT block with the exported masks applied at the same two sites."""
    _need_gpu()
    from oracle import t2s_oracle as O
    from vitxt_gqa_amd import functional as FN
    from vitxt_gqa_amd import ops
    from vitxt_gqa_amd.t2s import BertLayerParams
    torch.manual_seed(0)
    lp = BertLayerParams()
    for prm in lp.parameters():
        prm.data.normal_(0, 0.05)
    lp.attention.output.LayerNorm.weight.data.add_(1.0)
    lp.output.LayerNorm.weight.data.add_(1.0)
    B, L, p = 2, 70, 0.2
    x = torch.randn(B, L, 768)
    dy = torch.randn(B, L, 768)
    lp = lp.to(DEV)
    xg = x.to(DEV).requires_grad_(True)
    keys = ops.compact_keys(torch.ones(B, L, dtype=torch.bool, device=DEV))
    torch.manual_seed(77)
    y, _ = FN.bert_layer(xg, None, keys, lp, torch.float32, hidden_dropout=p)
    torch.manual_seed(77)                      # the two seeds the layer drew
    s1, s2 = FN._fresh_seed(), FN._fresh_seed()
    k1 = ops.dropout_mask(B * L * 768, p, s1, DEV).view(B, L, 768).double().cpu()
    k2 = ops.dropout_mask(B * L * 768, p, s2, DEV).view(B, L, 768).double().cpu()
    sd = {"l." + k: v.detach().double().cpu().requires_grad_(True) for k, v in lp.state_dict().items()}
    xr = x.double().requires_grad_(True)

    def layer(sd, x):      # oracle block with the two hidden dropouts
        pfx = "l."
        heads = lambda t: t.view(B, L, 12, 64).permute(0, 2, 1, 3)
        q = heads(O.linear(x, sd[pfx + "attention.self.query.weight"], sd[pfx + "attention.self.query.bias"]))
        k = heads(O.linear(x, sd[pfx + "attention.self.key.weight"], sd[pfx + "attention.self.key.bias"]))
        v = heads(O.linear(x, sd[pfx + "attention.self.value.weight"], sd[pfx + "attention.self.value.bias"]))
        ctx = (torch.softmax(q @ k.transpose(-1, -2) / 8.0, -1) @ v).permute(0, 2, 1, 3).reshape(B, L, 768)
        a = O.linear(ctx, sd[pfx + "attention.output.dense.weight"], sd[pfx + "attention.output.dense.bias"]) * k1 / (1 - p)
        a = O.layer_norm(a + x, sd[pfx + "attention.output.LayerNorm.weight"], sd[pfx + "attention.output.LayerNorm.bias"])
        i = O.gelu_erf(O.linear(a, sd[pfx + "intermediate.dense.weight"], sd[pfx + "intermediate.dense.bias"]))
        o = O.linear(i, sd[pfx + "output.dense.weight"], sd[pfx + "output.dense.bias"]) * k2 / (1 - p)
        return O.layer_norm(o + a, sd[pfx + "output.LayerNorm.weight"], sd[pfx + "output.LayerNorm.bias"])

    ref = layer(sd, xr)
    assert (y.double().cpu() - ref).abs().max().item() < 2e-4
    names = list(sd)
    grads = torch.autograd.grad(ref, [xr] + [sd[n] for n in names], dy.double())
    y.backward(dy.to(DEV))
    assert (xg.grad.double().cpu() - grads[0]).abs().max().item() < 1e-3 * max(1.0, grads[0].abs().max().item())
    got = dict(lp.named_parameters())
    for n, gr in zip(names, grads[1:]):
        if "key.bias" in n:
            continue
        a = got[n[2:]].grad.double().cpu()
        assert (a - gr).norm().item() / max(gr.norm().item(), 1e-9) < 5e-4, n
