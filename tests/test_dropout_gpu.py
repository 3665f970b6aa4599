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


# ---------------------------------------------------------------------------------------------------------------
# attention-probability dropout (BertSelfAttention: dropout(softmax(.)) before .V) inside the attention kernels
def _attn_ref(qkv, keep, p_eff, n_dec):
    """softmax(QK^T/8 + causal-tail mask) * keep / (1 - p') @ V with a dense key list."""
    B, L, _ = qkv.shape
    q, k, v = [t.view(B, L, 12, 64).permute(0, 2, 1, 3) for t in qkv.split(768, dim=-1)]
    s = (q @ k.transpose(-1, -2)) * 0.125
    vis = torch.ones(L, L, dtype=torch.bool, device=qkv.device)
    if n_dec:
        L1 = L - n_dec
        vis[:, L1:] = False
        vis[L1:, L1:] = torch.tril(torch.ones(n_dec, n_dec, dtype=torch.bool, device=qkv.device))
    s = s.masked_fill(~vis, float("-inf"))
    a = torch.softmax(s, -1) * keep / (1.0 - p_eff)
    return (a @ v).permute(0, 2, 1, 3).reshape(B, L, 768)


def test_attention_dropout_mask_statistics():
    _need_gpu()
    from vitxt_gqa_amd import ops
    m = ops.attn_dropout_mask(2, 300, 400, 0.1, 42, DEV).float()
    p_eff = 6554 / 65536          # round(65536 * 0.1): the 16-bit threshold of the kernels
    assert abs(m.mean().item() - (1 - p_eff)) < 2e-3
    assert torch.equal(m, ops.attn_dropout_mask(2, 300, 400, 0.1, 42, DEV).float())
    assert not torch.equal(m, ops.attn_dropout_mask(2, 300, 400, 0.1, 43, DEV).float())
    # per-row / per-column / per-head keep rates are flat, and neighbouring elements are uncorrelated
    assert (m.mean(-1) - (1 - p_eff)).abs().max().item() < 0.08
    assert (m.mean(-2) - (1 - p_eff)).abs().max().item() < 0.08
    assert (m.mean((0, 2, 3)) - (1 - p_eff)).abs().max().item() < 5e-3
    c = m - m.mean()
    for a, b in ((c[..., :-1], c[..., 1:]), (c[..., :-1, :], c[..., 1:, :]), (c[..., :-2], c[..., 2:]), (c[:, :-1], c[:, 1:])):
        assert abs((a * b).mean().item()) / c.var().item() < 0.01
    # The mask is t(q, k) = rowkey16(q, window(k)) * colkey16(k) mod 2^16 compared with a threshold: a product of per-row and
    # per-column keys, NOT i.i.d. Bernoulli like the reference's Philox draw (documented deviation).  What
    # that structure does and does not do, measured on a long sequence: the PAIRS of rows (and of columns) are uncorrelated - mean
    # pairwise correlation ~ 0, all but a fraction of a percent of the pairs inside 6.5 sigma of sampling noise.  Since round 4 the row
    # key changes with every 384-key window, so two rows whose keys collide in one window (2^-15 per pair and window) share 384 keys'
    # worth of mask, not all of it: NO identical row pairs any more (there were ~L^2 / 2^16 of them with one key per row).  The column
    # key changes with every 256-row window of queries in the same way: no identical column pairs either (9 at Lk = 2 048 before).
    Lq = Lk = 2048
    big = ops.attn_dropout_mask(1, Lq, Lk, 0.1, 7, DEV)[0, 0].float()          # one (sample, head): [Lq, Lk]
    z = (big - big.mean()) / big.std()
    corr = (z @ z.t()) / Lk                                                 # all row pairs
    off = corr - torch.diag(torch.diag(corr))
    same = int((off > 0.999).sum().item()) // 2                             # identical rows: none since the row key is per key window
    assert same == 0
    rest = off[off < 0.999]
    sig = 1.0 / Lk ** 0.5
    assert abs(rest.mean().item()) < 2e-3 and (rest.abs() > 6.5 * sig).float().mean().item() < 2e-3
    ccorr = (z.t() @ z) / Lq
    coff = ccorr - torch.diag(torch.diag(ccorr))
    crest = coff[coff < 0.999]
    assert abs(crest.mean().item()) < 2e-3 and (crest.abs() > 6.5 * sig).float().mean().item() < 2e-3
    csame = int((coff > 0.999).sum().item()) // 2                           # identical COLUMNS: none since the column key is per row window
    assert csame == 0
    print("dropout mask at L=2048: %d identical row pairs, %d identical column pairs; other row pairs: %.4f %% beyond 6.5 sigma, max |corr| %.3f; "
          "column pairs: %.4f %% beyond 6.5 sigma, max |corr| %.3f" % (same, csame, 100 * (rest.abs() > 6.5 * sig).float().mean().item(), rest.abs().max().item(),
                                                                         100 * (crest.abs() > 6.5 * sig).float().mean().item(), crest.abs().max().item()))
    del big, z, corr, off, rest, ccorr, coff, crest
    # The same at the BENCHMARK's length: L = 10 132 query rows x 10 132 key-list positions of one (sample, head).  With one
    # 16-bit key per row (rounds 1-3) this counted 1 098 identical row pairs - 1 872 of the 10 132 rows had a twin with the SAME mask over
    # all keys (measured in round 4 before the change; L^2 / 2^16 = 1 566 expected).  With the row key per 384-key window: no identical
    # pair, no pair correlated beyond 0.5; a pair that collides in one of its 27 windows correlates at 1 / 27 = 0.04, inside sampling
    # noise (6.5 sigma = 0.065), two windows at 0.07.
    Lb = 10132
    m = ops.attn_dropout_mask(1, Lb, Lb, 0.1, 11, DEV)[0, 0]                # uint8 [Lb, Lb]
    z = m.float()
    z = ((z - z.mean()) / z.std()).to(torch.bfloat16)                       # two values: exact enough in bf16 for counts
    corr = (z @ z.t()).float() / Lb
    corr.fill_diagonal_(0)
    same = int((corr > 0.9).sum().item()) // 2
    strong = int((corr.abs() > 0.5).sum().item()) // 2
    sig = 1.0 / Lb ** 0.5
    outl = (corr.abs() > 6.5 * sig).float().mean().item()
    assert same == 0 and strong == 0, (same, strong)
    assert outl < 2e-3
    ccorr = (z.t() @ z).float() / Lb                                        # the same over the 51 M column pairs
    ccorr.fill_diagonal_(0)
    csame, cstrong = int((ccorr > 0.9).sum().item()) // 2, int((ccorr.abs() > 0.5).sum().item()) // 2
    coutl = (ccorr.abs() > 6.5 * sig).float().mean().item()
    assert csame == 0 and cstrong == 0 and coutl < 2e-3, (csame, cstrong, coutl)
    print("dropout mask at L=10132: %d identical row pairs, %d with |correlation| > 0.5, max |correlation| %.3f; pairs beyond 6.5 sigma: %.4f %%; "
          "columns: %d identical, max |correlation| %.3f, beyond 6.5 sigma %.4f %%" % (same, strong, corr.abs().max().item(), 100 * outl, csame, ccorr.abs().max().item(), 100 * coutl))


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 3e-5), (torch.bfloat16, 3e-2)])
@pytest.mark.parametrize("L1,n_dec", [(150, 0), (300, 12)])
def test_attention_dropout_fwd_bwd(L1, n_dec, dtype, tol):
    _need_gpu()
    from vitxt_gqa_amd import ops
    g = torch.Generator().manual_seed(L1)
    B, L, p, seed = 2, L1 + n_dec, 0.1, 20241003
    qkv = torch.randn(B, L, 2304, generator=g).to(DEV).to(dtype)
    dout = torch.randn(B, L, 768, generator=g).to(DEV).to(dtype)
    keys = ops.compact_keys(torch.ones(B, L1, dtype=torch.bool, device=DEV), n_dec=n_dec, dec_row0=L1)   # dense list: position == row
    keep = ops.attn_dropout_mask(B, L, L, p, seed, DEV).double()
    out, lse = ops.attn_fwd(qkv, keys, drop_p=p, drop_seed=seed)
    xr = qkv.double().requires_grad_(True)
    ref = _attn_ref(xr, keep, 6554 / 65536, n_dec)
    assert (out.double() - ref).abs().max().item() < tol
    out0, lse0 = ops.attn_fwd(qkv, keys)                   # LSE is that of the undropped softmax
    assert (lse - lse0).abs().max().item() < 1e-5
    dqkv = ops.attn_bwd(qkv, out, dout, lse, keys, drop_p=p, drop_seed=seed)
    (gref,) = torch.autograd.grad(ref, xr, dout.double())
    sc = gref.abs().max().item()
    assert (dqkv.double() - gref).abs().max().item() < tol * max(1.0, sc) * (1 if dtype == torch.float32 else 2)
    if dtype == torch.bfloat16:          # the fused five-product backward regenerates the same mask (key on the lane, 96 keys per wave)
        fused = ops.attn_bwd(qkv, out, dout, lse, keys, drop_p=p, drop_seed=seed, fused=True)
        assert (fused.double() - gref).abs().max().item() < tol * max(1.0, sc) * 2


def test_attention_dropout_fused_backward_at_length():
    """Fused backward with dropout at a length where every code path runs (full 384-key blocks through the software pipeline, an
    edge block with decoder keys): against the two-kernel form with the same seed."""
    _need_gpu()
    from vitxt_gqa_amd import ops
    g = torch.Generator().manual_seed(9)
    B, L1, n_dec, p, seed = 2, 1300, 12, 0.1, 77
    L = L1 + n_dec
    qkv = (torch.randn(B, L, 2304, generator=g) * 1.2).to(DEV).to(torch.bfloat16)
    dout = torch.randn(B, L, 768, generator=g).to(DEV).to(torch.bfloat16)
    valid = (torch.rand(B, L1, generator=g) < 0.8).to(DEV)
    keys = ops.compact_keys(valid, n_dec=n_dec, dec_row0=L1)
    out, lse = ops.attn_fwd(qkv, keys, drop_p=p, drop_seed=seed)
    two = ops.attn_bwd(qkv, out, dout, lse, keys, drop_p=p, drop_seed=seed, fused=False)
    fus = ops.attn_bwd(qkv, out, dout, lse, keys, drop_p=p, drop_seed=seed, fused=True)
    sc = two.float().abs().max().item()
    assert (fus.float() - two.float()).abs().max().item() < 3e-2 * max(1.0, sc)
    other = ops.attn_bwd(qkv, out, dout, lse, keys, drop_p=p, drop_seed=seed + 1, fused=True)        # another seed: another mask
    assert (other.float() - two.float()).abs().max().item() > 0.1 * sc


def test_model_trains_with_dropout():
    """One train step with the reference's default dropout 0.1 everywhere: finite loss, gradients for exactly the live
    parameters, and a different loss than the dropout-free step."""
    _need_gpu()
    from vitxt_gqa_amd.synth import make_batch, make_noise
    from vitxt_gqa_amd.testing import make_model, to_device
    B, F, P, V = 2, 6, 8, 64
    batch = to_device(make_batch(B, F, P, V=V, seed=5, text_vocab=100), DEV)
    batch.grounding_noise = tuple(t.to(DEV) for t in make_noise(B, F, P, 5))
    losses = []
    for drop in (0.0, 0.1):
        torch.manual_seed(1)
        m = make_model(F, P, V, text_vocab=100, dtype=torch.bfloat16, dropout=drop).to(DEV).train()
        out = m(batch)
        loss = sum(v.mean() for v in out["losses"].values())
        loss.backward()
        assert torch.isfinite(loss)
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters() if p.requires_grad)
        losses.append(loss.item())
    assert abs(losses[0] - losses[1]) > 1e-4 * abs(losses[0])
