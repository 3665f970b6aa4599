"""GPU parity tests of the individual HIP kernels (called through the C ABI via vitxt_gqa_amd.ops)
against plain fp32/fp64 torch restatements of the same op.  Run on the MI355X box: pytest -m gpu."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def dense_mask(valid, n_dec, L):
    """[B, L, L] boolean visibility equal to the reference's extended mask (t2s.py:413-419, 609-618):
    prefix keys by `valid`, decoder key j visible to decoder row i iff i >= j."""
    B, L1 = valid.shape
    assert L1 + n_dec == L
    m = torch.zeros(B, L, L, dtype=torch.bool, device=valid.device)
    m[:, :, :L1] = valid.bool().unsqueeze(1)
    if n_dec:
        m[:, L1:, L1:] = torch.tril(torch.ones(n_dec, n_dec, dtype=torch.bool, device=valid.device))
    return m


def ref_attention(qkv, mask, scale):
    B, L, _ = qkv.shape
    q, k, v = [t.view(B, L, 12, 64).permute(0, 2, 1, 3) for t in qkv.split(768, dim=-1)]
    s = (q @ k.transpose(-1, -2)) * scale + (~mask).unsqueeze(1).to(q.dtype) * -10000.0
    p = torch.softmax(s, dim=-1)
    return (p @ v).permute(0, 2, 1, 3).reshape(B, L, 768)


CASES = [  # B, L1 (prefix rows), n_dec, keep probability
    (2, 20, 0, 0.6),
    (2, 77, 0, 0.7),
    (3, 200, 12, 0.7),
    (2, 300, 12, 0.05),
    (1, 515, 12, 1.0),
]


# bf16 tolerance: |out| reaches ~4 and the scores ~12 on this data (std-1.5 inputs); operand roundings of 2^-9 (Q*scale*log2e,
# P, the output itself) each contribute up to ~1e-2 absolute
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 3e-2)])
@pytest.mark.parametrize("B,L1,n_dec,keep", CASES)
def test_attention_fwd_bwd(B, L1, n_dec, keep, dtype, tol):
    _need_gpu()
    from vitxt_gqa_amd import ops
    g = torch.Generator(device="cpu").manual_seed(B * 1000 + L1)
    L = L1 + n_dec
    qkv = (torch.randn(B, L, 2304, generator=g) * 1.5).to(DEV)
    valid = (torch.rand(B, L1, generator=g) < keep).to(DEV)
    valid[:, 0] = True
    dout = torch.randn(B, L, 768, generator=g).to(DEV)
    keys = ops.compact_keys(valid, n_dec=n_dec, dec_row0=L1)
    # key list content
    cnt = valid.sum(1).to(torch.int32)
    assert torch.equal(keys.cnt, cnt)
    for b in range(B):
        exp = torch.nonzero(valid[b]).flatten().to(torch.int32)
        assert torch.equal(keys.idx[b, :cnt[b]], exp)
        if n_dec:
            assert torch.equal(keys.idx[b, cnt[b]:cnt[b] + n_dec], torch.arange(L1, L, device=DEV, dtype=torch.int32))

    x = qkv.to(dtype)
    out, lse = ops.attn_fwd(x, keys)
    xr = x.double().requires_grad_(True)
    ref = ref_attention(xr, dense_mask(valid, n_dec, L), 0.125)
    err = (out.double() - ref).abs().max().item()
    assert err < tol, "fwd max err %.3e" % err
    # log-sum-exp
    q, k, _ = [t.view(B, L, 12, 64).permute(0, 2, 1, 3) for t in xr.detach().split(768, dim=-1)]
    s = (q @ k.transpose(-1, -2)) * 0.125
    s = s.masked_fill(~dense_mask(valid, n_dec, L).unsqueeze(1), float("-inf"))
    # bf16: the kernel folds scale*log2(e) into Q with one more bf16 rounding (2^-9 relative on scores of |s| ~ 10)
    assert (lse.double() - torch.logsumexp(s, -1)).abs().max().item() < (1e-4 if dtype == torch.float32 else 4e-2)

    dqkv = ops.attn_bwd(x, out, dout.to(dtype), lse, keys)
    (gref,) = torch.autograd.grad(ref, xr, dout.to(dtype).double())
    scale = gref.abs().max().item()
    gerr = (dqkv.double() - gref).abs().max().item()
    assert gerr < tol * max(1.0, scale) * (1 if dtype == torch.float32 else 2), "bwd max err %.3e (scale %.3e)" % (gerr, scale)
    # masked keys get exactly zero dK / dV
    kvalid = torch.cat([valid, torch.ones(B, n_dec, dtype=torch.bool, device=DEV)], 1)
    assert dqkv[..., 768:][~kvalid].abs().max().item() == 0 if (~kvalid).any() else True


def test_attention_spiky_rows_force_rescale():
    """Online-softmax rescale path: one key far above the rest, placed in a late tile (guide rule 26)."""
    _need_gpu()
    from vitxt_gqa_amd import ops
    B, L = 1, 400
    g = torch.Generator().manual_seed(5)
    qkv = torch.randn(B, L, 2304, generator=g).to(DEV)
    qkv[:, 350, 768:1536] = qkv[:, 3, :768] * 6.0       # key 350 aligned with query 3 (all heads): score jumps late
    keys = ops.compact_keys(torch.ones(B, L, dtype=torch.bool, device=DEV))
    for dtype, tol in ((torch.float32, 2e-5), (torch.bfloat16, 3e-2)):
        x = qkv.to(dtype)
        out, _ = ops.attn_fwd(x, keys)
        ref = ref_attention(x.double(), torch.ones(B, L, L, dtype=torch.bool, device=DEV), 0.125)
        assert (out.double() - ref).abs().max().item() < tol


def test_attention_fwd_repair_launch():
    """The bf16 steady-state tile has no running maximum: a score more than 2^80 above the row's reference (set by key
    tile 0) poisons the wave and the repair launch redoes its workgroup through the general path.  Keys of the first
    tile score ~ -100 for query 3, a late key scores ~ +100 (log2 units ~ 290 apart): exact result required, and the
    rows of every other workgroup must be untouched by the repair."""
    _need_gpu()
    from vitxt_gqa_amd import ops
    B, L = 1, 900                                            # > 256 rows: 64 rows per wave, 14 whole key tiles
    g = torch.Generator().manual_seed(11)
    qkv = torch.randn(B, L, 2304, generator=g).to(DEV)
    q3 = qkv[:, 3, :768].clone()
    qkv[:, :64, 768:1536] = -q3.unsqueeze(1) * (100.0 * 8 / 64) / (q3.view(12, 64).pow(2).mean()).item()
    qkv[:, 700, 768:1536] = q3 * (100.0 * 8 / 64) / (q3.view(12, 64).pow(2).mean()).item()
    keys = ops.compact_keys(torch.ones(B, L, dtype=torch.bool, device=DEV))
    x = qkv.to(torch.bfloat16)
    out, lse = ops.attn_fwd(x, keys)
    assert torch.isfinite(out.float()).all() and torch.isfinite(lse).all()
    ref = ref_attention(x.double(), torch.ones(B, L, L, dtype=torch.bool, device=DEV), 0.125)
    err = (out.double() - ref).abs()
    # scores reach +-100 here: a 2^-9 operand rounding moves them by ~0.2, i.e. near-tied probabilities by ~20 %
    assert err.max().item() < 0.15, err.max().item()
    # query 3 attends (all heads) to key 700 alone
    v700 = x[:, 700, 1536:].double()
    assert (out[:, 3].double() - v700).abs().max().item() < 5e-2


@pytest.mark.parametrize("xdt,sdt,tol", [(torch.float32, torch.float32, 2e-5), (torch.bfloat16, torch.float32, 2e-5),
                                         (torch.bfloat16, torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("rows", [1, 7, 1000])
def test_add_layernorm(rows, xdt, sdt, tol):
    """x: GEMM-output dtype, residual stream (res / y / z) dtype sdt; checks y, y_lo, z, stats and backward."""
    _need_gpu()
    from vitxt_gqa_amd import ops
    g = torch.Generator().manual_seed(rows)
    x = torch.randn(rows, 768, generator=g).to(DEV).to(xdt)
    r = torch.randn(rows, 768, generator=g).to(DEV).to(sdt)
    gam = (1 + 0.1 * torch.randn(768, generator=g)).to(DEV)
    bet = (0.1 * torch.randn(768, generator=g)).to(DEV)
    dy = torch.randn(rows, 768, generator=g).to(DEV)
    xr, rr = x.double().requires_grad_(True), r.double().requires_grad_(True)
    gr, br = gam.double().requires_grad_(True), bet.double().requires_grad_(True)
    z = xr + rr
    ref = (z - z.mean(-1, keepdim=True)) / torch.sqrt(z.var(-1, unbiased=False, keepdim=True) + 1e-12) * gr + br
    y, y_lo, zz, st = ops.add_layernorm_fwd(x.clone(), r, gam, bet, inplace_z=False, stream_dtype=sdt, want_lo=True)
    assert y.dtype == sdt and zz.dtype == sdt and y_lo.dtype == torch.bfloat16
    assert (y.double() - ref).abs().max().item() < tol
    assert (y_lo.double() - ref).abs().max().item() < 2e-2
    assert (zz.double() - z).abs().max().item() < tol
    assert (st[:, 0].double() - z.mean(-1)).abs().max().item() < 1e-5
    for ddt, odt in ((torch.float32, torch.float32), (torch.float32, torch.bfloat16), (torch.bfloat16, torch.bfloat16)):
        if sdt == torch.bfloat16 and not (ddt == torch.bfloat16 and odt == torch.bfloat16):
            continue
        if sdt == torch.float32 and ddt == torch.bfloat16 and odt == torch.float32:
            continue
        d = dy.to(ddt)
        dz, dzx, dg, db = ops.add_layernorm_bwd(d, zz, st, gam, out_dtype=odt)
        assert dzx is dz                                   # no dropout: one gradient for branch and residual
        gx, gg, gb = torch.autograd.grad(ref, (xr, gr, br), d.double(), retain_graph=True)
        t2 = tol if odt == torch.float32 else 2e-2
        assert (dz.double() - gx).abs().max().item() < t2 * 5
        assert (dg.double() - gg).abs().max().item() < max(tol, 1e-4) * 5 * math.sqrt(rows)
        assert (db.double() - gb).abs().max().item() < max(tol, 1e-4) * 5 * math.sqrt(rows)
    # without residual; in place when dtypes match; y_lo only
    y2, _, z2, _ = ops.add_layernorm_fwd(x.clone(), None, gam, bet, stream_dtype=sdt)
    zr = x.double()
    ref2 = (zr - zr.mean(-1, keepdim=True)) / torch.sqrt(zr.var(-1, unbiased=False, keepdim=True) + 1e-12) * gam.double() + bet.double()
    assert (y2.double() - ref2).abs().max().item() < tol
    y3, y3_lo, _, _ = ops.add_layernorm_fwd(x.clone(), None, gam, bet, save=False, stream_dtype=sdt, want_lo=True, want_y=False)
    assert y3 is None and (y3_lo.double() - ref2).abs().max().item() < 2e-2


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 2e-2)])
def test_gelu(dtype, tol):
    _need_gpu()
    from vitxt_gqa_amd import ops
    g = torch.Generator().manual_seed(1)
    u = (torch.randn(333, 3072, generator=g) * 2).to(DEV).to(dtype)
    dy = torch.randn(333, 3072, generator=g).to(DEV).to(dtype)
    ur = u.double().requires_grad_(True)
    ref = ur * 0.5 * (1 + torch.erf(ur / math.sqrt(2)))
    y = ops.gelu_fwd(u)
    assert (y.double() - ref).abs().max().item() < tol
    du, db = ops.gelu_bwd(dy, u)
    (gu,) = torch.autograd.grad(ref, ur, dy.double())
    assert (du.double() - gu).abs().max().item() < tol * 4
    assert (db.double() - gu.sum(0)).abs().max().item() < tol * 40


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-4), (torch.bfloat16, 5e-2)])
def test_bert_layer_fn(dtype, tol):
    """Fused BertLayer autograd function vs the oracle's restatement of the third-party BERT block."""
    _need_gpu()
    from oracle import t2s_oracle as O
    from vitxt_gqa_amd import functional as FN
    from vitxt_gqa_amd import ops
    from vitxt_gqa_amd.t2s import BertLayerParams
    torch.manual_seed(0)
    lp = BertLayerParams()
    for p in lp.parameters():
        p.data.normal_(0, 0.05)
    lp.attention.output.LayerNorm.weight.data.add_(1.0)
    lp.output.LayerNorm.weight.data.add_(1.0)
    B, L1, n_dec = 2, 90, 12
    L = L1 + n_dec
    x = torch.randn(B, L, 768)
    valid = torch.rand(B, L1) < 0.6
    valid[:, 0] = True
    dy = torch.randn(B, L, 768)
    # oracle (fp64, CPU)
    sd = {"l." + k: v.detach().double().requires_grad_(True) for k, v in lp.state_dict().items()}
    xr = x.to(dtype).double().requires_grad_(True)
    ext = (~dense_mask(valid, n_dec, L)).double().unsqueeze(1) * -10000.0
    ref = O.bert_layer(sd, "l.", xr, ext)
    names = list(sd)
    grads = torch.autograd.grad(ref, [xr] + [sd[n] for n in names], dy.to(dtype).double())
    # HIP path
    lp = lp.to(DEV)
    xg = x.to(DEV).to(dtype).float().requires_grad_(True)        # fp32 residual stream holding operand-exact values
    keys = ops.compact_keys(valid.to(DEV), n_dec=n_dec, dec_row0=L1)
    y, y_lo = FN.bert_layer(xg, None, keys, lp, dtype)
    assert y.dtype == torch.float32 and y_lo.dtype == dtype
    assert (y.double().cpu() - ref).abs().max().item() < tol
    y.backward(dy.to(DEV).to(dtype).float())
    assert (xg.grad.double().cpu() - grads[0]).abs().max().item() < tol * max(1.0, grads[0].abs().max().item()) * 4
    got = dict(lp.named_parameters())
    for n, gr in zip(names, grads[1:]):
        a = got[n[2:]].grad.double().cpu()
        rel = (a - gr).norm().item() / max(gr.norm().item(), 1e-6 * gr.numel() ** 0.5)
        if "key.bias" in n:        # mathematically zero gradient: only rounding noise of the dK column sums
            assert a.abs().max().item() < (1e-3 if dtype == torch.float32 else 0.5)
            continue
        assert rel < (2e-4 if dtype == torch.float32 else 4e-2), "%s rel err %.3e" % (n, rel)


@pytest.mark.parametrize("rows,drop_p", [(5, 0.0), (1030, 0.0), (1030, 0.1)])
def test_add_layernorm_normalised_residual_and_bias_sums(rows, drop_p):
    """t2s_add_layernorm_fwd_nres: the residual handed over as (z, stats, gamma, beta) of the previous block gives the
    same output as handing over that block's materialised fp32 output; t2s_add_layernorm_bwd_bias: the extra partial sums
    are the column sums of the branch-input gradient."""
    _need_gpu()
    from vitxt_gqa_amd import ops
    torch.manual_seed(rows)
    x0 = torch.randn(rows, 768, device=DEV).to(torch.bfloat16)
    r0 = torch.randn(rows, 768, device=DEV)
    g0, b0 = torch.rand(768, device=DEV) + 0.5, torch.randn(768, device=DEV) * 0.1
    y0, _, z0, st0 = ops.add_layernorm_fwd(x0, r0, g0, b0, stream_dtype=torch.float32, want_lo=True)
    x1 = torch.randn(rows, 768, device=DEV).to(torch.bfloat16)
    g1, b1 = torch.rand(768, device=DEV) + 0.5, torch.randn(768, device=DEV) * 0.1
    kw = dict(stream_dtype=torch.float32, want_lo=True, drop_p=drop_p, drop_seed=99)
    ya, ya_lo, za, sta = ops.add_layernorm_fwd(x1.clone(), y0, g1, b1, **kw)
    yb, yb_lo, zb, stb = ops.add_layernorm_fwd(x1.clone(), ops.NormRes(z0, st0, g0, b0), g1, b1, **kw)
    assert (ya - yb).abs().max().item() < 2e-6 and (za - zb).abs().max().item() < 2e-6 and (sta - stb).abs().max().item() < 1e-5
    assert torch.equal(ya_lo, yb_lo) or (ya_lo.float() - yb_lo.float()).abs().max().item() < 0.04       # a bf16 ulp at |y| ~ 4
    yc, yc_lo, _, _ = ops.add_layernorm_fwd(x1.clone(), ops.NormRes(z0, st0, g0, b0), g1, b1, want_y=False, **kw)
    assert yc is None and torch.equal(yc_lo, yb_lo)
    dy = torch.randn(rows, 768, device=DEV).to(torch.bfloat16)
    dz, dzx, dg, db = ops.add_layernorm_bwd(dy, za, sta, g1, out_dtype=torch.bfloat16, drop_p=drop_p, drop_seed=99)
    dz2, dzx2, dg2, db2, dbias = ops.add_layernorm_bwd(dy, za, sta, g1, out_dtype=torch.bfloat16, drop_p=drop_p, drop_seed=99, want_bias=True)
    assert torch.equal(dz, dz2) and torch.equal(dzx, dzx2) and torch.allclose(dg, dg2) and torch.allclose(db, db2)
    ref = dzx.double().sum(0)
    assert (dbias.double() - ref).abs().max().item() < 2e-3 * (1 + ref.abs().max().item()) + 4e-3 * rows ** 0.5   # ref sums bf16-ROUNDED values


@pytest.mark.parametrize("keep,drop_p", [(0.6, 0.0), (0.05, 0.1), (1.0, 0.0)])
def test_attention_bwd_fills_unlisted_rows_in_kernel(keep, drop_p):
    """t2s_attn_bwd_fill (the dQ kernel writes the zero dK / dV rows of keys that are in no list) against t2s_attn_bwd on a
    zero-filled buffer: bit-equal, and no element of the uninitialised buffer survives."""
    _need_gpu()
    from vitxt_gqa_amd import ops
    torch.manual_seed(3)
    B, L1, n_dec = 3, 700, 12
    L = L1 + n_dec
    qkv = torch.randn(B, L, 2304, device=DEV).to(torch.bfloat16)
    dout = torch.randn(B, L, 768, device=DEV).to(torch.bfloat16)
    valid = torch.rand(B, L1, device=DEV) < keep
    valid[:, 5] = True
    keys = ops.compact_keys(valid, n_dec=n_dec, dec_row0=L1)
    assert keys.valid8 is not None
    kw = dict(drop_p=drop_p, drop_seed=17) if drop_p else {}
    out, lse = ops.attn_fwd(qkv, keys, **kw)
    # poison the allocator's free list so that a missed row shows up as NaN
    junk = torch.full((B, L, 2304), float("nan"), device=DEV, dtype=torch.bfloat16)
    del junk
    a = ops.attn_bwd(qkv, out, dout, lse, keys, **kw)
    plain = ops.KeyList(keys.idx, keys.cnt, keys.n_dec, keys.dec_q0, keys.cap_hint, None)
    b = ops.attn_bwd(qkv, out, dout, lse, plain, **kw)
    assert torch.isfinite(a.float()).all()
    assert torch.equal(a, b)
    unlisted = ~valid
    assert a[:, :L1, 768:][unlisted].abs().max().item() == 0.0 if unlisted.any() else True


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("which", [0, 1, 2])
def test_attention_shared_prefix_layout_equals_own_sequence(which, fused):
    """The layout of functional.SharedPrefixEncoderFn: one sequence [prefix | dec rows of pass 0 | pass 1 | pass 2], a call lists
    only ITS decoder rows as keys (dec_row0 = L1 + which * D).  For the prefix rows and the call's own decoder rows, forward
    output and all three gradients must equal those of the call's own [prefix | dec] sequence; the other passes' decoder rows
    get exact-zero dK / dV (written in-kernel: nothing of the uninitialised buffer survives) and, with a zero output
    gradient, exact-zero dQ."""
    _need_gpu()
    from vitxt_gqa_amd import ops
    torch.manual_seed(11 + which)
    B, L1, D = 2, 1100, 12
    L = L1 + 3 * D
    qkv = torch.randn(B, L, 2304, device=DEV).to(torch.bfloat16)
    dout = torch.randn(B, L, 768, device=DEV).to(torch.bfloat16)
    own = slice(L1 + which * D, L1 + (which + 1) * D)
    others = torch.ones(L, dtype=torch.bool, device=DEV)
    others[:L1] = False
    others[own] = False
    dout[:, others] = 0                                  # nothing reads the other passes' decoder rows of this call's output
    valid = torch.rand(B, L1, device=DEV) < 0.6
    valid[:, 3] = True
    keys = ops.compact_keys(valid, n_dec=D, dec_row0=L1 + which * D)
    out, lse = ops.attn_fwd(qkv, keys)
    junk = torch.full((B, L, 2304), float("nan"), device=DEV, dtype=torch.bfloat16)
    del junk
    g = ops.attn_bwd(qkv, out, dout, lse, keys, fused=fused)
    assert torch.isfinite(g.float()).all()
    # the same call on its own sequence
    rows = torch.cat([torch.arange(L1, device=DEV), torch.arange(own.start, own.stop, device=DEV)])
    qkv1, dout1 = qkv[:, rows].contiguous(), dout[:, rows].contiguous()
    keys1 = ops.compact_keys(valid, n_dec=D, dec_row0=L1)
    out1, lse1 = ops.attn_fwd(qkv1, keys1)
    g1 = ops.attn_bwd(qkv1, out1, dout1, lse1, keys1, fused=fused)
    assert torch.equal(out[:, rows], out1) and torch.equal(lse[:, :, rows], lse1)
    if not fused and which == 0:                         # same rows in the same tiles: the same arithmetic
        assert torch.equal(g[:, rows], g1)
    else:       # the decoder rows sit in another query tile (dK / dV sum the tiles in order), fused: dQ is summed with float atomics -
        for third in range(3):                           # equal up to the fp32 summation order, i.e. to a bf16 ulp of the result
            a, b = g[:, rows, 768 * third:768 * (third + 1)].float(), g1[:, :, 768 * third:768 * (third + 1)].float()
            assert (a - b).abs().max().item() <= 1e-2 * b.abs().max().item(), third
            assert (a - b).norm().item() <= 2e-3 * b.norm().item(), third
    assert g[:, others].abs().max().item() == 0.0
    assert g[:, :L1, 768:][~valid].abs().max().item() == 0.0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("drop_p", [0.0, 0.1])
def test_attention_bwd_survives_a_wrong_static_key_bound(dtype, drop_p):
    """``cap_hint`` (the static bound on visible keys the model passes for the pos / neg passes, t2s.py `_three_pass`) only sizes
    the dK/dV grid: when the real key count exceeds it - injected grounding masks, another temporal_id layout - the gradients
    must still be those of the full key list (the tail launch / the last block's loop walks the remaining key blocks)."""
    _need_gpu()
    from vitxt_gqa_amd import ops
    if dtype == torch.float32 and drop_p:
        pytest.skip("covered by the bf16 case")
    B, L1, n_dec = 2, 700, 12
    L = L1 + n_dec
    g = torch.Generator().manual_seed(17)
    x = (torch.randn(B, L, 2304, generator=g) * 0.7).to(DEV).to(dtype)
    dout = torch.randn(B, L, 768, generator=g).to(DEV).to(dtype)
    valid = (torch.rand(B, L1, generator=g) < 0.8).to(DEV)                   # ~570 visible keys per sample
    full = ops.compact_keys(valid, n_dec=n_dec, dec_row0=L1)
    kw = dict(drop_p=drop_p, drop_seed=99) if drop_p else {}
    out, lse = ops.attn_fwd(x, full, **kw)
    want = ops.attn_bwd(x, out, dout, lse, full, **kw)
    for hint in (40, 130, 300):                                            # far below the real count: 1, 2, 3 key blocks
        wrong = ops.compact_keys(valid, n_dec=n_dec, dec_row0=L1, cap_hint=hint)
        assert wrong.cap_hint == hint
        got = ops.attn_bwd(x, out, dout, lse, wrong, **kw)
        assert torch.equal(got, want), hint
        nofill = ops.KeyList(wrong.idx, wrong.cnt, wrong.n_dec, wrong.dec_q0, hint, None)      # zero-fill path
        assert torch.equal(ops.attn_bwd(x, out, dout, lse, nofill, **kw), want), hint


def test_pruned_kv_path_with_a_violated_structural_bound_stays_in_bounds():
    """The pruned K / V projection (functional._layer_forward, KeyList.compact) trusts a caller-vouched bound on the visible keys.
    With the bound VIOLATED (sample 1 has ~240 keys against a vouched 100) the compact list is clamped to its [B, capK] buffer:
    the launch must not fault, every output must be finite, the sample that respects the bound must come out exactly as through
    the un-pruned path, and the violating sample must equal the attention over its first capK - n_dec listed keys."""
    _need_gpu()
    from vitxt_gqa_amd import ops
    B, L1, n_dec = 2, 300, 12
    L = L1 + n_dec
    g = torch.Generator().manual_seed(31)
    x = (torch.randn(B, L, 2304, generator=g) * 0.7).to(DEV).to(torch.bfloat16)
    dout = torch.randn(B, L, 768, generator=g).to(DEV).to(torch.bfloat16)
    valid = torch.zeros(B, L1, dtype=torch.bool)
    valid[0, torch.randperm(L1, generator=g)[:60]] = True
    valid[1, torch.randperm(L1, generator=g)[:240]] = True
    valid = valid.to(DEV)
    keys = ops.compact_keys(valid, n_dec=n_dec, dec_row0=L1, cap_hint=100 + n_dec)
    keys.bound_is_structural = True
    keys_c, flat, capK = keys.compact(L)
    assert capK == 128 and int((keys_c.cnt + n_dec).max()) <= capK
    q = x[..., :768].contiguous()
    kv = x.view(B * L, 2304).index_select(0, flat)[:, 768:].contiguous().view(B, capK, 1536)
    out, lse = ops.attn_fwd(q, keys_c, kv=kv)
    dq, dkv = ops.attn_bwd(q, out, dout, lse, keys_c, kv=kv)
    torch.cuda.synchronize()
    for t in (out, lse, dq, dkv):
        assert torch.isfinite(t.float()).all()
    # reference: the un-pruned path over the list each sample effectively has (sample 1: its first capK - n_dec keys)
    eff = valid.clone()
    rows1 = torch.nonzero(valid[1]).flatten()
    eff[1] = False
    eff[1, rows1[:capK - n_dec]] = True
    full = ops.compact_keys(eff, n_dec=n_dec, dec_row0=L1)
    want, wlse = ops.attn_fwd(x, full)
    assert (out[0].float() - want[0].float()).abs().max().item() < 1e-6 and (lse[0] - wlse[0]).abs().max().item() < 1e-6
    assert (out[1].float() - want[1].float()).abs().max().item() < 2e-2 and (lse[1] - wlse[1]).abs().max().item() < 2e-2


@pytest.mark.parametrize("drop_p", [0.0, 0.1])
def test_pruned_kv_backward_takes_the_fused_form(drop_p):
    """The pos / neg MMT passes hand the backward a COMPACT K | V buffer (only the rows that are keys: KeyList.compact) and a short
    list (<= 549 / 74 keys for 10 132 query rows).  With the ordered hand-off the fused five-product kernel is the faster form there
    too (profiles/r04_light_launches.txt), so ops.attn_bwd routes that call through it: same gradients as the two-kernel form
    (bf16 tolerance; dK / dV rows behind a sample's list exactly zero), bit-reproducible, and equal to the un-pruned self-attention
    path's gradients gathered at the key rows."""
    _need_gpu()
    from vitxt_gqa_amd import ops
    B, L1, n_dec = 2, 1500, 12
    L = L1 + n_dec
    g = torch.Generator().manual_seed(77)
    x = (torch.randn(B, L, 2304, generator=g) * 0.7).to(DEV).to(torch.bfloat16)
    dout = torch.randn(B, L, 768, generator=g).to(DEV).to(torch.bfloat16)
    valid = torch.zeros(B, L1, dtype=torch.bool)
    valid[0, torch.randperm(L1, generator=g)[:70]] = True
    valid[1, torch.randperm(L1, generator=g)[:520]] = True
    valid = valid.to(DEV)
    keys = ops.compact_keys(valid, n_dec=n_dec, dec_row0=L1, cap_hint=537 + n_dec)
    keys.bound_is_structural = True
    keys_c, flat, capK = keys.compact(L)
    q = x[..., :768].contiguous()
    kv = x.view(B * L, 2304).index_select(0, flat)[:, 768:].contiguous().view(B, capK, 1536)
    kw = dict(drop_p=drop_p, drop_seed=99) if drop_p else {}
    out, lse = ops.attn_fwd(q, keys_c, kv=kv, **kw)
    assert ops._fused_policy(None, q, keys_c, L, 1) and not ops._fused_policy(None, q, keys_c, 20, 1) and not ops._fused_policy(None, q, keys_c, L, 0)
    dq_f, dkv_f = ops.attn_bwd(q, out, dout, lse, keys_c, kv=kv, **kw)                    # default policy: fused, hand-off
    assert ops.LAST_ATTN_BWD_PRODUCTS == 5 and ops.fused_handoff_status() == 0
    dq_g, dkv_g = ops.attn_bwd(q, out, dout, lse, keys_c, kv=kv, **kw)
    assert torch.equal(dq_f, dq_g) and torch.equal(dkv_f, dkv_g)
    dq_t, dkv_t = ops.attn_bwd(q, out, dout, lse, keys_c, kv=kv, fused=False, **kw)
    assert ops.LAST_ATTN_BWD_PRODUCTS == 7
    sq, sk = dq_t.float().abs().max().item(), dkv_t.float().abs().max().item()
    assert (dq_f.float() - dq_t.float()).abs().max().item() < 3e-2 * max(1.0, sq)
    assert (dkv_f.float() - dkv_t.float()).abs().max().item() < 2e-2 * max(1.0, sk)
    live = torch.arange(capK, device=DEV).unsqueeze(0) < (keys_c.cnt + n_dec).unsqueeze(1)
    assert dkv_f[~live].abs().max().item() == 0
    # the un-pruned path (same list over the fused QKV buffer): dQ the same, dK / dV of the listed rows the same rows of its output
    out_u, lse_u = ops.attn_fwd(x, keys, **kw)
    assert (out_u.float() - out.float()).abs().max().item() < 1e-6
    full = ops.attn_bwd(x, out_u, dout, lse_u, keys, fused=True, **kw)
    assert (full[..., :768].float() - dq_f.float()).abs().max().item() < 3e-2 * max(1.0, sq)
    gathered = full.view(B * L, 2304).index_select(0, flat)[:, 768:].view(B, capK, 1536)
    assert (gathered[live].float() - dkv_f[live].float()).abs().max().item() < 2e-2 * max(1.0, sk)


def test_attention_entry_points_refuse_bad_arguments():
    """Error behaviour of the attention entry points (return code + t2s_last_error, no launch): a fused-backward workspace smaller
    than t2s_attn_bwd_fused_workspace_bytes says how many bytes are needed; K / V rows that would not fit the kernels' 32-bit
    offsets (row stride or row count >= 2^24, span >= 4 GB) are refused before anything is launched; kv_idx without kv_cnt is."""
    _need_gpu()
    from vitxt_gqa_amd import hipext as X
    from vitxt_gqa_amd import ops
    B, L = 1, 256
    x = torch.randn(B, L, 2304, device=DEV).to(torch.bfloat16)
    valid = torch.ones(B, L, dtype=torch.bool, device=DEV)
    keys = ops.compact_keys(valid)
    out, lse = ops.attn_fwd(x, keys)
    q, k, v = x[..., :768], x[..., 768:1536], x[..., 1536:]
    dq = torch.empty_like(x)
    delta = torch.empty_like(lse)
    ws = torch.empty(1024, dtype=torch.float32, device=DEV)
    need = int(X.lib().t2s_attn_bwd_fused_workspace_bytes(B, 12, L))
    assert need > ws.numel() * 4
    rc = X.lib().t2s_attn_bwd_fused(X.ptr(q), X.ptr(k), X.ptr(v), X.ptr(out), X.ptr(out), X.ptr(lse), X.ptr(delta),
                                    X.ptr(dq[..., :768]), X.ptr(dq[..., 768:1536]), X.ptr(dq[..., 1536:]), X.ptr(ws), ws.numel() * 4, 1,
                                    X.ptr(keys.idx), X.ptr(keys.cnt), None, B, 12, L, keys.idx.shape[1], 0, 0, keys.cap_hint,
                                    x.stride(1), x.stride(0), x.stride(1), x.stride(0), out.stride(1), out.stride(0),
                                    0.125, X.dtype_code(x), 0.0, 0, X.stream())
    assert rc != 0
    with pytest.raises(RuntimeError, match="workspace of 4096 bytes, %d needed" % need):
        X.check(rc, "t2s_attn_bwd_fused")

    def fwd(kv_rs, idx=keys.idx, cnt=keys.cnt):
        o2, l2 = torch.empty_like(out), torch.empty_like(lse)
        return X.lib().t2s_attn_fwd(X.ptr(q), X.ptr(k), X.ptr(v), X.ptr(o2), X.ptr(l2), X.ptr(idx) if idx is not None else None,
                                    X.ptr(cnt) if cnt is not None else None, B, 12, L, keys.idx.shape[1], 0, 0,
                                    x.stride(1), x.stride(0), kv_rs, x.stride(0), out.stride(1), out.stride(0),
                                    0.125, X.dtype_code(x), 0.0, 0, X.stream())
    assert fwd(x.stride(1)) == 0
    for bad in (1 << 24, 1 << 23):                       # 256 rows x 2^24 (or 2^23) elements x 2 B >= 4 GB of 32-bit byte offsets
        rc = fwd(bad)
        assert rc != 0
        with pytest.raises(RuntimeError, match="span < 4 GB"):
            X.check(rc, "t2s_attn_fwd")
    rc = fwd(x.stride(1), idx=keys.idx, cnt=None)
    assert rc != 0
    with pytest.raises(RuntimeError, match="kv_idx and kv_cnt"):
        X.check(rc, "t2s_attn_fwd")
    torch.cuda.synchronize()


def test_c_abi_from_two_threads():
    """nn.DataParallel (the reference's shipped default, base_trainer.py:121-126) calls forward from one Python thread per
    replica: the C ABI keeps no global mutable state and its error string is thread-local.  Two threads hammer different
    kernels on their own streams; results must equal the single-threaded ones, and an argument error raised in one thread
    must not leak into the other's t2s_last_error()."""
    _need_gpu()
    import threading
    from vitxt_gqa_amd import hipext as X, ops
    g = torch.Generator().manual_seed(23)
    u = (torch.randn(4096, 3072, generator=g)).to(DEV).to(torch.bfloat16)
    x = torch.randn(4096, 768, generator=g).to(DEV)
    gam, bet = torch.ones(768, device=DEV), torch.zeros(768, device=DEV)
    qkv = torch.randn(2, 300, 2304, generator=g).to(DEV).to(torch.bfloat16)
    keys = ops.compact_keys(torch.ones(2, 300, dtype=torch.bool, device=DEV))
    want_g = ops.gelu_fwd(u)
    want_l = ops.add_layernorm_fwd(x.clone(), None, gam, bet, save=False)[0]
    want_a = ops.attn_fwd(qkv, keys)[0]
    torch.cuda.synchronize()
    errs, msgs = [], {}

    def worker(which):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for i in range(30):
                    if which == 0:
                        assert torch.equal(ops.gelu_fwd(u), want_g)
                        assert torch.equal(ops.attn_fwd(qkv, keys)[0], want_a)
                    else:
                        assert torch.equal(ops.add_layernorm_fwd(x.clone(), None, gam, bet, save=False)[0], want_l)
                        if i % 5 == 0:          # an argument error in THIS thread: rc != 0 and a message of its own
                            rc = X.lib().t2s_gelu_fwd(None, None, 16, X.T2S_BF16, None)
                            assert rc != 0
                            msgs[which] = X.lib().t2s_last_error().decode()
                st.synchronize()
            if which == 0:
                msgs[which] = X.lib().t2s_last_error().decode()
        except BaseException as e:          # noqa: BLE001
            errs.append((which, repr(e)))

    th = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    assert "gelu" in msgs[1] and "gelu" not in msgs[0], msgs


@pytest.mark.parametrize("dq_mode", [1, 0])
@pytest.mark.parametrize("drop_p", [0.0, 0.1])
@pytest.mark.parametrize("B,L1,n_dec,keep", CASES + [(2, 1000, 12, 0.7), (2, 900, 0, 1.0), (1, 1500, 12, 0.3), (2, 768, 0, 1.0), (1, 1140, 12, 1.0),
                                                    (1, 1150, 12, 1.0)])
def test_attention_bwd_fused_five_products(B, L1, n_dec, keep, drop_p, dq_mode):
    """t2s_attn_bwd_fused (one key-stationary kernel, S and dP computed once, dQ summed across the 384-key blocks of a pair by the
    ordered hand-off - dq_mode 1, the default - or with fp32 atomics - dq_mode 0) against the fp64 gradient and against the
    two-kernel form; dK / dV of rows outside the key list exactly zero, with and without the in-call zero fill.  The last three
    cases end the key list exactly ON a block boundary (768 = 2 x 384 prefix keys without decoder keys; 1140 + 12 = 3 x 384) and
    split the 12 decoder keys over two edge blocks (1150 + 12: two keys in the third block, ten in the fourth): who finishes dQ."""
    _need_gpu()
    from vitxt_gqa_amd import ops
    if dq_mode == 0 and drop_p and L1 < 700:
        pytest.skip("the atomic form's small dropout cases add nothing over the hand-off's")
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
    scale = two.float().abs().max().item()
    gref = None
    if not drop_p:                                     # without dropout: the fp64 gradient (with it: the two-kernel form, which
        xr = x.double().requires_grad_(True)           # tests/test_dropout_gpu.py checks against a dense restatement with the same mask)
        ref = ref_attention(xr, dense_mask(valid, n_dec, L), 0.125)
        (gref,) = torch.autograd.grad(ref, xr, dout.double())
        scale = gref.abs().max().item()
    for kl in (keys, ops.KeyList(keys.idx, keys.cnt, keys.n_dec, keys.dec_q0, None, None)):       # in-call fill / caller's zero fill
        got = ops.attn_bwd(x, out, dout, lse, kl, fused=True, dq_mode=dq_mode, **kw)
        assert ops.fused_handoff_status() == 0
        if dq_mode == 1:                               # the ordered hand-off sums in a fixed order: bit-reproducible
            assert torch.equal(got, ops.attn_bwd(x, out, dout, lse, kl, fused=True, dq_mode=1, **kw))
        if gref is not None:
            gerr = (got.double() - gref).abs().max().item()
            assert gerr < 3e-2 * max(1.0, scale) * 2, "fused bwd max err %.3e (scale %.3e)" % (gerr, scale)
        assert (got.double() - two.double()).abs().max().item() < 3e-2 * max(1.0, scale)
        kvalid = torch.cat([valid, torch.ones(B, n_dec, dtype=torch.bool, device=DEV)], 1)
        if (~kvalid).any():
            assert got[..., 768:][~kvalid].abs().max().item() == 0
        # dK / dV are deterministic and computed by the same arithmetic as the two-kernel form up to the K pre-scaling path
        assert (got[..., 768:].double() - two[..., 768:].double()).abs().max().item() < 2e-2 * max(1.0, scale)


@pytest.mark.parametrize("dq_mode", [1, 0])
def test_attention_bwd_of_a_sample_without_keys(dq_mode):
    """Empty input: one sample of the batch lists NO key (no visible prefix row, no decoder rows).  Its attention output is zero, so
    every gradient of that sample is exactly zero - in the hand-off form too, where dQ is written by the last key block of a pair and
    such a sample has none (the prep kernel writes its rows) - and the other sample comes out as when it runs alone."""
    _need_gpu()
    from vitxt_gqa_amd import ops
    B, L = 2, 1100
    g = torch.Generator().manual_seed(23)
    x = (torch.randn(B, L, 2304, generator=g) * 0.7).to(DEV).to(torch.bfloat16)
    dout = torch.randn(B, L, 768, generator=g).to(DEV).to(torch.bfloat16)
    valid = torch.zeros(B, L, dtype=torch.bool)
    valid[0, torch.randperm(L, generator=g)[:600]] = True
    valid = valid.to(DEV)
    keys = ops.compact_keys(valid)
    assert keys.cnt.tolist() == [600, 0]
    out, lse = ops.attn_fwd(x, keys)
    assert out[1].abs().max().item() == 0
    poison = torch.full((64 << 20,), float("nan"), device=DEV)        # recycle NaN-filled blocks through the caching allocator
    del poison
    got = ops.attn_bwd(x, out, dout, lse, keys, fused=True, dq_mode=dq_mode)
    assert dq_mode == 0 or ops.fused_handoff_status() == 0
    assert torch.isfinite(got.float()).all() and got[1].abs().max().item() == 0
    two = ops.attn_bwd(x, out, dout, lse, keys, fused=False)
    assert two[1].abs().max().item() == 0
    k1 = ops.compact_keys(valid[:1])
    o1, l1 = ops.attn_fwd(x[:1].contiguous(), k1)
    alone = ops.attn_bwd(x[:1].contiguous(), o1, dout[:1].contiguous(), l1, k1, fused=True, dq_mode=dq_mode)
    if dq_mode == 1:
        assert torch.equal(got[:1], alone)
    else:
        assert (got[:1].float() - alone.float()).abs().max().item() <= 2.0 ** -6 * alone.float().abs().max().item()


@pytest.mark.parametrize("B", [16, 3])
@pytest.mark.parametrize("drop_p", [0.0, 0.1])
def test_attention_bwd_fused_handoff_under_uneven_load(drop_p, B):
    """The dQ hand-off protocol (csrc/attn_bwd_fused_bf16.hip, FbWork; Guideline 16 R1) where it is stressed: 16 x 12 = 192
    (sample, head) chains of 1 .. 14 key blocks each (visible keys from 3 % to 100 % of 5 300 rows) - about 1 500 workgroups for 256
    CUs, so later tickets start while earlier chains are mid-sweep, consumers re-read lines their CU has seen before (the running
    sums are rewritten in place by every block) and the chains differ in length by an order of magnitude.  Asserted: no spin
    timed out; two launches give bit-identical gradients; dQ equals the atomic form's to fp32 summation-order noise (one bf16
    rounding step at most, and only rarely); dK / dV are identical in both forms.  The tickets are compact over the key blocks that
    exist (slot table built by the prep kernel): B = 3 gives 36 pairs = 4 full XCD groups + one with 4 of 8 members (empty table tails)
    with one-block and fourteen-block chains side by side."""
    _need_gpu()
    from vitxt_gqa_amd import ops
    L1, n_dec = 5300, 12
    L = L1 + n_dec
    g = torch.Generator().manual_seed(41)
    x = (torch.randn(B, L, 2304, generator=g) * 0.8).to(DEV).to(torch.bfloat16)
    dout = torch.randn(B, L, 768, generator=g).to(DEV).to(torch.bfloat16)
    keep = torch.linspace(0.03, 1.0, B).view(B, 1)
    valid = (torch.rand(B, L1, generator=g) < keep).to(DEV)
    valid[:, 0] = True
    keys = ops.compact_keys(valid, n_dec=n_dec, dec_row0=L1)
    kw = dict(drop_p=drop_p, drop_seed=515) if drop_p else {}
    out, lse = ops.attn_fwd(x, keys, **kw)
    a = ops.attn_bwd(x, out, dout, lse, keys, fused=True, dq_mode=1, **kw)
    assert ops.fused_handoff_status() == 0
    for _ in range(3):                                  # warm caches, other tickets orders
        b = ops.attn_bwd(x, out, dout, lse, keys, fused=True, dq_mode=1, **kw)
        assert ops.fused_handoff_status() == 0
        assert torch.equal(a, b)
    c = ops.attn_bwd(x, out, dout, lse, keys, fused=True, dq_mode=0, **kw)
    assert torch.equal(a[..., 768:], c[..., 768:])
    dq_a, dq_c = a[..., :768].float(), c[..., :768].float()
    diff = (dq_a - dq_c).abs()
    assert (diff <= 2.0 ** -7 * dq_c.abs().clamp_min(1e-3)).all()                  # at most one bf16 step
    assert (diff > 0).float().mean().item() < 0.02
    two = ops.attn_bwd(x, out, dout, lse, keys, fused=False, **kw)
    assert (a.float() - two.float()).abs().max().item() < 3e-2 * max(1.0, two.float().abs().max().item())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_glue_passes_tanh_residual_and_add_cast(dtype):
    """csrc/glue.hip: QTV's residual x + tanh(enc) (t2s.py:428-432) forward / backward and the fp32 + operand-dtype add, against the
    framework ops they replace; the gradient input as a ROW SLICE of a longer buffer (how it arrives from the MMT input gradient)."""
    _need_gpu()
    from vitxt_gqa_amd import ops
    g = torch.Generator().manual_seed(5)
    B, L, extra = 3, 37, 9
    x = torch.randn(B, L, 768, generator=g).to(DEV)
    enc = (torch.randn(B, L, 768, generator=g) * 1.5).to(DEV)
    y = ops.tanh_residual_fwd(x, enc)
    assert (y - (x + torch.tanh(enc))).abs().max().item() < 2e-6
    big = torch.randn(B, L + extra, 768, generator=g).to(DEV)
    gy = big[:, :L]                                            # batch stride (L + extra) * 768
    assert not gy.is_contiguous()
    ge = ops.tanh_residual_bwd(gy, enc, dtype)
    ref = gy * (1 - torch.tanh(enc) ** 2)
    assert ge.dtype == dtype and (ge.float() - ref).abs().max().item() < (2e-6 if dtype == torch.float32 else 2e-2)
    d = torch.randn(B * L, 768, generator=g).to(DEV).to(dtype)
    out = ops.add_cast(gy, d)
    assert out.is_contiguous() and torch.equal(out, gy + d.float().view(B, L, 768))
    with pytest.raises(AssertionError):
        ops.tanh_residual_bwd(big[:, :L, :767], enc[..., :767], dtype)


def test_attention_score_on_a_row_slice_of_a_longer_sequence():
    """The grounding scorers read the frame / OCR rows in place inside QTV's [question; frames; OCR] output (batch stride)."""
    _need_gpu()
    from vitxt_gqa_amd import ops
    g = torch.Generator().manual_seed(6)
    B, T, M = 3, 20, 50
    buf = torch.randn(B, T + M + 7, 768, generator=g).to(DEV)
    q = torch.randn(B, 768, generator=g).to(DEV) * 0.05
    mask = (torch.rand(B, M, generator=g) < 0.7).float().to(DEV)
    view = buf[:, T:T + M]
    assert torch.equal(ops.attention_score(q, view, mask), ops.attention_score(q, view.contiguous(), mask))


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("drop_p", [0.0, 0.25])
def test_ocr_encoding_tail_matches_the_framework_ops(dtype, tol, drop_p):
    """t2s_ocr_tail_fwd / _bwd against the ops of T2S._forward_ocr_encoding they replace (t2s.py:249-257): LN(a) + LN(Linear(bbox)),
    dropout with the exported mask; every gradient (a, the two LayerNorm affines, the box weight and bias) against autograd."""
    _need_gpu()
    from vitxt_gqa_amd import ops
    g = torch.Generator().manual_seed(11)
    rows, seed = 1037, 987654
    a = (torch.randn(rows, 768, generator=g) * 0.7 + 0.1).to(DEV).to(dtype)
    bbox = torch.rand(rows, 4, generator=g).to(DEV)
    wb = (torch.randn(768, 4, generator=g) * 0.5).to(DEV)
    bb = (torch.randn(768, generator=g) * 0.1).to(DEV)
    ga, gb = [(1 + 0.1 * torch.randn(768, generator=g)).to(DEV) for _ in range(2)]
    ba, be = [(0.1 * torch.randn(768, generator=g)).to(DEV) for _ in range(2)]
    gout = torch.randn(rows, 768, generator=g).to(DEV)
    out, stats = ops.ocr_tail_fwd(a, bbox, wb, bb, ga, ba, gb, be, drop_p=drop_p, drop_seed=seed)
    keep = ops.dropout_mask(rows * 768, drop_p, seed, DEV).view(rows, 768).double() / (1 - drop_p) if drop_p else 1.0
    P = [t.double().requires_grad_(True) for t in (a, wb, bb, ga, ba, gb, be)]
    ar, wbr, bbr, gar, bar_, gbr, ber = P
    ln = lambda x, w, b: (x - x.mean(-1, keepdim=True)) / torch.sqrt(x.var(-1, unbiased=False, keepdim=True) + 1e-12) * w + b
    ref = (ln(ar, gar, bar_) + ln(bbox.double() @ wbr.t() + bbr, gbr, ber)) * keep
    assert (out.double() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())
    grads = torch.autograd.grad(ref, P, gout.double())
    d_a, dga, dba, dgb, dbe, db_box, dw_box = ops.ocr_tail_bwd(gout, a, bbox, wb, bb, ga, gb, stats, drop_p=drop_p, drop_seed=seed)
    assert d_a.dtype == dtype and (d_a.double() - grads[0]).abs().max().item() < tol * max(1.0, grads[0].abs().max().item())
    for name, got, want in (("dw_box", dw_box, grads[1]), ("db_box", db_box, grads[2]), ("dga", dga, grads[3]), ("dba", dba, grads[4]),
                            ("dgb", dgb, grads[5]), ("dbe", dbe, grads[6])):
        assert got.shape == want.shape, name
        assert (got.double() - want).norm().item() < 1e-4 * want.norm().item() + 1e-6, name


@pytest.mark.gpu
def test_recorded_gemm_selections_load_and_change_no_result_beyond_rounding():
    """vitxt_gqa_amd/gemm_tuning.py: the recorded hipBLASLt selections load on this card (torch's validators accept the file) and a
    GEMM of a recorded shape gives the library default's result up to the rounding of a different summation order."""
    _need_gpu()
    import torch.cuda.tunable as tunable
    from vitxt_gqa_amd import gemm_tuning
    if not gemm_tuning.enable_tuned_gemms():
        pytest.skip("recorded selections not active (T2S_TUNED_GEMMS=0, TunableOp driven from outside, or other library versions)")
    assert tunable.is_enabled() and not tunable.tuning_is_enabled() and len(tunable.get_results()) > 20
    g = torch.Generator(device="cpu").manual_seed(3)
    M = 647680 // 8                     # (a recorded shape has M = 647 680; any M exercises the lookup, the small one keeps the test light)
    a = torch.randn(M, 768, generator=g).to(DEV).to(torch.bfloat16)
    w = (torch.randn(2304, 768, generator=g) * 0.05).to(DEV).to(torch.bfloat16)
    b = torch.randn(2304, generator=g).to(DEV).to(torch.bfloat16)
    y1 = torch.addmm(b, a, w.t())
    tunable.enable(False)
    try:
        y0 = torch.addmm(b, a, w.t())
    finally:
        tunable.enable(True)
    assert (y1.float() - y0.float()).abs().max().item() <= 2.0 ** -6 * max(1.0, y0.float().abs().max().item())
