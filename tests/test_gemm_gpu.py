"""Own bf16 MFMA GEMMs (vitxt_gqa_amd/csrc/gemm_bf16.hip) through the C ABI against fp32 torch on the shapes of the train step
(M = token rows, N / K in {768, 2304, 3072}) and on ragged shapes; bf16 operands, fp32 accumulation: the error against an fp64 product
of the SAME bf16 operands is the output rounding (2^-9 relative) plus fp32 accumulation noise."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(a, w):
    return a.double() @ w.double().t()


def _mk(M, N, K, seed, dev="cuda"):
    g = torch.Generator(device=dev).manual_seed(seed)
    a = (torch.rand(M, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    w = (torch.rand(N, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    return a, w


def _check_bf16(out, ref, what):
    err = (out.double() - ref).abs()
    tol = ref.abs() * 2.0 ** -8 + 1e-2
    bad = (err > tol).sum().item()
    print("%s: max abs err %.4g (max |ref| %.4g), %d outside tolerance" % (what, err.max().item(), ref.abs().max().item(), bad))
    assert bad == 0, what


@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (512, 768, 768), (1000, 776, 256), (3 * 256 + 17, 3072, 768), (2048, 768, 3072), (777, 2304, 768), (1, 8, 128)])
def test_gemm_nt_store_and_bias(M, N, K):
    from vitxt_gqa_amd import gemm as G
    a, w = _mk(M, N, K, 1)
    bias = (torch.randn(N, device="cuda")).to(torch.bfloat16)
    out = G.gemm_nt(a, w, bias)
    torch.cuda.synchronize()
    _check_bf16(out, _ref(a, w) + bias.double(), "nt+bias %dx%dx%d" % (M, N, K))
    out2 = G.gemm_nt(a, w)
    _check_bf16(out2, _ref(a, w), "nt %dx%dx%d" % (M, N, K))
    # asymmetric exact-integer check (catches a transposed tile or a swapped k order exactly)
    ai = torch.randint(-3, 4, (M, K), device="cuda").to(torch.bfloat16)
    wi = torch.randint(-3, 4, (N, K), device="cuda").to(torch.bfloat16)
    oi = G.gemm_nt(ai, wi)
    assert torch.equal(oi.double(), _ref(ai, wi).to(torch.bfloat16).double())


def test_gemm_nt_strided_operands_and_accumulate():
    from vitxt_gqa_amd import gemm as G
    M, N, K = 1300, 768, 768
    torch.manual_seed(11)
    big = (torch.rand(M, 2304, device="cuda") * 2 - 1).to(torch.bfloat16)
    a = big[:, 768:1536]                                   # row stride 2304
    w = ((torch.rand(N, K, device="cuda") * 2 - 1) * 0.1).to(torch.bfloat16)
    c0 = torch.randn(M, N, device="cuda").to(torch.bfloat16)
    c = c0.clone()
    G.gemm_nt(a, w, out=c, accumulate=True)
    prod = _ref(a, w).to(torch.bfloat16).double()          # the product is rounded to bf16, then added in fp32 and rounded again
    want = (c0.double() + prod).to(torch.bfloat16)
    # (the kernel's fp32 sum may round the product one bf16 step away from the fp64 product's rounding: one ulp of the product, and
    # the second rounding may then flip too: one ulp of the sum)
    diff = (c.double() - want.double()).abs()
    ulp = (want.double().abs() + prod.abs()) * 2.0 ** -7 + 1e-2
    assert (diff <= ulp).all(), diff.max().item()
    assert (diff == 0).double().mean().item() > 0.97


@pytest.mark.parametrize("M", [512, 1300])
def test_gemm_nt_gelu_grad_epilogue(M):
    from vitxt_gqa_amd import gemm as G, ops
    N, K = 3072, 768
    dy, w = _mk(M, N, K, 3)
    w = (w.float() * 0.05).to(torch.bfloat16)
    u = (torch.randn(M, N, device="cuda") * 1.5).to(torch.bfloat16)
    du, db = G.gemm_nt_gelu_grad(dy, w, u)
    uf = u.double()
    gp = 0.5 * (1 + torch.erf(uf / 2 ** 0.5)) + uf * torch.exp(-0.5 * uf * uf) / (2 * torch.pi) ** 0.5
    ref = _ref(dy, w) * gp
    _check_bf16(du, ref, "gelu' epilogue M=%d" % M)
    assert torch.allclose(db.double(), du.double().sum(0), rtol=1e-4, atol=1e-2 * M ** 0.5)
    # against the two-pass form of the product: library GEMM -> bf16 -> gelu_bwd (one more rounding)
    du2, db2 = ops.gelu_bwd((dy @ w.t()).contiguous(), u)
    assert (du.float() - du2.float()).abs().max().item() <= 2.0 ** -6 * ref.abs().max().item() + 2e-2
    assert torch.allclose(db, db2, rtol=2e-2, atol=0.05 * M ** 0.5)


def test_gemm_nt_gelu_dual_epilogue_equals_the_standalone_gelu():
    from vitxt_gqa_amd import gemm as G, ops
    M, N, K = 900, 3072, 768
    a, w = _mk(M, N, K, 4)
    w = (w.float() * 0.1).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda").to(torch.bfloat16)
    u, g = G.gemm_nt_gelu_dual(a, w, bias)
    _check_bf16(u, _ref(a, w) + bias.double(), "dual: u")
    assert torch.equal(g, ops.gelu_fwd(u)), "gelu(u) through the table must be bit-equal to gelu_fwd_kernel on the same u"


@pytest.mark.parametrize("rows,n_out,n_in,splits", [(1024, 256, 256, 1), (4096, 768, 768, None), (5000, 256, 512, 3), (20000, 3072, 768, None),
                                                    (20000, 768, 3072, None), (8192 + 77, 2304, 768, None)])
def test_gemm_wgrad(rows, n_out, n_in, splits):
    from vitxt_gqa_amd import gemm as G
    g = torch.Generator(device="cuda").manual_seed(5)
    dy = (torch.rand(rows, n_out, device="cuda", generator=g) * 2 - 1).to(torch.bfloat16)
    x = (torch.rand(rows, n_in, device="cuda", generator=g) * 2 - 1).to(torch.bfloat16)
    dw = G.gemm_wgrad(dy, x, splits=splits)
    ref = dy.double().t() @ x.double()
    err = (dw.double() - ref).abs().max().item()
    print("wgrad %d x %d x %d: max abs err %.4g (max |ref| %.4g)" % (rows, n_out, n_in, err, ref.abs().max().item()))
    assert err <= 1e-5 * rows ** 0.5 * 4 + 1e-3                     # fp32 accumulation of exact bf16 products
    dw2 = G.gemm_wgrad(dy, x, splits=splits)
    assert torch.equal(dw, dw2), "the split-K sum runs in a fixed order: bit-reproducible"
    # exact-integer, asymmetric
    dyi = torch.randint(-2, 3, (rows, n_out), device="cuda").to(torch.bfloat16)
    xi = torch.randint(-2, 3, (rows, n_in), device="cuda").to(torch.bfloat16)
    assert torch.equal(G.gemm_wgrad(dyi, xi, splits=splits).double(), dyi.double().t() @ xi.double())
    # accumulate form
    base = torch.randn(n_out, n_in, device="cuda")
    acc = base.clone()
    G.gemm_wgrad(dy, x, out=acc, accumulate=True, splits=splits)
    assert torch.allclose(acc, base + dw, rtol=0, atol=1e-4 * (1 + dw.abs().max().item()))


def test_gemm_wgrad_rows_behind_the_operands_are_never_read():
    """The last row group is ragged: its K-tiles past the end must contribute zeros - with NaN right behind both operands."""
    from vitxt_gqa_amd import gemm as G
    rows, n_out, n_in = 3000, 256, 256
    buf_dy = torch.full((rows + 512, n_out), float("nan"), device="cuda", dtype=torch.bfloat16)
    buf_x = torch.full((rows + 512, n_in), float("nan"), device="cuda", dtype=torch.bfloat16)
    buf_dy[:rows] = torch.randint(-2, 3, (rows, n_out), device="cuda").to(torch.bfloat16)
    buf_x[:rows] = torch.randint(-2, 3, (rows, n_in), device="cuda").to(torch.bfloat16)
    dy, x = buf_dy[:rows], buf_x[:rows]
    for splits in (1, 2, 5):
        dw = G.gemm_wgrad(dy, x, splits=splits)
        assert torch.isfinite(dw).all()
        assert torch.equal(dw.double(), dy.double().t() @ x.double())


def test_bert_layer_with_own_gemms_equals_the_library_form(monkeypatch):
    """The fused BertLayer function (vitxt_gqa_amd/functional.py) with the own GEMM family (weight gradients, gelu' epilogue, gelu dual
    epilogue: the default) against the same function with every GEMM on the library (T2S_OWN_GEMM=none): output, input gradient and all
    16 parameter gradients agree to bf16 noise; the own form is bit-reproducible."""
    from vitxt_gqa_amd import functional as FN, ops
    from vitxt_gqa_amd.t2s import BertLayerParams
    torch.manual_seed(0)
    lp = BertLayerParams()
    for p in lp.parameters():
        p.data.normal_(0, 0.05)
    lp.attention.output.LayerNorm.weight.data.add_(1.0)
    lp.output.LayerNorm.weight.data.add_(1.0)
    lp = lp.to("cuda")
    B, L1, n_dec = 2, 700, 12
    L = L1 + n_dec
    x = torch.randn(B, L, 768, device="cuda").to(torch.bfloat16).float()
    valid = (torch.rand(B, L1, device="cuda") < 0.6)
    valid[:, 0] = True
    dy = torch.randn(B, L, 768, device="cuda").to(torch.bfloat16).float()
    keys = ops.compact_keys(valid, n_dec=n_dec, dec_row0=L1)

    def run(own):
        monkeypatch.setattr(FN, "OWN_GEMM", frozenset(own))
        for p in lp.parameters():
            p.grad = None
        xg = x.clone().requires_grad_(True)
        y, _ = FN.bert_layer(xg, None, keys, lp, torch.bfloat16)
        y.backward(dy)
        return y.detach(), xg.grad.clone(), {n: p.grad.clone() for n, p in lp.named_parameters()}

    y0, dx0, g0 = run(())
    y1, dx1, g1 = run(("wgrad", "gelu_bwd", "gelu_fwd"))
    y2, dx2, g2 = run(("wgrad", "gelu_bwd", "gelu_fwd"))
    assert torch.equal(y1, y2) and torch.equal(dx1, dx2) and all(torch.equal(g1[n], g2[n]) for n in g1)
    assert (y1 - y0).abs().max().item() < 3e-2
    assert (dx1 - dx0).abs().max().item() < 3e-2 * max(1.0, dx0.abs().max().item())
    for n in g0:
        rel = (g1[n] - g0[n]).norm().item() / max(g0[n].norm().item(), 1e-6 * g0[n].numel() ** 0.5)
        if "key.bias" in n:
            continue
        assert rel < 2e-2, "%s: own vs library GEMMs rel %.3e" % (n, rel)


@pytest.mark.parametrize("per_xcd", ["1", "2", "5"])
def test_persistent_nt_walk_long_walks_are_exact(per_xcd, monkeypatch):
    """Round 6: the NT kernel is persistent (one workgroup per CU walks its XCD's tiles; K-tile 0 of the next tile is fetched under the
    current tile's epilogue; the epilogue stages through LDS buffer 1 only).  On a 256-CU card the unit tests' shapes give most workgroups a
    single tile - here the probe knob T2S_GEMM_NT_PER_XCD (read per call) shrinks the grid to 8 / 16 / 40 workgroups, so EVERY workgroup
    walks many tiles through the prefetched continuation, for all four epilogues; exact-integer operands make a stale LDS buffer, a tile
    computed twice or skipped, or a wrong row of the staging rounds show as a wrong integer."""
    from vitxt_gqa_amd import gemm as G
    monkeypatch.setenv("T2S_GEMM_NT_PER_XCD", per_xcd)
    M, N, K = 9 * 256 + 77, 1280, 384                      # 10 x 5 tiles, ragged last M-block; 6 K-tiles
    g = torch.Generator(device="cuda").manual_seed(5)
    ai = torch.randint(-3, 4, (M, K), device="cuda", generator=g).to(torch.bfloat16)
    wi = torch.randint(-3, 4, (N, K), device="cuda", generator=g).to(torch.bfloat16)
    bi = torch.randint(-2, 3, (N,), device="cuda", generator=g).to(torch.bfloat16)
    want = _ref(ai, wi) + bi.double()                      # |values| <= 9 * 384 + 2: exact in fp32, rounded once to bf16
    out = G.gemm_nt(ai, wi, bi)
    assert torch.equal(out.double(), want.to(torch.bfloat16).double())
    c0 = torch.randint(-4, 5, (M, N), device="cuda", generator=g).to(torch.bfloat16)
    c = c0.clone()
    G.gemm_nt(ai, wi, out=c, accumulate=True)
    assert torch.equal(c.double(), (c0.double() + _ref(ai, wi).to(torch.bfloat16).double()).to(torch.bfloat16).double())
    # the two GELU forms against the standalone kernels on the same pre-activation / product
    from vitxt_gqa_amd import ops
    a, w = _mk(M, N, K, 7)
    w = (w.float() * 0.05).to(torch.bfloat16)
    bias = (torch.randn(N, device="cuda") * 0.1).to(torch.bfloat16)
    u, gact = G.gemm_nt_gelu_dual(a, w, bias)
    u_ref = G.gemm_nt(a, w, bias)
    assert torch.equal(u, u_ref) and torch.equal(gact, ops.gelu_fwd(u_ref))
    du, db = G.gemm_nt_gelu_grad(a, w, u)
    prod = _ref(a, w)
    x = u.double()
    gp = 0.5 * (1 + torch.erf(x * 0.7071067811865476)) + x * torch.exp(-0.5 * x * x) * 0.3989422804014327
    _check_bf16(du, prod * gp, "persistent walk, gelu' epilogue, per_xcd %s" % per_xcd)
    assert (db.double() - (prod * gp).sum(0)).abs().max().item() < 2e-3 * (prod * gp).abs().sum(0).max().item() + 1e-2
