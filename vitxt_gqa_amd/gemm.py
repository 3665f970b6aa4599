"""Own bf16 MFMA GEMMs of the BERT block's linear layers (``csrc/gemm_bf16.hip`` through ``t2s_gemm_nt`` / ``t2s_gemm_wgrad``):
the ``torch.nn.Linear`` calls inside the third-party BertSelfOutput / BertIntermediate / BertOutput the reference runs at
pythia/models/t2s.py:423-427,538-542,622-626, and their gradients, where an epilogue the library GEMM cannot fuse pays
(DESIGN.md section 5): the GELU backward + FFN bias gradient inside the dgrad of BertOutput.dense, the GELU forward inside the
BertIntermediate GEMM, and the weight gradients as one deterministic split-K kernel.  No fallback inside these functions: callers
(functional.py) decide per shape which form a GEMM takes."""
import torch

from . import hipext as X

EPI_STORE, EPI_ACCUM, EPI_GELU_GRAD, EPI_GELU_DUAL = 0, 1, 2, 3
_TABLES = {}


def gelu_tables(device):
    """(gelu bf16 [65536], gelu' fp32 [65536]) of every bf16 bit pattern, built once per device by the exact-erf arithmetic of the
    standalone GELU kernels (so the fused epilogues are bit-equal to gelu_fwd / use the very gelu' of gelu_bwd)."""
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    if key not in _TABLES:
        fwd = torch.empty(65536, dtype=torch.bfloat16, device=device)
        grad = torch.empty(65536, dtype=torch.float32, device=device)
        X.check(X.lib().t2s_gelu_tables(X.ptr(fwd), X.ptr(grad), X.stream()), "t2s_gelu_tables")
        _TABLES[key] = (fwd, grad)
    return _TABLES[key]


def _rows(t):
    assert t.dim() == 2 and t.stride(1) == 1 and t.dtype == torch.bfloat16, "expected a bf16 [rows, cols] matrix with dense rows"
    return t.stride(0)


def nt_supported(M, N, K, lda=None, ldw=None):
    """The shape rules of t2s_gemm_nt (csrc/gemm_bf16.hip): K a multiple of 128, N of 8, M < 2^31, and 256 operand rows spanning
    < 2^31 bytes (lda, ldw * 512 < 2^31)."""
    lda = K if lda is None else lda
    ldw = K if ldw is None else ldw
    return K % 128 == 0 and N % 8 == 0 and 0 < M < 1 << 31 and lda * 512 < 1 << 31 and ldw * 512 < 1 << 31


def gemm_nt(a, w, bias=None, out=None, accumulate=False):
    """a [M, K] @ w[N, K]^T (+ bias) -> bf16 [M, N]; ``accumulate``: out += product (out given, no bias)."""
    M, K = a.shape
    N = w.shape[0]
    assert w.shape[1] == K and nt_supported(M, N, K)
    if out is None:
        assert not accumulate
        out = torch.empty(M, N, dtype=torch.bfloat16, device=a.device)
    assert out.shape == (M, N)
    X.check(X.lib().t2s_gemm_nt(X.ptr(a), X.ptr(w), X.ptr(bias), X.ptr(out), M, N, K, _rows(a), _rows(w), _rows(out),
                                EPI_ACCUM if accumulate else EPI_STORE, None, None, None, None, X.stream()), "t2s_gemm_nt")
    return out


def gemm_nt_gelu_grad(dy, w, u):
    """du = (dy [M, K] @ w[N, K]^T) * gelu'(u [M, N]) with the product kept in fp32 (one rounding), and db = column sums of du
    (fp32 [N]): BertIntermediate's GELU backward and bias gradient inside the input-gradient GEMM of BertOutput.dense."""
    M, K = dy.shape
    N = w.shape[0]
    assert u.shape == (M, N) and u.is_contiguous() and nt_supported(M, N, K)
    du = torch.empty_like(u)
    parts = torch.empty(X.lib().t2s_gemm_nt_colsum_rows(M), N, dtype=torch.float32, device=u.device)
    X.check(X.lib().t2s_gemm_nt(X.ptr(dy), X.ptr(w), None, X.ptr(du), M, N, K, _rows(dy), _rows(w), N, EPI_GELU_GRAD,
                                X.ptr(u), None, X.ptr(gelu_tables(u.device)[1]), X.ptr(parts), X.stream()), "t2s_gemm_nt")
    return du, parts.sum(0)


def gemm_nt_gelu_dual(a, w, bias):
    """u = a @ w^T + bias (bf16) and g = gelu(u) from one GEMM: (u, g)."""
    M, K = a.shape
    N = w.shape[0]
    assert nt_supported(M, N, K)
    u = torch.empty(M, N, dtype=torch.bfloat16, device=a.device)
    g = torch.empty_like(u)
    X.check(X.lib().t2s_gemm_nt(X.ptr(a), X.ptr(w), X.ptr(bias), X.ptr(u), M, N, K, _rows(a), _rows(w), N, EPI_GELU_DUAL,
                                None, X.ptr(g), X.ptr(gelu_tables(a.device)[0]), None, X.stream()), "t2s_gemm_nt")
    return u, g


def wgrad_supported(rows, n_out, n_in, ld_dy=None, ld_x=None, splits=None):
    """Whether t2s_gemm_wgrad takes the shape: the tile rule AND the limits its host side checks (csrc/gemm_bf16.hip t2s_gemm_wgrad:
    rows < 2^31, at most 4096 row splits, and a row split - padded to 128 rows, + one 64-row K-tile of read-ahead - spanning < 2 GB of
    each operand, whose offsets are 32-bit).  ``splits`` None = the automatic count (one round of workgroups on the card); a shape that
    fails here goes to the library's batched GEMM (functional._wgrad) instead of raising inside the call."""
    if n_out % 256 or n_in % 256 or rows < 1024 or rows >= 1 << 31:
        return False
    if splits is None:
        splits = int(X.lib().t2s_gemm_wgrad_splits(rows, n_out, n_in))
    if not 1 <= splits <= 4096:
        return False
    chunk = -(-rows // splits)
    chunk = -(-chunk // 128) * 128
    ld_dy = n_out if ld_dy is None else ld_dy
    ld_x = n_in if ld_x is None else ld_x
    return (chunk + 64) * ld_dy * 2 < 1 << 31 and (chunk + 64) * ld_x * 2 < 1 << 31


def gemm_wgrad(dy, x, out=None, accumulate=False, splits=None):
    """dW [n_out, n_in] fp32 = dy[rows, n_out]^T @ x[rows, n_in] (bf16 operands): deterministic split-K over row groups."""
    rows, n_out = dy.shape
    n_in = x.shape[1]
    if splits is None:
        splits = X.lib().t2s_gemm_wgrad_splits(rows, n_out, n_in)
    assert x.shape[0] == rows and wgrad_supported(rows, n_out, n_in, _rows(dy), _rows(x), splits)
    if out is None:
        assert not accumulate
        out = torch.empty(n_out, n_in, dtype=torch.float32, device=dy.device)
    slabs = torch.empty(splits, n_out, n_in, dtype=torch.float32, device=dy.device)
    X.check(X.lib().t2s_gemm_wgrad(X.ptr(dy), X.ptr(x), X.ptr(out), X.ptr(slabs), rows, n_out, n_in, _rows(dy), _rows(x), splits,
                                   1 if accumulate else 0, X.stream()), "t2s_gemm_wgrad")
    return out


# ---- Python mirror of the weight-gradient kernel's workgroup map (csrc/gemm_bf16.hip: tn_item and the grid rule of t2s_gemm_wgrad), for the
# CPU test that enumerates it: every (row split, tile) must be taken exactly once for any tile count / split count / card size.
def wgrad_grid(T, S, cus=256):
    items = T * S
    c = (items + 7) // 8
    if items <= cus and c < cus // 8:
        c = cus // 8
    return c


def wgrad_item(wg, c, T, S):
    x, j = wg % 8, wg // 8
    if T <= c:
        spx = c // T
        hosted = spx * T
        if j < hosted:
            split, tile = x * spx + j // T, j % T
        else:
            q = x * (c - hosted) + (j - hosted)
            split, tile = 8 * spx + q // T, q % T
        return (split, tile) if split < S else None
    if x < S:
        return (x, j) if j < c else None
    L = T - c
    q = (x - S) * c + j
    split, tile = q // L, c + q % L
    return (split, tile) if split < S else None


# ---- the NT kernel's work map (csrc/gemm_bf16.hip nt_item), mirrored for the CPU tests: XCD x = wg % 8 owns a contiguous range of
# M-blocks; inside it the N-tiles go in groups of b - group-major, then M-block, then the N-tile inside the group
def nt_grid(tiles_m, tiles_n):
    """Number of VIRTUAL work items of the NT kernel (8 x the items of the longest XCD range); the launch has one workgroup per CU, and
    workgroup (XCD x, slot s of ``per_xcd``) walks the items j = s, s + per_xcd, ... of XCD x (``nt_walk``)."""
    return 8 * ((tiles_m + 7) // 8) * tiles_n


def nt_walk(xcd, slot, per_xcd, tiles_m, tiles_n, b):
    """The tiles the persistent workgroup (xcd, slot) computes, in order (csrc/gemm_bf16.hip gemm_nt_bf16_kernel's loop)."""
    out, j = [], slot
    while True:
        it = nt_item(j * 8 + xcd, tiles_m, tiles_n, b)
        if it is None:
            return out
        out.append(it)
        j += per_xcd


def nt_group(tiles_n):
    return tiles_n          # the shipped default (fewer N-tiles per group move fewer bytes but take longer: profiles/r05_gemm_nt_map.txt)


def nt_item(wg, tiles_m, tiles_n, b):
    x, j = wg % 8, wg // 8
    mbx = (tiles_m + 7) // 8
    m_lo = x * mbx
    mcount = min(mbx, tiles_m - m_lo)
    if mcount <= 0:
        return None
    full = tiles_n // b
    rem = tiles_n - full * b
    in_full = full * mcount * b
    if j < in_full:
        per = mcount * b
        g, r = j // per, j % per
        return (m_lo + r // b, g * b + r % b)
    jr = j - in_full
    if rem == 0 or jr >= mcount * rem:
        return None
    return (m_lo + jr // rem, full * b + jr % rem)

