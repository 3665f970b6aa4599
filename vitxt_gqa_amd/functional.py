"""Autograd functions of the T2S hot path, built on the HIP kernels (ops.py) plus library GEMMs.

Precision policy.  ``dtype`` is the GEMM / attention OPERAND type (bf16 for throughput, fp32 for the
parity mode).  The residual stream (layer inputs/outputs, pre-LayerNorm sums) is always fp32: the
LayerNorm kernel reads the bf16 GEMM output plus the fp32 residual and emits the fp32 stream value
together with a bf16 copy for the next GEMM.  bf16 rounding therefore only ever touches matmul
operands, which is what keeps bf16 logits within 1e-2 of the fp32 reference after 8 layers.

One fused ``torch.autograd.Function`` per BERT layer (third-party BertLayer used by TextBert / QTV /
MMT, call sites pythia/models/t2s.py:423-427,538-542,622-626) with a hand-written backward, so that
only what a flash-style backward needs is kept: the operand copy of the layer input, fused QKV,
attention output, log-sum-exp, the two pre-LayerNorm sums (+ statistics) and the FFN pre-activation.
LayerNorm outputs and GELU outputs are recomputed in backward (HBM-cheap) instead of stored.
"""
import contextlib
import os
import threading

import torch

from . import gemm as own
from . import ops

HID = ops.HID
F32 = torch.float32
# True: keep only the FFN pre-activation and rebuild GELU / LN1 outputs in backward (-5 GB per layer at B=64,
# +2 HBM passes); False (default): keep them -- 288 GB of HBM3E leave the room (peak ~210 GB at B=64).
RECOMPUTE_ACTIVATIONS = False
# A pass whose key list has a small STRUCTURAL bound (the pos / neg MMT passes: <= 537 / 62 of the 10 132 rows are ever keys,
# guaranteed by the top-k masks) projects K and V only for the rows that are keys: masked keys contribute exactly nothing
# (deviation 1), so their K / V rows - two thirds of the QKV GEMM, of its input-gradient and of its
# weight-gradient GEMM - are never read.  Lists with a bound above this many keys keep the fused [B, L, 2304] projection.
PRUNE_KV_MAX_KEYS = int(os.environ.get("T2S_PRUNE_KV_MAX_KEYS", "2047"))      # (below ops.ATTN_BWD_FUSED_MIN_KEYS: pruned launches take the two-kernel backward)


# Which GEMMs of the BERT block run on this build's own MFMA kernels (vitxt_gqa_amd/gemm.py, csrc/gemm_bf16.hip) instead of the library,
# bf16 operand mode only - the forms measured faster in the step (profiles/r05_gemm_probe.txt):
#   wgrad     the four weight gradients of a layer: ONE deterministic split-K kernel over all token rows (library: batched GEMM over row
#             groups + an fp32 sum)
#   gelu_bwd  dgrad of BertOutput.dense with gelu'(u) and the FFN bias gradient in its epilogue (library GEMM + standalone gelu_bwd pass)
#   gelu_fwd  BertIntermediate: u and gelu(u) both written by the GEMM (library GEMM + standalone gelu_fwd pass)
# T2S_OWN_GEMM=none (or a comma list) for same-box A/B runs.
OWN_GEMM = frozenset(t for t in os.environ.get("T2S_OWN_GEMM", "wgrad,gelu_bwd,gelu_fwd").split(",") if t and t != "none")


# ---- one rounding less on each side of the attention (late round 6; T2S_FOLD_QSCALE=0 restores the raw projection).  The attention kernels work on
# Q' = scale * log2(e) * Q (forward: Q is pre-scaled into LDS) resp. K' = scale * log2(e) * K (fused backward: the K image) and round the product
# to bf16 AGAIN - a second rounding of an operand that the QKV GEMM had already rounded, and a different one in the two directions: the backward's
# recomputed P = exp2(Q K' - LSE) is not the forward's exp2(Q' K - m); at |S| ~ 100 nats (peaky attention) the two differ by up to
# 2^-9 |S| log2(e) = 0.3 in the exponent.  In the bf16 operand mode the factor c0 = scale * log2(e) is therefore folded into the OPERAND copy of the
# query projection (W_q' = c0 W_q, b_q' = c0 b_q, rounded once from the fp32 masters), the kernels are called with scale' = 1 / log2(e) (their own
# factor scale' * log2(e) = 1: multiplying a bf16 value by it and rounding is exact), and the master gradients of W_q / b_q get the factor c0
# back (chain rule through W_q' = c0 W_q).  Same function, same LSE (natural units of the scaled scores), one rounding per operand, identical
# scores in forward and backward, no cost in time.  The fp32 parity mode is untouched; sequences under FOLD_MIN_L rows (TextBert: the two-kernel
# backward) and the eval decode keep the raw projection.
# Measured on the three reference-generated full-length fixtures (profiles/r06_fold_qscale_gradient_vector.txt, r06_fold_qscale_grad_dev.txt):
# ELEMENT by element the whole bf16 gradient vector moves closer to the fp32 one on all three - relative L2 0.166 -> 0.146 (gain 6; parameters
# beyond 0.20: 74 -> 13 of 154), 0.0329 -> 0.0261 (seed 29, gain 4), 0.0051 -> 0.0048 (reference init) - and the key-weight rows of the peakiest
# layer, 1.6 - 1.8 x the reference's own autocast deviation before, come down to it (0.286 -> 0.188 | 0.177; 0.115 -> 0.071 | 0.065).  What the
# first measurement (round 6, norms only) had held against it - 29 instead of 7 parameter NORMS beyond 3 % at gain 6, the reference's autocast
# run has 31 - is the other side of the same thing: uncorrelated noise inflates a norm, and with it gone the norms sit 1.3 % low.  Logits alike.
LOG2E = 1.4426950408889634
ATTN_SCALE = 0.125                                  # 1 / sqrt(64): BertSelfAttention (third-party block the reference calls, t2s.py:423-427,538-542,622-626)
FOLD_QSCALE = os.environ.get("T2S_FOLD_QSCALE", "1") != "0"      # ON (late round 6): closer to the fp32 gradients element by element on all three full-length fixtures
FOLD_MIN_L = int(os.environ.get("T2S_FOLD_MIN_L", "1024"))          # sequences shorter than this keep the raw query projection (the two-kernel backward's regime)
Q_FOLD = ATTN_SCALE * LOG2E


def _fold(lo, L):
    return bool(lo and FOLD_QSCALE and L >= FOLD_MIN_L)


def _attn_scale(lo, L):
    """The ``scale`` argument of the attention kernels for operands of dtype ``lo`` (True: bf16 mode) on a sequence of L rows."""
    return (1.0 / LOG2E) if _fold(lo, L) else ATTN_SCALE


def _own(kind, *mats):
    return kind in OWN_GEMM and all(m.dtype == torch.bfloat16 and m.dim() == 2 and m.stride(1) == 1 and m.stride(0) % 8 == 0 for m in mats)


def _mm_bias(x2, w, b):
    """x2 [rows, in] @ w[out, in]^T + b  (library GEMM with bias epilogue)."""
    return torch.addmm(b, x2, w.t())


def _wgrad(dy2, x2, B):
    """dW[out, in] = dy2[rows, out]^T @ x2[rows, in] with rows = B * L.  The contraction runs over ALL rows (648 k at
    B=64, L=10132) into a tiny output, which the library runs at 0.4-0.8 PFLOP/s as one GEMM; as a batched GEMM per
    sample (contraction L) followed by an fp32 sum over the B partial products it runs at 0.8-1.1 PFLOP/s
    (tools/wgrad_probe.py), and the partials are summed in fp32 instead of inside a bf16-output GEMM.  Returns fp32."""
    rows = dy2.size(0)
    if _own("wgrad", dy2, x2) and own.wgrad_supported(rows, dy2.size(1), x2.size(1), dy2.stride(0), x2.stride(0)):
        return own.gemm_wgrad(dy2, x2)
    if B < 2 or rows // B < 1024 or dy2.dtype == F32:
        return (dy2.t() @ x2).float()
    # number of row groups: one per sample, except for the two FFN weights (3072 x 768 / 768 x 3072), where 16 larger groups run 2-3 %
    # faster on this library (tools/wgrad_layout_probe.py, rounds 3 and 4: 2.93-2.95 vs 3.00-3.03 ms at 650 k rows; for the 2304- and
    # 768-wide outputs one group per sample stays the best)
    G = B
    if dy2.size(1) * x2.size(1) == 3072 * 768 and B % 16 == 0 and rows % 16 == 0 and B > 16:
        G = 16
    part = torch.bmm(dy2.view(G, rows // G, -1).transpose(1, 2), x2.view(G, rows // G, -1))     # [G, out, in]
    return part.sum(0, dtype=F32)           # fp32: it goes straight into the fp32 gradient of the master weight


def _layer_forward(x2, xl, keys, B, L, W, drop_p, seeds, attn_drop_p, recompute, materialise=True, qkv=None, kv_given=None):
    """One BERT layer on rows: x2 the fp32 residual stream entering the layer ([B*L, 768] tensor, or an ``ops.NormRes`` left
    by the previous layer), xl its operand-dtype copy.  W = operand_weights(...) = (w_qkv, b_qkv, w_ao, b_ao, g1, be1, w_i, b_i, w_o, b_o, g2,
    be2).  Returns (y2, y2_lo or None, tensors to keep for backward).  In the bf16 operand mode the fp32 stream value
    between two residual+LayerNorm blocks exists only in normalised form (pre-LN sum + statistics, both kept for backward
    anyway): the next block's kernel re-normalises it on the fly, so it is never written to HBM; y2 is then an
    ``ops.NormRes`` unless ``materialise`` (last layer of a stack)."""
    w_qkv, b_qkv, w_ao, b_ao, g1, be1, w_i, b_i, w_o, b_o, g2, be2, w_qkv_t, w_ao_t, w_i_t, w_o_t = W
    lo = w_qkv.dtype != F32
    kvc = None
    if kv_given is not None:
        # layer 0 of a pruned pass of SharedPrefixEncoderFn: ``qkv`` is the contiguous Q third of the shared projection and
        # ``kv_given`` the K | V rows of this pass's keys gathered from it
        kvc = kv_given
        att, lse = ops.attn_fwd(qkv, keys.compact(L)[0], scale=_attn_scale(lo, L), drop_p=attn_drop_p, drop_seed=seeds[2], kv=kvc)
    elif qkv is None and _prunable(keys, lo):
        # Q for every row, K | V for the key rows only (KeyList.compact): [B, capK, 1536] instead of [B, L, 1536]
        keys_c, flat_rows, capK = keys.compact(L)
        qkv = _mm_bias(xl, w_qkv[:HID], b_qkv[:HID]).view(B, L, HID)
        kvc = _mm_bias(xl.index_select(0, flat_rows), w_qkv[HID:], b_qkv[HID:]).view(B, capK, 2 * HID)
        att, lse = ops.attn_fwd(qkv, keys_c, scale=_attn_scale(lo, L), drop_p=attn_drop_p, drop_seed=seeds[2], kv=kvc)
    else:
        if qkv is None:      # (given: the projection of layer 0, shared by the passes of SharedPrefixEncoderFn)
            qkv = _mm_bias(xl, w_qkv, b_qkv).view(B, L, 3 * HID)
        att, lse = ops.attn_fwd(qkv, keys, scale=_attn_scale(lo, L), drop_p=attn_drop_p, drop_seed=seeds[2])
    a = _mm_bias(att.view(B * L, HID), w_ao, b_ao)
    y1, y1_lo, z1, st1 = ops.add_layernorm_fwd(a, x2, g1, be1, stream_dtype=F32, want_lo=lo, want_y=not lo, drop_p=drop_p, drop_seed=seeds[0])
    del a
    y1_op = y1_lo if lo else y1
    res1 = ops.NormRes(z1, st1, g1, be1) if lo else y1
    if _own("gelu_fwd", y1_op, w_i):
        u, gact = own.gemm_nt_gelu_dual(y1_op, w_i, b_i)
    else:
        u = _mm_bias(y1_op, w_i, b_i)
        gact = ops.gelu_fwd(u)
    o = _mm_bias(gact, w_o, b_o)
    keep_y2 = materialise or not lo
    y2, y2_lo, z2, st2 = ops.add_layernorm_fwd(o, res1, g2, be2, stream_dtype=F32, want_lo=lo, want_y=keep_y2, drop_p=drop_p, drop_seed=seeds[1])
    if not keep_y2:
        y2 = ops.NormRes(z2, st2, g2, be2)
    if recompute:      # store less: GELU output and the LN1 operand copy are rebuilt in backward
        gact = y1_op = None
    return y2, (y2_lo if lo else None), (xl, qkv, att, lse, z1, st1, u, z2, st2, w_qkv_t, w_ao_t, g1, be1, w_i_t, w_o_t, g2, gact, y1_op, kvc)


def _prunable(keys, lo):
    return lo and keys.bound_is_structural and keys.cap_hint <= PRUNE_KV_MAX_KEYS and keys.cap_hint * 4 <= keys.idx.shape[1]


def _layer_backward(saved, keys, dy, drop_p, seeds, attn_drop_p, stop_at_qkv=False):
    """dy: [B*L, 768] gradient of the layer output (fp32 or the operand dtype).  Returns (dx [B*L, 768] in the operand
    dtype, the 12 parameter gradients in the order of W, all fp32: they are gradients of the fp32 master parameters).  The bias gradients of the two projections that feed a
    residual+LayerNorm block come out of that block's backward kernel (column sums in the same pass)."""
    xl, qkv, att, lse, z1, st1, u, z2, st2, w_qkv_t, w_ao_t, g1, be1, w_i_t, w_o_t, g2, gact, y1_op, kvc = saved      # weights: [in, out] copies
    B, L, _ = qkv.shape
    dt = w_qkv_t.dtype
    lo = dt != F32
    # ---- output LayerNorm + FFN
    dz2, dz2x, dg2, dbe2, db_o = ops.add_layernorm_bwd(dy, z2, st2, g2, out_dtype=dt, drop_p=drop_p, drop_seed=seeds[1], want_bias=True)
    if gact is None:
        gact = ops.gelu_fwd(u)
    dw_o = _wgrad(dz2x, gact, B)
    del gact
    if _own("gelu_bwd", dz2x, w_o_t, u):
        du, db_i = own.gemm_nt_gelu_grad(dz2x, w_o_t, u)
        del dz2x
    else:
        dgact = dz2x @ w_o_t.t()
        del dz2x
        du, db_i = ops.gelu_bwd(dgact, u)
        del dgact
    if y1_op is None:                                                    # recompute the LN1 output
        y1, y1_lo, _, _ = ops.add_layernorm_fwd(z1, None, g1, be1, save=False, stream_dtype=F32, want_lo=lo, want_y=not lo)
        y1_op = y1_lo if lo else y1
        del y1, y1_lo
    dw_i = _wgrad(du, y1_op, B)
    dy1 = dz2.addmm_(du, w_i_t.t())    # + residual branch of LN2, accumulated IN PLACE (out-of-place addmm first copies C)
    del du, y1_op, dz2
    # ---- attention output LayerNorm + projection
    dz1, dz1x, dg1, dbe1, db_ao = ops.add_layernorm_bwd(dy1, z1, st1, g1, out_dtype=dt, drop_p=drop_p, drop_seed=seeds[0], want_bias=True)
    del dy1
    dw_ao = _wgrad(dz1x, att.view(B * L, HID), B)
    datt = (dz1x @ w_ao_t.t()).view(B, L, HID)
    del dz1x
    # ---- attention
    if kvc is not None:          # pruned K / V (see _layer_forward): gradients of Q for every row, of K | V for the key rows
        keys_c, flat_rows, capK = keys.compact(L)
        dq, dkv = ops.attn_bwd(qkv, att, datt, lse, keys_c, scale=_attn_scale(lo, L), drop_p=attn_drop_p, drop_seed=seeds[2], kv=kvc)
        del datt
        dq, dkv = dq.view(B * L, HID), dkv.view(B * capK, 2 * HID)
        if stop_at_qkv:          # SharedPrefixEncoderFn adds them into the summed gradient of the shared projection
            return (dz1, (dq, dkv, flat_rows)), (None, None, dw_ao, db_ao, dg1, dbe1, dw_i, db_i, dw_o, db_o, dg2, dbe2)
        xc = xl.index_select(0, flat_rows)
        dw_qkv = torch.cat([_wgrad(dq, xl, B), (dkv.t() @ xc).float()], 0)
        db_qkv = torch.cat([dq.sum(0, dtype=F32), dkv.sum(0, dtype=F32)], 0)
        dx = dz1.addmm_(dq, w_qkv_t[:, :HID].t())                        # + residual branch of LN1 (in place)
        dx.index_add_(0, flat_rows, dkv @ w_qkv_t[:, HID:].t())          # the key rows' share (positions behind a list: zeros onto row 0)
        return dx, (dw_qkv, db_qkv, dw_ao, db_ao, dg1, dbe1, dw_i, db_i, dw_o, db_o, dg2, dbe2)
    dqkv = ops.attn_bwd(qkv, att, datt, lse, keys, scale=_attn_scale(lo, L), drop_p=attn_drop_p, drop_seed=seeds[2]).view(B * L, 3 * HID)
    del datt
    if stop_at_qkv:          # SharedPrefixEncoderFn sums dqkv over the passes that share this projection and finishes once
        return (dz1, dqkv), (None, None, dw_ao, db_ao, dg1, dbe1, dw_i, db_i, dw_o, db_o, dg2, dbe2)
    dw_qkv = _wgrad(dqkv, xl, B)
    db_qkv = dqkv.sum(0, dtype=F32)
    dx = dz1.addmm_(dqkv, w_qkv_t.t())                                   # + residual branch of LN1 (in place, as above)
    return dx, (dw_qkv, db_qkv, dw_ao, db_ao, dg1, dbe1, dw_i, db_i, dw_o, db_o, dg2, dbe2)


# ---- operand copies of the master parameters.  The autograd functions below take the fp32 MASTER parameters (16 per layer,
# reference names) and hand back fp32 gradients; the operand-dtype copies the GEMMs read (fused [2304, 768] QKV weight, bf16
# casts) are made inside forward, so no gradient makes a bf16 round trip or passes through a cast / cat node of the autograd
# graph.  Inside a ``shared_operands()`` scope (one model forward) the copies are made once per layer and shared by every
# call on that layer - the three MMT passes of a train step; outside a scope nothing is cached, so a parameter edited
# between two calls (optimizer step, ``.data`` surgery, load_state_dict) can never meet a stale copy.
_SCOPE = threading.local()
MASTERS_PER_LAYER = 16
_DGRAD_NT = os.environ.get("T2S_DGRAD_NT", "1") != "0"
W_PER_LAYER = 16          # operand_weights(): the 12 forward operands + 4 transposed weight copies for the input-gradient GEMMs


@contextlib.contextmanager
def shared_operands():
    prev = getattr(_SCOPE, "cache", None)
    _SCOPE.cache = {} if prev is None else prev           # nested scopes share the outer one
    try:
        yield
    finally:
        _SCOPE.cache = prev


def layer_masters(lp):
    """The 16 master parameters of one BERT layer (module tree of t2s.BertLayerParams), in the order the functions expect."""
    a, so = lp.attention.self, lp.attention.output
    return (a.query.weight, a.key.weight, a.value.weight, a.query.bias, a.key.bias, a.value.bias,
            so.dense.weight, so.dense.bias, so.LayerNorm.weight, so.LayerNorm.bias,
            lp.intermediate.dense.weight, lp.intermediate.dense.bias,
            lp.output.dense.weight, lp.output.dense.bias, lp.output.LayerNorm.weight, lp.output.LayerNorm.bias)


@torch.no_grad()
def operand_weights(m, dtype, fold=False):
    """16 masters -> the 12-tuple W of _layer_forward in ``dtype`` (LayerNorm affine stays fp32)."""
    cache = getattr(_SCOPE, "cache", None)
    key = (id(m[0]), dtype, bool(fold))
    if cache is not None and key in cache:
        return cache[key][1]
    wq, wk, wv, bq, bk, bv, w_ao, b_ao, g1, be1, w_i, b_i, w_o, b_o, g2, be2 = m
    c = (lambda t: t.detach().to(dtype))
    if dtype != F32 and fold:                 # the query projection carries scale * log2(e) (see FOLD_QSCALE above): rounded ONCE, from fp32
        wq, bq = wq.detach() * Q_FOLD, bq.detach() * Q_FOLD
    W = (torch.cat([wq, wk, wv], 0).to(dtype), torch.cat([bq, bk, bv], 0).to(dtype), c(w_ao), c(b_ao), g1.detach(), be1.detach(),
         c(w_i), c(b_i), c(w_o), c(b_o), g2.detach(), be2.detach())
    # Transposed copies [in, out] for the input-gradient GEMMs dx = dy W: handed to the library as dy @ Wt.t() ("NT", the layout of
    # the forward) they run 8-12 % faster than dy @ W ("NN") on every shape of the layer (tools/gemm_layout_probe.py: e.g.
    # [650k, 768] x [768, 3072]: 2.63 vs 3.00 ms) - the NN path gets a 256x256x32 tile, the NT path 256x256x64.
    # (T2S_DGRAD_NT=0: views instead of copies, i.e. the NN layout - for same-box A/B runs)
    W = W + tuple((W[i].t().contiguous() if _DGRAD_NT else W[i].t()) for i in (0, 2, 6, 8))
    if cache is not None:
        cache[key] = (m[0], W)                            # holding m[0] keeps its id() from being reused inside the scope
    return W


def _master_grads(g, fold=False):
    """12 gradients in the order of W -> 16 in the order of the masters (views of the fused QKV gradients).  ``lo``: the bf16 operand mode,
    whose query projection ran with W_q' = c0 W_q (FOLD_QSCALE): dL/dW_q = c0 dL/dW_q' (in place, on the Q rows of the fused gradients)."""
    dw_qkv, db_qkv = g[0], g[1]
    if fold and dw_qkv is not None:
        dw_qkv[:HID].mul_(Q_FOLD)
        db_qkv[:HID].mul_(Q_FOLD)
    return (dw_qkv[:HID], dw_qkv[HID:2 * HID], dw_qkv[2 * HID:], db_qkv[:HID], db_qkv[HID:2 * HID], db_qkv[2 * HID:]) + tuple(g[2:])


class BertLayerFn(torch.autograd.Function):
    """(y, y_lo) = BertLayer(x; keys).  x / y: fp32 residual stream; x_lo / y_lo: operand-dtype copies
    (y_lo is y itself in fp32 mode).  ``masters``: the layer's 16 fp32 parameters (layer_masters)."""

    @staticmethod
    def forward(ctx, x, x_lo, keys, dt, drop_p, seeds, attn_drop_p, *masters):
        B, L, _ = x.shape
        W = operand_weights(masters, dt, _fold(dt != F32, L))
        x2 = x.contiguous().view(B * L, HID)
        xl = (x_lo if x_lo is not None else x.to(dt)).contiguous().view(B * L, HID)
        y2, y2_lo, saved = _layer_forward(x2, xl, keys, B, L, W, drop_p, seeds, attn_drop_p, RECOMPUTE_ACTIVATIONS)
        ctx.keys = keys
        ctx.drop = (drop_p, seeds, attn_drop_p)
        ctx.lo = _fold(dt != F32, L)
        ctx.save_for_backward(*saved)
        y2 = y2.view(B, L, HID)
        y2_lo = y2_lo.view(B, L, HID) if y2_lo is not None else y2.detach()
        ctx.mark_non_differentiable(y2_lo)
        return y2, y2_lo

    @staticmethod
    def backward(ctx, dy, _dy_lo):
        B, L, _ = dy.shape
        drop_p, seeds, attn_drop_p = ctx.drop
        dx, g = _layer_backward(ctx.saved_tensors, ctx.keys, dy.contiguous().view(B * L, HID), drop_p, seeds, attn_drop_p)
        return (dx.view(B, L, HID).float(), None, None, None, None, None, None) + _master_grads(g, ctx.lo)


def _encoder_forward(ctx, x, keys, n_layers, dt, drop_p, seeds, attn_drop_p, masters, extra=(), extra_out=None):
    """A stack of layers on x [B, L, 768] fp32; what backward needs goes into ctx (``extra``: further tensors to save ahead of the
    layers' own; ``extra_out``: a list - the stack's output rows are saved too, as the first saved tensor).
    Returns the output rows [B * L, 768] fp32."""
    B, L, _ = x.shape
    flat_w = []
    for l in range(n_layers):
        flat_w.extend(operand_weights(masters[MASTERS_PER_LAYER * l:MASTERS_PER_LAYER * (l + 1)], dt, _fold(dt != F32, L)))
    x2 = x.contiguous().view(B * L, HID)
    xl = x2.to(dt) if dt != F32 else x2
    keep, counts = [], []
    for l in range(n_layers):
        x2, x_lo, saved = _layer_forward(x2, xl, keys, B, L, flat_w[W_PER_LAYER * l:W_PER_LAYER * (l + 1)], drop_p, seeds[l], attn_drop_p,
                                         RECOMPUTE_ACTIVATIONS, materialise=(l == n_layers - 1))
        xl = x_lo if x_lo is not None else x2
        counts.append([t is not None for t in saved])
        keep.extend(t for t in saved if t is not None)
    if extra_out is not None:
        extra = (x2,) + tuple(extra)
    ctx.keys, ctx.drop, ctx.counts, ctx.n_extra = keys, (drop_p, seeds, attn_drop_p), counts, len(extra)
    ctx.lo = _fold(dt != F32, L)
    ctx.save_for_backward(*extra, *keep)
    return x2


def _encoder_backward(ctx, d):
    """d: [B * L, 768] gradient of the stack's output (fp32 or the operand dtype).  Returns (dx [B * L, 768] in the operand
    dtype, the master gradients of every layer as one flat tuple)."""
    drop_p, seeds, attn_drop_p = ctx.drop
    flat = list(ctx.saved_tensors)[ctx.n_extra:]
    per_layer, pos = [], 0
    for mask in ctx.counts:
        cur = []
        for present in mask:
            cur.append(flat[pos] if present else None)
            pos += present
        per_layer.append(cur)
    grads = [None] * len(per_layer)
    for l in reversed(range(len(per_layer))):
        d, grads[l] = _layer_backward(per_layer[l], ctx.keys, d, drop_p, seeds[l], attn_drop_p)
        per_layer[l] = None
    out = ()
    for g in grads:
        out += _master_grads(g, ctx.lo)
    return d, out


class BertEncoderFn(torch.autograd.Function):
    """y = BertEncoder(x; keys) for a stack of layers as ONE autograd node.  Between layers the backward hands the
    operand-dtype dx of layer l+1 straight to the LayerNorm-backward kernel of layer l; as separate nodes autograd casts
    every layer's bf16 dx up to the fp32 dtype of that layer's input (a 2 GB cast kernel per layer at B=64) only for the
    next kernel to read it back.  Same values either way."""

    @staticmethod
    def forward(ctx, x, keys, n_layers, dt, drop_p, seeds, attn_drop_p, *masters):
        B, L, _ = x.shape
        return _encoder_forward(ctx, x, keys, n_layers, dt, drop_p, seeds, attn_drop_p, masters).view(B, L, HID)

    @staticmethod
    def backward(ctx, dy):
        B, L, _ = dy.shape
        d, g = _encoder_backward(ctx, dy.contiguous().view(B * L, HID))
        return (d.view(B, L, HID).float(), None, None, None, None, None, None) + g


class QTVFn(torch.autograd.Function):
    """y = x + tanh(BertEncoder(x; keys)): QTV.forward (t2s.py:384-432, Q5) on the concatenated [question; frames; OCR] rows as
    ONE node.  The reference slices the encoder output per modality, applies tanh and adds each slice to its input: as framework
    ops that is three slice-backward nodes (a zero fill + copy of the full [B, L1, 768] gradient each, then two adds), tanh and
    add passes, and an fp32 cast + add where the encoder's input gradient meets the residual path.  Here: one pass forward
    (t2s_tanh_residual_fwd), two backward (g_enc = gy (1 - tanh^2) in the operand dtype; dx = gy + dx_enc), gy read in place
    even when it arrives as a row slice of the MMT input gradient."""

    @staticmethod
    def forward(ctx, x, keys, n_layers, dt, drop_p, seeds, attn_drop_p, *masters):
        B, L, _ = x.shape
        # the encoder output is this node's own intermediate: it is kept through save_for_backward (freed with the other saved
        # tensors when backward has run, version-checked by autograd), written by the last LayerNorm kernel into a buffer allocated here
        enc = _encoder_forward(ctx, x, keys, n_layers, dt, drop_p, seeds, attn_drop_p, masters, extra_out=[]).view(B, L, HID)
        ctx.dt = dt
        return ops.tanh_residual_fwd(x.contiguous(), enc)

    @staticmethod
    def backward(ctx, gy):
        B, L, _ = gy.shape
        if gy.dtype != F32 or gy.stride(2) != 1 or gy.stride(1) != HID:
            gy = gy.float().contiguous()
        enc = ctx.saved_tensors[0].view(B, L, HID)
        g_enc = ops.tanh_residual_bwd(gy, enc, ctx.dt)
        del enc
        d, g = _encoder_backward(ctx, g_enc.view(B * L, HID))
        del g_enc
        return (ops.add_cast(gy, d), None, None, None, None, None, None) + g


class SharedPrefixEncoderFn(torch.autograd.Function):
    """The three MMT passes of a train step (ref / pos / neg key lists, t2s.py:293-313) over ONE sequence
    ``x = [prefix rows | decoder rows of pass 0 | of pass 1 | of pass 2]``: the prefix rows ([q; frames; OCR]) enter every
    pass unchanged, and a pass differs from the others only in its key list - the decoder rows of the other passes are never
    keys in it, so whatever they compute as queries touches nothing (their output rows are ignored, their gradients are exact
    zeros).  Shared by the passes: the one concatenated input, its operand-dtype copy and the fused QKV projection of layer 0
    (one GEMM over the rows instead of three); in backward the three passes' gradients of that projection are summed first and
    its weight / input gradients are computed once, and the input gradient of the sequence is accumulated here instead of by
    three autograd adds.  24 extra rows per sample ride along in every pass (+0.24 % of the row work at L1 = 10 120).
    Returns one [B, L, 768] fp32 output per pass (the caller reads the OCR rows and the pass's own decoder rows), followed - in
    the bf16 operand mode - by the bf16 copy of each (the last LayerNorm kernel writes both)."""

    @staticmethod
    def forward(ctx, x, keys_list, n_layers, dt, drop_p, seeds_list, attn_drop_p, *masters):
        B, L, _ = x.shape
        flat_w = []
        for l in range(n_layers):
            flat_w.extend(operand_weights(masters[MASTERS_PER_LAYER * l:MASTERS_PER_LAYER * (l + 1)], dt, _fold(dt != F32, L)))
        x2 = x.contiguous().view(B * L, HID)
        xl = x2.to(dt) if dt != F32 else x2
        qkv0 = _mm_bias(xl, flat_w[0], flat_w[1]).view(B, L, 3 * HID)
        keep, counts, outs, los = [], [], [], []
        q0c = None                  # contiguous Q third of the shared projection, for the passes that prune their keys
        for keys, seeds in zip(keys_list, seeds_list):
            c2, cl = x2, xl
            for l in range(n_layers):
                q_in, kv_in = (qkv0 if l == 0 else None), None
                if l == 0 and _prunable(keys, dt != F32):
                    # this pass sees a handful of keys: its attention reads Q from a contiguous copy (made once) and K | V from the
                    # gathered rows of its keys, so that its backward produces [B, L, 768] + [B, capK, 1536] instead of a full
                    # [B, L, 2304] buffer (zero-filled K / V thirds) that then has to be added to the other passes'
                    if q0c is None:
                        q0c = qkv0[..., :HID].contiguous()
                    _, flat_rows, capK = keys.compact(L)
                    q_in = q0c
                    kv_in = qkv0.view(B * L, 3 * HID).index_select(0, flat_rows)[:, HID:].contiguous().view(B, capK, 2 * HID)
                c2, c_lo, saved = _layer_forward(c2, cl, keys, B, L, flat_w[W_PER_LAYER * l:W_PER_LAYER * (l + 1)], drop_p, seeds[l], attn_drop_p,
                                                 RECOMPUTE_ACTIVATIONS, materialise=(l == n_layers - 1), qkv=q_in, kv_given=kv_in)
                cl = c_lo if c_lo is not None else c2
                saved = list(saved)
                if l == 0:          # xl and qkv0 are kept once (below), not per pass (a pruned pass keeps its Q copy and K | V rows)
                    saved[0] = None
                    if kv_in is None:
                        saved[1] = None
                counts.append([t is not None for t in saved])
                keep.extend(t for t in saved if t is not None)
            outs.append(c2.view(B, L, HID))
            if dt != F32:
                los.append(cl.view(B, L, HID))
        ctx.keys_list, ctx.drop, ctx.counts, ctx.n_layers = keys_list, (drop_p, seeds_list, attn_drop_p), counts, n_layers
        ctx.lo = _fold(dt != F32, L)
        ctx.save_for_backward(xl, qkv0, *keep)
        # an output nobody sent a gradient to arrives as None in backward, not as a zero tensor (each pass's gradient comes through ONE
        # of its two outputs: materialised zeros would cost a 2 GB fill, a cast and an add per pass)
        ctx.set_materialize_grads(False)
        # operand-dtype mode: the bf16 copies of the outputs follow the fp32 outputs; a consumer may send its gradient through either
        return tuple(outs) + tuple(los)

    @staticmethod
    def backward(ctx, *dys):
        drop_p, seeds_list, attn_drop_p = ctx.drop
        n_layers = ctx.n_layers
        flat = list(ctx.saved_tensors)
        xl, qkv0 = flat[0], flat[1]
        B, L, _ = qkv0.shape
        pos = 2
        per = []
        for mask in ctx.counts:
            cur = []
            for present in mask:
                cur.append(flat[pos] if present else None)
                pos += present
            per.append(cur)
        w_qkv0_t = per[0][9]               # the fused QKV operand weight of layer 0, [in, out] copy (same tensor in every pass's list)
        grads = [None] * n_layers          # parameter gradients summed over the passes, per layer (12-tuples, fp32)
        dz1_sum = dqkv_sum = None
        n_pass = len(ctx.keys_list)
        for pi, (keys, seeds) in enumerate(zip(ctx.keys_list, seeds_list)):
            dy = dys[pi]
            d_lo = dys[n_pass + pi] if len(dys) > n_pass else None
            if dy is None and d_lo is None:
                continue
            if dy is None:
                dy = d_lo                    # the whole gradient of this pass came through the operand-dtype copy (PassHeadFn)
            elif d_lo is not None:
                dy = dy + d_lo.to(dy.dtype)
            d = dy.contiguous().view(B * L, HID)
            for l in reversed(range(n_layers)):
                saved = per[pi * n_layers + l]
                if l == 0:
                    saved[0] = xl
                    if saved[1] is None:
                        saved[1] = qkv0
                d, g = _layer_backward(saved, keys, d, drop_p, seeds[l], attn_drop_p, stop_at_qkv=(l == 0))
                per[pi * n_layers + l] = None
                if grads[l] is None:
                    grads[l] = list(g)
                else:                      # one multi-tensor add per layer and pass instead of ten small launches
                    pairs = [(a, t) for a, t in zip(grads[l], g) if t is not None]
                    torch._foreach_add_([a for a, _ in pairs], [t for _, t in pairs])
            dz1, dqkv = d
            # sums over the passes in the operand dtype, in place in the first pass's buffers
            dz1_sum = dz1 if dz1_sum is None else dz1_sum.add_(dz1)
            if isinstance(dqkv, tuple):          # a pruned pass: Q gradient for every row + K | V gradient of its key rows
                dq_p, dkv_p, rows_p = dqkv
                if dqkv_sum is None:
                    dqkv_sum = torch.zeros(B * L, 3 * HID, dtype=dq_p.dtype, device=dq_p.device)
                dqkv_sum[:, :HID].add_(dq_p)
                dqkv_sum[:, HID:].index_add_(0, rows_p, dkv_p)
                del dq_p, dkv_p
            else:
                dqkv_sum = dqkv if dqkv_sum is None else dqkv_sum.add_(dqkv)
            del d, dz1, dqkv
        if dqkv_sum is None:
            return (None,) * (7 + MASTERS_PER_LAYER * n_layers)
        dqkv2 = dqkv_sum.view(B * L, 3 * HID)
        grads[0][0] = _wgrad(dqkv2, xl, B)
        grads[0][1] = dqkv2.sum(0, dtype=F32)
        dx = dz1_sum.addmm_(dqkv2, w_qkv0_t.t())            # + residual branch of LN1 (in place)
        out = (dx.view(B, L, HID).float(), None, None, None, None, None, None)
        for g in grads:
            out += _master_grads(tuple(g), ctx.lo)
        return out


class LayerNormFn(torch.autograd.Function):
    """y = LN(x + res) over rows of 768 (res optional); x may be a bf16 GEMM output, y/res are fp32.
    Used outside the BERT layers: t2s.py:87-88,116-117 (embedding LayerNorms), :685-687
    (PrevPredEmbeddings), BertEmbeddings."""

    @staticmethod
    def forward(ctx, x, res, gamma, beta):
        shape = x.shape
        x2 = x.contiguous().view(-1, HID)
        r2 = res.contiguous().view(-1, HID) if res is not None else None
        y, _, z, st = ops.add_layernorm_fwd(x2, r2, gamma, beta, inplace_z=False, stream_dtype=F32)
        ctx.save_for_backward(z, st, gamma)
        ctx.has_res = res is not None
        ctx.x_dtype = x.dtype
        return y.view(shape)

    @staticmethod
    def backward(ctx, dy):
        z, st, gamma = ctx.saved_tensors
        dy2 = dy.contiguous().view(-1, HID)
        dz, _, dg, db = ops.add_layernorm_bwd(dy2.float() if dy2.dtype != F32 else dy2, z, st, gamma, out_dtype=F32)
        dz = dz.view(dy.shape)
        return dz.to(ctx.x_dtype), (dz if ctx.has_res else None), dg, db


def layer_norm(x, gamma, beta, res=None):
    """-> fp32."""
    return LayerNormFn.apply(x, res, gamma, beta)


class PtrLogitsFn(torch.autograd.Function):
    """logits = cat([fixed, q.k^T/sqrt(768) + mask01], -1)  (T2S._forward_output t2s.py:279-286 with
    OcrPtrNet.forward :648-670).  fixed: [B, D, V] fp32 classifier scores; q: [B, D, 768] fp32; k: [B, N, 768]
    operand dtype; mask01: [B, N] fp32.  The pointer scores are written by the HIP kernel straight into the
    concatenated buffer."""

    @staticmethod
    def forward(ctx, fixed, q, k, mask01):
        B, D, V = fixed.shape
        N = k.shape[1]
        logits = torch.empty(B, D, V + N, dtype=torch.float32, device=fixed.device)
        logits[:, :, :V] = fixed
        ops.ptr_scores(q.contiguous(), k.contiguous(), mask01, logits, V, exact_fp32=(k.dtype == F32))
        ctx.save_for_backward(q, k)
        ctx.V = V
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        q, k = ctx.saved_tensors
        V = ctx.V
        scale = 1.0 / (HID ** 0.5)
        ds = dlogits[:, :, V:]                                            # [B, D, N] fp32
        dsl = (ds * scale).to(k.dtype)
        dq = torch.bmm(dsl, k).float()                                    # [B, D, 768]
        dk = torch.bmm(dsl.transpose(1, 2), q.to(k.dtype))                # [B, N, 768]
        return dlogits[:, :, :V], dq, dk, None


class OcrTailFn(torch.autograd.Function):
    """dropout(LN_feat(a) + LN_bbox(Linear_bbox(bbox))): the tail of T2S._forward_ocr_encoding (t2s.py:221-258) as one kernel each
    way (ops.ocr_tail_fwd / _bwd): the K = 4 box projection is recomputed per row, neither LayerNorm input is copied for
    backward, the add and the dropout are in the same pass."""

    @staticmethod
    def forward(ctx, a, bbox, w_box, b_box, ga, ba, gb, bb, drop_p, seed):
        shape = a.shape
        a2 = a.contiguous().view(-1, HID)
        bbox2 = bbox.float().contiguous().view(-1, 4)
        args = [t.detach().float().contiguous() for t in (w_box, b_box, ga, ba, gb, bb)]
        out, stats = ops.ocr_tail_fwd(a2, bbox2, *args, drop_p=drop_p, drop_seed=seed)
        ctx.save_for_backward(a2, bbox2, stats, args[0], args[1], args[2], args[4])
        ctx.drop = (drop_p, seed)
        return out.view(*shape[:-1], HID)

    @staticmethod
    def backward(ctx, g):
        a2, bbox2, stats, w_box, b_box, ga, gb = ctx.saved_tensors
        g2 = g.contiguous().view(-1, HID)
        if g2.dtype != F32:
            g2 = g2.float()
        d_a, dga, dba, dgb, dbb, db_box, dw_box = ops.ocr_tail_bwd(g2, a2, bbox2, w_box, b_box, ga, gb, stats, drop_p=ctx.drop[0], drop_seed=ctx.drop[1])
        return d_a.view(g.shape), None, dw_box, db_box, dga, dba, dgb, dbb, None, None


def ocr_tail(a, bbox, box_linear, ln_feat, ln_box, drop_p=0.0):
    seed = _fresh_seed() if drop_p > 0 else 0
    return OcrTailFn.apply(a, bbox, box_linear.weight, box_linear.bias, ln_feat.weight, ln_feat.bias, ln_box.weight, ln_box.bias, float(drop_p), seed)


class PassHeadFn(torch.autograd.Function):
    """What the heads of one MMT pass read from its encoder output (t2s.py:628-631 -> :279-286, :648-670): the pointer-network
    KEYS k = Linear_key(out[:, a:b]) of the OCR rows and the decoder rows out[:, d0:d1].  ``out_lo`` is the operand-dtype copy of
    ``out`` that the last LayerNorm kernel wrote anyway: the key projection reads its OCR rows from it (one bf16 row copy instead
    of a cast of 2 GB of fp32 rows), and in backward the input gradient dk W_key goes into rows [a, b) of ONE [B, L, 768]
    operand-dtype buffer that also takes the decoder rows' gradient - instead of a bf16 -> fp32 cast, an fp32 slice copy and zero
    fills per pass.  The gradient leaves through ``out_lo`` when it is given.
    (The GEMMs run on contiguous [B * N, 768] row copies: a batched GEMM straight on the row slice - batch stride L * 768, weight
    broadcast with batch stride 0 - returned wrong values from the library at B=64, N=10 000: tools/glue_probe.py.)"""

    @staticmethod
    def forward(ctx, out, out_lo, w_key, b_key, a, b, d0, d1):
        src = out_lo if out_lo is not None else out
        B, L, _ = src.shape
        wk = w_key.detach().to(src.dtype)
        rows = src[:, a:b].reshape(B * (b - a), HID)
        k = torch.addmm(b_key.detach().to(src.dtype), rows, wk.t()).view(B, b - a, wk.shape[0])
        ctx.save_for_backward(src, wk)
        ctx.geom = (a, b, d0, d1, out_lo is not None)
        return k, out[:, d0:d1]

    @staticmethod
    def backward(ctx, g_k, g_dec):
        src, wk = ctx.saved_tensors
        a, b, d0, d1, via_lo = ctx.geom
        B, L, _ = src.shape
        g = torch.empty(B, L, HID, dtype=src.dtype, device=src.device)
        g[:, :a].zero_()
        g[:, b:].zero_()
        dw = db = None
        if g_k is not None:
            gk2 = g_k.contiguous().view(B * (b - a), wk.shape[0])
            g[:, a:b].copy_((gk2 @ wk).view(B, b - a, HID))
            dw = _wgrad(gk2, src[:, a:b].reshape(B * (b - a), HID), B)
            db = gk2.sum(0, dtype=F32)
        else:
            g[:, a:b].zero_()
        if g_dec is not None:
            g[:, d0:d1] = g_dec
        return (None, g, dw, db, None, None, None, None) if via_lo else (g, None, dw, db, None, None, None, None)


def pass_head(out, out_lo, key_linear, a, b, d0, d1):
    return PassHeadFn.apply(out, out_lo, key_linear.weight, key_linear.bias, a, b, d0, d1)


class EmbedRowsFn(torch.autograd.Function):
    """[ L2norm(f0) | L2norm(f1) | emb0[id0] | emb1[id1] ] as the GEMM operand (t2s.py:192-258).  The features are
    inputs (no gradient); the id-embedding tables receive index-add gradients."""

    @staticmethod
    def forward(ctx, f0, f1, id0, emb0, id1, emb1, dtype):
        out = ops.embed_rows(f0.float().contiguous(), f1.float().contiguous() if f1 is not None else None,
                             id0.contiguous(), emb0, id1.contiguous() if id1 is not None else None, emb1, dtype)
        ctx.save_for_backward(id0, id1 if id1 is not None else id0)
        ctx.has1 = id1 is not None
        ctx.d = f0.shape[-1] + (f1.shape[-1] if f1 is not None else 0)
        ctx.shape0 = emb0.shape
        return out

    @staticmethod
    def backward(ctx, g):
        id0, id1 = ctx.saved_tensors
        d, e = ctx.d, ctx.shape0[1]
        g2 = g.reshape(-1, g.shape[-1])
        ge0 = torch.zeros(ctx.shape0, dtype=F32, device=g.device).index_add_(0, id0.reshape(-1), g2[:, d:d + e].float())
        ge1 = None
        if ctx.has1:
            ge1 = torch.zeros(ctx.shape0, dtype=F32, device=g.device).index_add_(0, id1.reshape(-1), g2[:, d + e:d + 2 * e].float())
        return None, None, None, ge0, None, ge1, None


def embed_rows(f0, f1, id0, emb0, id1, emb1, dtype):
    return EmbedRowsFn.apply(f0, f1, id0, emb0, id1, emb1, dtype)


def ptr_logits(fixed, q, k, mask01):
    return PtrLogitsFn.apply(fixed, q, k, mask01)


class SplitRowsFn(torch.autograd.Function):
    """(x[:, a:b], x[:, b:]) of a [B, L, 768] encoder output (the OCR rows and the decoder rows of an MMT pass,
    t2s.py:628-631).  As two plain slices autograd builds the input gradient as zeros + copy, zeros + copy, add (three
    passes over a 2 GB tensor per MMT pass at B=64); here it is ONE buffer written once."""

    @staticmethod
    def forward(ctx, x, a, b):
        ctx.a, ctx.b, ctx.shape = a, b, x.shape
        return x[:, a:b], x[:, b:]

    @staticmethod
    def backward(ctx, g_mid, g_tail):
        a, b = ctx.a, ctx.b
        ref = g_mid if g_mid is not None else g_tail
        g = torch.empty(ctx.shape, dtype=ref.dtype, device=ref.device)
        g[:, :a].zero_()
        if g_mid is not None:
            g[:, a:b].copy_(g_mid)
        else:
            g[:, a:b].zero_()
        if g_tail is not None:
            g[:, b:].copy_(g_tail)
        else:
            g[:, b:].zero_()
        return g, None, None


def split_rows(x, a, b):
    return SplitRowsFn.apply(x, a, b)


def _fresh_seed():
    """64-bit seed from torch's CPU generator (follows torch.manual_seed; no device sync)."""
    return int(torch.randint(0, 2 ** 62, (1,)).item())


def bert_layer(x, x_lo, keys, lp, dtype, hidden_dropout=0.0, attn_dropout=0.0):
    """lp: a module holding one layer's parameters under the reference's names (see t2s.BertLayerParams).
    hidden_dropout: p of the dropout after the attention-output and FFN-output dense layers (training only);
    attn_dropout: p of the attention-probability dropout.
    Returns (y fp32, y_lo operand dtype)."""
    drop = hidden_dropout > 0 or attn_dropout > 0
    seeds = (_fresh_seed(), _fresh_seed(), _fresh_seed()) if drop else (0, 0, 0)
    return BertLayerFn.apply(x, x_lo, keys, dtype, float(hidden_dropout), seeds, float(attn_dropout), *layer_masters(lp))


def bert_encoder(x, keys, layers, dtype, hidden_dropout=0.0, attn_dropout=0.0):
    """x: fp32 [B, L, 768] -> fp32.  The whole stack is one autograd node (BertEncoderFn)."""
    layers = list(layers)
    masters = []
    for lp in layers:
        masters.extend(layer_masters(lp))
    drop = hidden_dropout > 0 or attn_dropout > 0
    seeds = tuple((_fresh_seed(), _fresh_seed(), _fresh_seed()) if drop else (0, 0, 0) for _ in layers)
    return BertEncoderFn.apply(x, keys, len(layers), dtype, float(hidden_dropout), seeds, float(attn_dropout), *masters)


def qtv_encoder(x, keys, layers, dtype, hidden_dropout=0.0, attn_dropout=0.0):
    """x + tanh(BertEncoder(x)) on fp32 [B, L, 768] (QTVFn)."""
    layers = list(layers)
    masters = []
    for lp in layers:
        masters.extend(layer_masters(lp))
    drop = hidden_dropout > 0 or attn_dropout > 0
    seeds = tuple((_fresh_seed(), _fresh_seed(), _fresh_seed()) if drop else (0, 0, 0) for _ in layers)
    return QTVFn.apply(x, keys, len(layers), dtype, float(hidden_dropout), seeds, float(attn_dropout), *masters)


def shared_prefix_encoder(x, keys_list, layers, dtype, hidden_dropout=0.0, attn_dropout=0.0):
    """x: fp32 [B, L1 + n_pass * D, 768] = [prefix | decoder rows of each pass]; keys_list[i]: the key list of pass i (its decoder
    rows at L1 + i * D).  Returns a list of (out fp32 [B, L, 768], out_lo = its operand-dtype copy or None) per pass
    (SharedPrefixEncoderFn)."""
    layers = list(layers)
    masters = []
    for lp in layers:
        masters.extend(layer_masters(lp))
    drop = hidden_dropout > 0 or attn_dropout > 0
    seeds_list = tuple(tuple((_fresh_seed(), _fresh_seed(), _fresh_seed()) if drop else (0, 0, 0) for _ in layers) for _ in keys_list)
    res = SharedPrefixEncoderFn.apply(x, tuple(keys_list), len(layers), dtype, float(hidden_dropout), seeds_list, float(attn_dropout), *masters)
    n = len(keys_list)
    return [(res[i], res[n + i] if len(res) > n else None) for i in range(n)]


# ------------------------------------------------------------------------------------------------
# inference-only helpers for the greedy decoder with prefix reuse (T2S eval branch, t2s.py:315-354).
# The prefix rows (question / frames / OCR) never see the decoder columns (t2s.py:574-618), so their hidden
# states -- and therefore their K/V projections in every layer -- do not depend on the decoding step.  Step 0
# runs the full sequence once and keeps each layer's fused QKV buffer; later steps recompute only the 12
# decoder rows against the cached K/V (the reference recomputes the whole [L2 x L2] pass 12 times).
def _layer_weights(lp, dtype):
    return operand_weights(layer_masters(lp), dtype)


def _layer_tail(att2, res32, w, dtype):
    """attention output [rows, 768] -> layer output (fp32 stream, operand copy)."""
    _, _, w_ao, b_ao, g1, be1, w_i, b_i, w_o, b_o, g2, be2 = w[:12]
    lo = dtype != F32
    a = _mm_bias(att2, w_ao, b_ao)
    y1, y1_lo, _, _ = ops.add_layernorm_fwd(a, res32, g1, be1, save=False, stream_dtype=F32, want_lo=lo)
    o = _mm_bias(ops.gelu_fwd(_mm_bias(y1_lo if lo else y1, w_i, b_i)), w_o, b_o)
    y2, y2_lo, _, _ = ops.add_layernorm_fwd(o, y1, g2, be2, save=False, stream_dtype=F32, want_lo=lo)
    return y2, (y2_lo if lo else y2)


@torch.no_grad()
def encoder_prefill(x, keys, layers, dtype):
    """Full-sequence inference pass.  Returns (out fp32 [B, L, 768], caches = per-layer fused QKV [B, L, 2304])."""
    B, L, _ = x.shape
    x2 = x.contiguous().view(B * L, HID)
    xl = x2.to(dtype)
    caches = []
    for lp in layers:
        w = _layer_weights(lp, dtype)
        qkv = _mm_bias(xl, w[0], w[1]).view(B, L, 3 * HID)
        att, _ = ops.attn_fwd(qkv, keys)
        caches.append(qkv)
        x2, xl = _layer_tail(att.view(B * L, HID), x2, w, dtype)
    return x2.view(B, L, HID), caches


@torch.no_grad()
def encoder_decode_rows(x_dec, caches, keys, layers, dtype, row0):
    """Recompute only the decoder rows row0.. of the sequence: x_dec fp32 [B, D, 768] (their layer-0 inputs);
    ``caches`` from encoder_prefill (the decoder rows of each cache are overwritten with this step's Q/K/V)."""
    B, D, _ = x_dec.shape
    x2 = x_dec.contiguous().view(B * D, HID)
    xl = x2.to(dtype)
    for lp, qkv in zip(layers, caches):
        w = _layer_weights(lp, dtype)
        qkv[:, row0:row0 + D] = _mm_bias(xl, w[0], w[1]).view(B, D, 3 * HID)
        att = ops.attn_fwd_rows(qkv[:, row0:row0 + D, :HID], qkv, keys)
        x2, xl = _layer_tail(att.view(B * D, HID), x2, w, dtype)
    return x2.view(B, D, HID)
