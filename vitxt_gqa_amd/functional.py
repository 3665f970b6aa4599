"""Autograd functions of the T2S hot path, built on the HIP kernels (ops.py) plus library GEMMs.

One fused ``torch.autograd.Function`` per BERT layer (third-party BertLayer used by TextBert / QTV /
MMT, call sites pythia/models/t2s.py:423-427,538-542,622-626) with a hand-written backward, so that
only the tensors a flash-style backward needs are kept: layer input, fused QKV, attention output,
log-sum-exp, the two pre-LayerNorm sums (+ statistics) and the FFN pre-activation.  LayerNorm outputs
and GELU outputs are recomputed in backward (HBM-cheap) instead of stored.
"""
import torch

from . import ops

HID = ops.HID


def _mm_bias(x2, w, b):
    """x2 [rows, in] @ w[out, in]^T + b  (library GEMM with bias epilogue)."""
    return torch.addmm(b, x2, w.t())


class BertLayerFn(torch.autograd.Function):
    """y = BertLayer(x; keys).  Weights arrive already in the compute dtype (bf16 or fp32); LayerNorm
    affine parameters stay fp32."""

    @staticmethod
    def forward(ctx, x, keys, w_qkv, b_qkv, w_ao, b_ao, g1, be1, w_i, b_i, w_o, b_o, g2, be2):
        B, L, _ = x.shape
        x2 = x.reshape(B * L, HID)
        qkv = _mm_bias(x2, w_qkv, b_qkv).view(B, L, 3 * HID)
        att, lse = ops.attn_fwd(qkv, keys)
        a = _mm_bias(att.view(B * L, HID), w_ao, b_ao)
        y1, z1, st1 = ops.add_layernorm_fwd(a, x2, g1, be1)                 # z1 overwrites a
        u = _mm_bias(y1, w_i, b_i)
        gact = ops.gelu_fwd(u)
        o = _mm_bias(gact, w_o, b_o)
        del gact
        y2, z2, st2 = ops.add_layernorm_fwd(o, y1, g2, be2)                 # z2 overwrites o
        ctx.keys = keys
        ctx.save_for_backward(x2, qkv, att, lse, z1, st1, u, z2, st2, w_qkv, w_ao, g1, be1, w_i, w_o, g2)
        return y2.view(B, L, HID)

    @staticmethod
    def backward(ctx, dy):
        x2, qkv, att, lse, z1, st1, u, z2, st2, w_qkv, w_ao, g1, be1, w_i, w_o, g2 = ctx.saved_tensors
        keys = ctx.keys
        B, L, _ = qkv.shape
        dy = dy.contiguous().view(B * L, HID)
        # ---- output LayerNorm + FFN
        dz2, dg2, dbe2 = ops.add_layernorm_bwd(dy, z2, st2, g2)
        gact = ops.gelu_fwd(u)
        dw_o = dz2.t() @ gact
        db_o = dz2.sum(0)
        dgact = dz2 @ w_o
        del gact
        du, db_i = ops.gelu_bwd(dgact, u)
        del dgact
        y1, _, _ = ops.add_layernorm_fwd(z1, None, g1, be1, save=False)      # recompute LN1 output
        dw_i = du.t() @ y1
        dy1 = torch.addmm(dz2, du, w_i)                                      # + residual branch of LN2
        del du, y1, dz2
        # ---- attention output LayerNorm + projection
        dz1, dg1, dbe1 = ops.add_layernorm_bwd(dy1, z1, st1, g1)
        del dy1
        att2 = att.view(B * L, HID)
        dw_ao = dz1.t() @ att2
        db_ao = dz1.sum(0)
        datt = (dz1 @ w_ao).view(B, L, HID)
        # ---- attention
        dqkv = ops.attn_bwd(qkv, att, datt, lse, keys).view(B * L, 3 * HID)
        del datt
        dw_qkv = dqkv.t() @ x2
        db_qkv = dqkv.sum(0)
        dx = torch.addmm(dz1, dqkv, w_qkv)                                   # + residual branch of LN1
        return (dx.view(B, L, HID), None, dw_qkv, db_qkv.to(dw_qkv.dtype), dw_ao, db_ao.to(dw_ao.dtype), dg1, dbe1,
                dw_i, db_i.to(dw_i.dtype), dw_o, db_o.to(dw_o.dtype), dg2, dbe2)


class LayerNormFn(torch.autograd.Function):
    """y = LN(x + res) over rows of 768 (res optional).  Used outside the BERT layers:
    t2s.py:87-88,116-117 (embedding LayerNorms), :685-687 (PrevPredEmbeddings), BertEmbeddings."""

    @staticmethod
    def forward(ctx, x, res, gamma, beta):
        shape = x.shape
        x2 = x.contiguous().view(-1, HID)
        r2 = res.contiguous().view(-1, HID) if res is not None else None
        y, z, st = ops.add_layernorm_fwd(x2, r2, gamma, beta, inplace_z=False)
        ctx.save_for_backward(z, st, gamma)
        ctx.has_res = res is not None
        return y.view(shape)

    @staticmethod
    def backward(ctx, dy):
        z, st, gamma = ctx.saved_tensors
        dz, dg, db = ops.add_layernorm_bwd(dy.contiguous().view(-1, HID), z, st, gamma)
        dz = dz.view(dy.shape)
        return dz, (dz if ctx.has_res else None), dg, db


def layer_norm(x, gamma, beta, res=None):
    return LayerNormFn.apply(x, res, gamma, beta)


def bert_layer(x, keys, lp, dtype):
    """lp: a module holding one layer's parameters under the reference's names (see t2s.BertLayerParams)."""
    att = lp.attention
    w_qkv = torch.cat([att.self.query.weight, att.self.key.weight, att.self.value.weight], 0).to(dtype)
    b_qkv = torch.cat([att.self.query.bias, att.self.key.bias, att.self.value.bias], 0).to(dtype)
    return BertLayerFn.apply(
        x, keys, w_qkv, b_qkv,
        att.output.dense.weight.to(dtype), att.output.dense.bias.to(dtype),
        att.output.LayerNorm.weight, att.output.LayerNorm.bias,
        lp.intermediate.dense.weight.to(dtype), lp.intermediate.dense.bias.to(dtype),
        lp.output.dense.weight.to(dtype), lp.output.dense.bias.to(dtype),
        lp.output.LayerNorm.weight, lp.output.LayerNorm.bias)


def bert_encoder(x, keys, layers, dtype):
    for lp in layers:
        x = bert_layer(x, keys, lp, dtype)
    return x
