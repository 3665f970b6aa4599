"""CPU ORACLE for the T2S-QA fusion path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this module.  The product path (``vitxt_gqa_amd``) never imports it and
fails loudly when the HIP extension is missing.

This is a *restatement* (own code, plain torch CPU ops with the arithmetic spelled
out) of the reference algorithm; every function cites the reference ``file:line``
it follows (paths relative to the reference checkout).

Parity pinning: the reference holds NO tests and NO golden vectors for this path
(SURVEY.md section 4), and its BERT arithmetic lives in the un-vendored, un-pinned
third-party module ``pytorch_transformers.modeling_bert``.  The oracle is therefore
pinned against outputs of the reference itself, run in the build container through
the import shim of ``tests/golden/make_golden.py`` and committed as
``tests/golden/*.npz`` (``tests/test_oracle_golden.py`` checks them).

All functions are functional over a ``state_dict`` whose keys follow the
reference's checkpoint schema, so the same weights drive
the reference, the oracle and the HIP path.
"""
import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

LN_EPS = 1e-12          # BertConfig.layer_norm_eps default (Appendix A, Q2)
NUM_HEADS = 12          # BertConfig.num_attention_heads default (Q1)
NEG_FILL = -10000.0     # mask fill, t2s.py:416,538,612 / spatio_temporal_grounding.py:21


# ----------------------------------------------------------------------------------
# third-party BERT block (pytorch_transformers.modeling_bert; call sites t2s.py:423-427,
# 538-542, 622-626).  Standard BERT-base post-LN layer, Appendix A Q1/Q2.
# ----------------------------------------------------------------------------------
def layer_norm(x, w, b, eps=LN_EPS):
    """BertLayerNorm: biased variance, eps inside the sqrt (Q2)."""
    u = x.mean(-1, keepdim=True)
    s = ((x - u) ** 2).mean(-1, keepdim=True)
    return (x - u) / torch.sqrt(s + eps) * w + b


def gelu_erf(x):
    """BertIntermediate 'gelu' = erf form (Q1)."""
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def linear(x, w, b=None):
    y = x @ w.t()
    return y if b is None else y + b


# How bert_layer evaluates softmax(q k^T / sqrt(d) + mask) v.  "eager" spells the reference's arithmetic out (BertSelfAttention:
# scores materialised as [B, 12, L, L]); "sdpa" hands the same additive mask to torch's fused CPU attention, which never
# materialises the scores - BASELINE.md section 3 allows it for the timed CPU baseline ("or use torch SDPA CPU"), where the eager
# form needs ~100 GB of autograd state at L = 10 132.  Same function (tests/test_oracle_golden.py::test_sdpa_attention_equals_the_
# eager_form, L = 632, forward and every gradient); parity tests always run "eager".
ATTENTION_IMPL = "eager"


def bert_layer(sd, prefix, x, ext_mask):
    """One BertLayer.  ``ext_mask`` is additive, broadcastable to [B, h, Lq, Lk]."""
    p = prefix
    B, L, H = x.shape
    dh = H // NUM_HEADS

    def heads(t):
        return t.view(B, L, NUM_HEADS, dh).permute(0, 2, 1, 3)

    q = heads(linear(x, sd[p + "attention.self.query.weight"], sd[p + "attention.self.query.bias"]))
    k = heads(linear(x, sd[p + "attention.self.key.weight"], sd[p + "attention.self.key.bias"]))
    v = heads(linear(x, sd[p + "attention.self.value.weight"], sd[p + "attention.self.value.bias"]))
    if ATTENTION_IMPL == "sdpa":
        m = ext_mask.to(q.dtype)
        if m.dim() < 4:
            m = m.view((1,) * (4 - m.dim()) + tuple(m.shape))
        ctx = F.scaled_dot_product_attention(q, k, v, attn_mask=m.expand(B, -1, L, L), scale=1.0 / math.sqrt(dh))
        ctx = ctx.permute(0, 2, 1, 3).reshape(B, L, H)
    else:
        scores = q @ k.transpose(-1, -2) / math.sqrt(dh) + ext_mask
        probs = torch.softmax(scores, dim=-1)
        ctx = (probs @ v).permute(0, 2, 1, 3).reshape(B, L, H)
    a = linear(ctx, sd[p + "attention.output.dense.weight"], sd[p + "attention.output.dense.bias"])
    a = layer_norm(a + x, sd[p + "attention.output.LayerNorm.weight"], sd[p + "attention.output.LayerNorm.bias"])
    i = gelu_erf(linear(a, sd[p + "intermediate.dense.weight"], sd[p + "intermediate.dense.bias"]))
    o = linear(i, sd[p + "output.dense.weight"], sd[p + "output.dense.bias"])
    return layer_norm(o + a, sd[p + "output.LayerNorm.weight"], sd[p + "output.LayerNorm.bias"])


def bert_encoder(sd, prefix, x, ext_mask, num_layers):
    for i in range(num_layers):
        x = bert_layer(sd, "%slayer.%d." % (prefix, i), x, ext_mask)
    return x


def count_layers(sd, prefix):
    n = 0
    while "%slayer.%d.attention.self.query.weight" % (prefix, n) in sd:
        n += 1
    return n


# ----------------------------------------------------------------------------------
# mask helpers
# ----------------------------------------------------------------------------------
def get_mask(nums, max_num):
    """t2s.py:726-732 -- length -> 0/1 float mask."""
    ar = torch.arange(0, max_num).unsqueeze(0).expand(nums.size(0), -1)
    return ar.lt(nums.unsqueeze(-1)).to(torch.float32)


def get_causal_mask(n):
    """t2s.py:735-742 -- lower-triangular ones."""
    return torch.tril(torch.ones(n, n))


# ----------------------------------------------------------------------------------
# encoders
# ----------------------------------------------------------------------------------
def text_bert(sd, txt_inds, txt_mask):
    """TextBert.forward t2s.py:529-545 (+ BertEmbeddings: word+pos+type(0) -> LN)."""
    L = txt_inds.size(1)
    dt = sd["text_bert.embeddings.word_embeddings.weight"].dtype
    e = (sd["text_bert.embeddings.word_embeddings.weight"][txt_inds]
         + sd["text_bert.embeddings.position_embeddings.weight"][:L].unsqueeze(0)
         + sd["text_bert.embeddings.token_type_embeddings.weight"][0])
    e = layer_norm(e, sd["text_bert.embeddings.LayerNorm.weight"], sd["text_bert.embeddings.LayerNorm.bias"])
    ext = (1.0 - txt_mask.to(dt)).unsqueeze(1).unsqueeze(2) * NEG_FILL
    return bert_encoder(sd, "text_bert.encoder.", e, ext, count_layers(sd, "text_bert.encoder."))


def l2_normalize(x, eps=1e-12):
    """F.normalize(x, dim=-1): x / max(||x||, eps)."""
    return x / x.norm(dim=-1, keepdim=True).clamp_min(eps)


def obj_encoding(sd, video_feat, frame_id):
    """T2S._forward_obj_encoding t2s.py:192-219 (dropout omitted: parity runs use p=0)."""
    x = torch.cat([l2_normalize(video_feat), sd["frame_embeddings.weight"][frame_id]], dim=-1)
    y = linear(x, sd["linear_obj_feat_to_mmt_in.weight"], sd["linear_obj_feat_to_mmt_in.bias"])
    return layer_norm(y, sd["obj_feat_layer_norm.weight"], sd["obj_feat_layer_norm.bias"])


def ocr_encoding(sd, fasttext, phoc, temporal_id, track_id, bbox):
    """T2S._forward_ocr_encoding t2s.py:221-258."""
    x = torch.cat([l2_normalize(fasttext), l2_normalize(phoc),
                   sd["temporal_position_embeddings.weight"][temporal_id],
                   sd["track_position_embeddings.weight"][track_id]], dim=-1)
    a = layer_norm(linear(x, sd["linear_ocr_feat_to_mmt_in.weight"], sd["linear_ocr_feat_to_mmt_in.bias"]),
                   sd["ocr_feat_layer_norm.weight"], sd["ocr_feat_layer_norm.bias"])
    b = layer_norm(linear(bbox, sd["linear_ocr_bbox_to_mmt_in.weight"], sd["linear_ocr_bbox_to_mmt_in.bias"]),
                   sd["ocr_bbox_layer_norm.weight"], sd["ocr_bbox_layer_norm.bias"])
    return a + b


def qtv(sd, txt_emb, txt_mask, obj_emb, obj_mask, ocr_emb, ocr_mask):
    """QTV.forward t2s.py:384-432: encoder over [q;frames;ocr], key-padding mask identical for
    every query row, then x += tanh(out_slice) per modality (Q4, Q5)."""
    dt = txt_emb.dtype
    x = torch.cat([txt_emb, obj_emb, ocr_emb], dim=1)
    m = torch.cat([txt_mask.to(dt), obj_mask.to(dt), ocr_mask.to(dt)], dim=1)
    ext = (1.0 - m).unsqueeze(1).unsqueeze(2) * NEG_FILL
    out = bert_encoder(sd, "TransLayer.encoder.", x, ext, count_layers(sd, "TransLayer.encoder."))
    T, Fn = txt_emb.size(1), obj_emb.size(1)
    return (txt_emb + torch.tanh(out[:, :T]),
            obj_emb + torch.tanh(out[:, T:T + Fn]),
            ocr_emb + torch.tanh(out[:, T + Fn:]))


# ----------------------------------------------------------------------------------
# grounding (forward only -- no gradient reaches it)
# ----------------------------------------------------------------------------------
def question_pool(sd, q_feat, q_mask):
    """Grounding_Module q_linear + _calculate_self_attn t2s.py:453-459,472-473 (Q6):
    softmax over ALL 20 positions, then mask and renormalise."""
    qp = linear(q_feat, sd["Grounding_Module.q_linear.weight"], sd["Grounding_Module.q_linear.bias"])
    a = linear(qp, sd["Grounding_Module.self_attn.weight"], sd["Grounding_Module.self_attn.bias"]).squeeze(-1)
    a = torch.softmax(a, dim=-1) * q_mask
    a = a / (a.sum(1, keepdim=True) + 1e-12)
    return torch.bmm(a.unsqueeze(1), qp)


def attention_score(q, k, attn_mask):
    """AttentionScore.forward spatio_temporal_grounding.py:15-23 (Q7): unscaled dot, softmax
    over all M, mask, renormalise(+1e-12), fill -10000."""
    a = torch.bmm(q, k.transpose(-2, -1)).squeeze(1)
    a = torch.softmax(a, dim=-1) * attn_mask
    a = a / (a.sum(dim=-1, keepdim=True) + 1e-12)
    return torch.where(attn_mask == 0, torch.full_like(a, NEG_FILL), a)


def gumbel_hard_split(score, expo):
    """F.gumbel_softmax(cat(pos,neg)[B,2,M], tau=1, hard=True, dim=1) with the exponential
    draw ``expo`` [B,2,M] injected (Q8).  pos==neg scores, so the split is decided by the
    noise; argmax tie -> index 0 (= pos).  Returns exact 0/1 masks."""
    g = -torch.log(expo)
    y = torch.stack([score, score], dim=1) + g
    idx = y.argmax(dim=1)
    pos = (idx == 0).to(score.dtype)
    return pos, 1.0 - pos


def topk_lowest_index_first(score, k, largest):
    """Deterministic tie rule of THIS build (Q9): among equal values the lowest index wins.
    (ATen-CPU topk/sort tie order is implementation defined; parity tests therefore inject
    the reference's masks or use tie-free rows.)"""
    key = -score if largest else score
    order = torch.sort(key, dim=-1, stable=True).indices
    return order[..., :k]


def temporal_grounding(score, frame_mask, frame_id, expo, topk):
    """Temporal_Grounding_Indicator.forward spatio_temporal_grounding.py:34-68."""
    pos_m, neg_m = gumbel_hard_split(score, expo)
    pos_m = pos_m * frame_mask
    neg_m = neg_m * frame_mask
    pos_s = torch.where(pos_m == 0, torch.full_like(score, NEG_FILL), score * pos_m)
    neg_s = torch.where(neg_m == 0, torch.full_like(score, NEG_FILL), score * neg_m)
    pos_idx = topk_lowest_index_first(pos_s, topk, largest=True)
    neg_idx = topk_lowest_index_first(neg_s, topk, largest=False)
    pos_top = torch.zeros_like(score).scatter_(1, pos_idx, 1.0)
    neg_top = torch.zeros_like(score).scatter_(1, neg_idx, 1.0)
    # ground_frame in ascending frame-index order via nonzero (Q11)
    pos_f = torch.nonzero(pos_top, as_tuple=False)[:, 1].view(score.size(0), topk)
    ground_frame = torch.gather(frame_id, 1, pos_f)
    return ground_frame, pos_top, neg_top


def new_ocr_mask_from_frames(ground_frame, temporal_id):
    """Grounding_Module.forward t2s.py:486-494: frame id 0 -> 1; all OCR slots whose
    temporal id equals a grounded frame id."""
    g = torch.where(ground_frame == 0, torch.ones_like(ground_frame), ground_frame)
    eq = torch.eq(temporal_id.unsqueeze(1), g.unsqueeze(-1))      # [B, topk, N]
    return eq.any(dim=1).to(torch.float32)


def spatial_grounding(score, bbox, new_mask, expo, o_topk, frame_num, o_frame_num):
    """Spatial_Grounding_Indicator.forward spatio_temporal_grounding.py:79-142.
    pos mask: 5 per frame for ALL frames (the `* attn_mask` is commented out, :137);
    neg mask: 5 smallest per frame, times new_mask."""
    B = score.size(0)
    pos_m, neg_m = gumbel_hard_split(score, expo)
    pos_m = pos_m * new_mask
    neg_m = neg_m * new_mask
    pos_s = torch.where(pos_m == 0, torch.full_like(score, NEG_FILL), score * pos_m)
    neg_s = torch.where(neg_m == 0, torch.full_like(score, NEG_FILL), score * neg_m)
    pos_idx = topk_lowest_index_first(pos_s.view(B, frame_num, o_frame_num), o_topk, largest=True)
    neg_idx = topk_lowest_index_first(neg_s.view(B, frame_num, o_frame_num), o_topk, largest=False)
    pos_top = torch.zeros(B, frame_num, o_frame_num, dtype=score.dtype).scatter_(2, pos_idx, 1.0).view(B, -1)
    neg_top = torch.zeros(B, frame_num, o_frame_num, dtype=score.dtype).scatter_(2, neg_idx, 1.0).view(B, -1)
    neg_top = neg_top * new_mask
    box = torch.masked_select(bbox, pos_top.unsqueeze(-1).expand(B, -1, 4).bool()).view(B, -1, 4)
    return box, pos_top, neg_top


def grounding(sd, txt_emb, txt_mask, obj_emb, obj_mask, ocr_emb, frame_id, temporal_id, bbox,
              expo_frame, expo_ocr, frame_topk, ocr_topk, frame_num, o_frame_num):
    """Grounding_Module.forward t2s.py:461-518.  Returns dict of masks/outputs + scorer values."""
    gq = question_pool(sd, txt_emb, txt_mask)
    fm = obj_mask.to(txt_emb.dtype)
    f_score = attention_score(gq, obj_emb, fm)
    ground_frame, pos_f, neg_f = temporal_grounding(f_score, fm, frame_id, expo_frame, frame_topk)
    pos_f = pos_f * fm
    neg_f = neg_f * fm
    new_mask = new_ocr_mask_from_frames(ground_frame, temporal_id).to(txt_emb.dtype)
    o_score = attention_score(gq, ocr_emb, new_mask)
    box, pos_o, neg_o = spatial_grounding(o_score, bbox, new_mask, expo_ocr, ocr_topk, frame_num, o_frame_num)
    return dict(global_q=gq, frame_score=f_score, ocr_score=o_score, ground_frame=ground_frame,
                ground_box=box, new_ocr_mask=new_mask, pos_obj_mask=pos_f, neg_obj_mask=neg_f,
                pos_ocr_mask=pos_o, neg_ocr_mask=neg_o)


# ----------------------------------------------------------------------------------
# MMT + heads
# ----------------------------------------------------------------------------------
def prev_pred_embeddings(sd, ans_emb, ocr_emb, prev_inds):
    """PrevPredEmbeddings.forward t2s.py:690-723 + _batch_gather :745-757 (Q13)."""
    p = "mmt.prev_pred_embeddings."
    B, D = prev_inds.shape
    V = ans_emb.size(0)
    ans = layer_norm(ans_emb, sd[p + "ans_layer_norm.weight"], sd[p + "ans_layer_norm.bias"])
    ocr = layer_norm(ocr_emb, sd[p + "ocr_layer_norm.weight"], sd[p + "ocr_layer_norm.bias"])
    cat = torch.cat([ans.unsqueeze(0).expand(B, -1, -1), ocr], dim=1)
    raw = torch.gather(cat, 1, prev_inds.unsqueeze(-1).expand(-1, -1, cat.size(-1)))
    pos = sd[p + "position_embeddings.weight"][:D].unsqueeze(0).expand(B, -1, -1)
    typ = sd[p + "token_type_embeddings.weight"][prev_inds.ge(V).long()]
    emb = layer_norm(pos + typ, sd[p + "emb_layer_norm.weight"], sd[p + "emb_layer_norm.bias"])
    return raw + emb


def mmt(sd, txt_emb, txt_mask, obj_emb, obj_mask, ocr_emb, ocr_mask, prev_inds, keep=None):
    """MMT.forward t2s.py:556-633: [q;frames;ocr;dec], prefix-LM mask (Q4).  ``keep``: optional dict that receives the
    decoder-step embeddings (``dec_emb``)."""
    dt = txt_emb.dtype
    dec = prev_pred_embeddings(sd, sd["classifier.module.weight"], ocr_emb, prev_inds)
    if keep is not None:
        keep["dec_emb"] = dec
    B, D = prev_inds.shape
    x = torch.cat([txt_emb, obj_emb, ocr_emb, dec], dim=1)
    m = torch.cat([txt_mask.to(dt), obj_mask.to(dt), ocr_mask.to(dt), torch.zeros(B, D, dtype=dt)], dim=1)
    L = x.size(1)
    ext = m.unsqueeze(1).unsqueeze(2).repeat(1, 1, L, 1)
    ext[:, :, -D:, -D:] = get_causal_mask(D).to(dt)
    ext = (1.0 - ext) * NEG_FILL
    out = bert_encoder(sd, "mmt.encoder.", x, ext, count_layers(sd, "mmt.encoder."))
    T, Fn, N = txt_emb.size(1), obj_emb.size(1), ocr_emb.size(1)
    return out[:, T + Fn:T + Fn + N], out[:, -D:]


def ocr_ptr_net(sd, dec_out, ocr_out, mask01):
    """OcrPtrNet.forward t2s.py:648-670 (Q12): q.k^T/sqrt(768) + RAW 0/1 mask."""
    q = linear(dec_out, sd["ocr_ptr_net.query.weight"], sd["ocr_ptr_net.query.bias"])
    k = linear(ocr_out, sd["ocr_ptr_net.key.weight"], sd["ocr_ptr_net.key.bias"])
    s = q @ k.transpose(-1, -2) / math.sqrt(q.size(-1))
    return s + mask01.to(s.dtype).unsqueeze(1)


def forward_output(sd, ocr_out, dec_out, mask01):
    """T2S._forward_output t2s.py:279-286."""
    fixed = linear(dec_out, sd["classifier.module.weight"], sd["classifier.module.bias"])
    return torch.cat([fixed, ocr_ptr_net(sd, dec_out, ocr_out, mask01)], dim=-1)


# ----------------------------------------------------------------------------------
# whole forward
# ----------------------------------------------------------------------------------
def t2s_forward(sd: Dict[str, torch.Tensor], s: dict, cfg: dict, training: bool = True,
                expo_frame: Optional[torch.Tensor] = None, expo_ocr: Optional[torch.Tensor] = None,
                inject_masks: Optional[dict] = None, bos_idx: int = 1, keep: bool = False):
    """T2S.forward t2s.py:153-175 (train branch :288-313, eval greedy decode :315-354).

    ``s``: dict with the 12 sample_list fields.  ``cfg``: frame_topk, ocr_topk, frame_num,
    ocr_frame_num.  ``expo_*``: the injected exponential draws of the two gumbel_softmax
    calls; ``inject_masks``: optional dict overriding pos/neg obj/ocr masks (+ground_frame,
    ground_box) with the reference's own selection (tie-breaking, Q9)."""
    txt_mask = get_mask(s["text_len"], s["text"].size(1))
    txt_emb = text_bert(sd, s["text"], txt_mask)
    obj_in = obj_encoding(sd, s["video_feat"], s["frame_id"])
    ocr_in = ocr_encoding(sd, s["context_feature_0"], s["context_feature_1"], s["temporal_id"],
                          s["track_id"], s["ocr_bbox_coordinates"])
    inter = dict(txt_emb0=txt_emb, obj_in0=obj_in, ocr_in0=ocr_in)
    obj_mask, ocr_mask = s["frame_mask"], s["ocr_mask"]
    txt_emb, obj_in, ocr_in = qtv(sd, txt_emb, txt_mask, obj_in, obj_mask, ocr_in, ocr_mask)
    inter.update(txt_emb=txt_emb, obj_in=obj_in, ocr_in=ocr_in)

    if expo_frame is None:      # sampled as the reference does (F.gumbel_softmax: exponential_())
        expo_frame = torch.empty(obj_in.size(0), 2, obj_in.size(1)).exponential_()
        expo_ocr = torch.empty(ocr_in.size(0), 2, ocr_in.size(1)).exponential_()
    with torch.no_grad():
        g = grounding(sd, txt_emb.detach(), txt_mask, obj_in.detach(), obj_mask, ocr_in.detach(),
                      s["frame_id"], s["temporal_id"], s["ocr_bbox_coordinates"],
                      expo_frame.to(txt_emb.dtype), expo_ocr.to(txt_emb.dtype),
                      cfg["frame_topk"], cfg["ocr_topk"], cfg["frame_num"], cfg["ocr_frame_num"])
    if inject_masks is not None:
        g.update({k: v for k, v in inject_masks.items()})
        # the outputs the reference derives from its masks follow the injected ones (spatio_temporal_grounding.py:65-66:
        # frame ids at the nonzero positions of the pos frame mask, ascending; :139-140: boxes under the pos OCR mask)
        if "ground_frame" not in inject_masks and "pos_obj_mask" in inject_masks:
            pf = torch.nonzero(g["pos_obj_mask"], as_tuple=False)[:, 1].view(obj_in.size(0), -1)
            g["ground_frame"] = torch.gather(s["frame_id"], 1, pf)
        if "ground_box" not in inject_masks and "pos_ocr_mask" in inject_masks:
            B_ = ocr_in.size(0)
            g["ground_box"] = torch.masked_select(s["ocr_bbox_coordinates"],
                                                  g["pos_ocr_mask"].unsqueeze(-1).expand(B_, -1, 4).bool()).view(B_, -1, 4)
    inter.update(g)

    def three_pass(prev_inds):
        out = {}
        for name, om, cm in (("ref", obj_mask, ocr_mask),
                             ("pos", g["pos_obj_mask"], g["pos_ocr_mask"]),
                             ("neg", g["neg_obj_mask"], g["neg_ocr_mask"])):
            kp = {}
            ocr_out, dec_out = mmt(sd, txt_emb, txt_mask, obj_in, om, ocr_in, cm, prev_inds, keep=kp)
            out[name + "_scores"] = forward_output(sd, ocr_out, dec_out, cm)
            inter[name + "_mmt_ocr"], inter[name + "_mmt_dec"], inter[name + "_dec_emb"] = ocr_out, dec_out, kp["dec_emb"]
        return out

    if training:
        res = three_pass(s["train_prev_inds"].clone())
    else:
        prev = torch.zeros_like(s["train_prev_inds"])
        prev[:, 0] = bos_idx
        for _ in range(prev.size(1)):
            res = three_pass(prev)
            prev[:, 1:] = res["pos_scores"].argmax(dim=-1)[:, :-1]
        res["prev_inds"] = prev
    res.update(ground_box=g["ground_box"], ground_frame=g["ground_frame"],
               frame_topk=torch.tensor(cfg["frame_topk"]), ocr_topk=torch.tensor(cfg["ocr_topk"]))
    if keep:
        res["_inter"] = inter
    return res


# ----------------------------------------------------------------------------------
# losses  (pythia/modules/losses.py)
# ----------------------------------------------------------------------------------
def pos_bce_loss(pos_scores, targets, loss_mask):
    """POSBCEWithMaskLoss.forward losses.py:329-343."""
    x, t = pos_scores, targets.to(pos_scores.dtype)
    l = torch.clamp(x, min=0) - x * t + torch.log1p(torch.exp(-x.abs()))
    l = l * loss_mask.to(x.dtype).unsqueeze(-1)
    cnt = torch.clamp(loss_mask.to(x.dtype).sum(), min=1.0)
    return l.sum() / cnt


def info_nce(ref, pos, neg, temperature=0.1):
    """InfoNCE.forward losses.py:361-385: normalise last dim, flatten, cosine(ref,pos) and
    cosine(ref,neg), CE([p,n]/0.1, label 0), mean over batch (Q15)."""
    B = ref.size(0)

    def cos(a, b):          # F.cosine_similarity(dim=1), eps=1e-8 on each norm
        an = a.norm(dim=1).clamp_min(1e-8)
        bn = b.norm(dim=1).clamp_min(1e-8)
        return (a * b).sum(1) / (an * bn)

    q = l2_normalize(ref).reshape(B, -1)
    p = l2_normalize(pos).reshape(B, -1)
    n = l2_normalize(neg).reshape(B, -1)
    logits = torch.stack([cos(q, p), cos(q, n)], dim=1) / temperature
    return (torch.logsumexp(logits, dim=1) - logits[:, 0]).mean()


def total_loss(res, targets, loss_mask, w_bce=1.0, w_nce=1000.0):
    """PythiaLoss weight (losses.py:158-173) + trainer sum of .mean() (base_trainer.py:274-278)."""
    a = w_bce * pos_bce_loss(res["pos_scores"], targets, loss_mask)
    b = w_nce * info_nce(res["ref_scores"], res["pos_scores"], res["neg_scores"])
    return a + b, a, b


# ----------------------------------------------------------------------------------
# optimiser step (reference semantics: clip_grad_norm_(0.25) -> Adam, base_trainer.py:262-272)
# ----------------------------------------------------------------------------------
DEAD_PREFIXES = ("Grounding_Module.encoder.", "Grounding_Module.frame_attn.",
                 "Grounding_Module.frame_grounding_indicator.", "Grounding_Module.ocr_grounding_indicator.",
                 "Grounding_Module.q_linear.", "Grounding_Module.self_attn.",
                 "linear_obj_frame_to_mmt_in.", "obj_frame_layer_norm.")


def is_dead(name):
    """Parameters that never receive a gradient (Q14)."""
    return name.startswith(DEAD_PREFIXES)


def lr_lambda(i_iter, warmup_iterations=1000, warmup_factor=0.2, lr_steps=(10000, 20000), lr_ratio=0.1):
    """lr_lambda_update general.py:20-29."""
    if i_iter <= warmup_iterations:
        alpha = float(i_iter) / float(warmup_iterations)
        return warmup_factor * (1.0 - alpha) + alpha
    import bisect
    return pow(lr_ratio, bisect.bisect(list(lr_steps), i_iter))


def train_step(sd, s, cfg, adam_state, step, lr=1e-4, eps=1e-8, betas=(0.9, 0.999), max_norm=0.25,
               expo_frame=None, expo_ocr=None, inject_masks=None, w_nce=1000.0):
    """One reference train step on leaf tensors in ``sd`` (requires_grad set by caller):
    forward, losses, backward, clip_grad_norm_(0.25), Adam(wd 0).  Returns (loss, grad_norm)."""
    res = t2s_forward(sd, s, cfg, training=True, expo_frame=expo_frame, expo_ocr=expo_ocr,
                      inject_masks=inject_masks)
    loss, a, b = total_loss(res, s["targets"], s["train_loss_mask"], w_nce=w_nce)
    params = [(k, v) for k, v in sd.items() if v.requires_grad]
    grads = torch.autograd.grad(loss, [v for _, v in params], allow_unused=True)
    sq = sum((g.double() ** 2).sum() for g in grads if g is not None)
    gnorm = float(sq.sqrt())
    coef = min(1.0, max_norm / (gnorm + 1e-6))
    with torch.no_grad():
        for (k, p), g in zip(params, grads):
            if g is None:
                continue
            g = g * coef
            m, v = adam_state.setdefault(k, (torch.zeros_like(p), torch.zeros_like(p)))
            m.mul_(betas[0]).add_(g, alpha=1 - betas[0])
            v.mul_(betas[1]).addcmul_(g, g, value=1 - betas[1])
            bc1 = 1 - betas[0] ** step
            bc2 = 1 - betas[1] ** step
            p.addcdiv_(m, (v.sqrt() / math.sqrt(bc2)).add_(eps), value=-lr / bc1)
    return float(loss.detach()), gnorm, float(a.detach()), float(b.detach())
