"""MI355X-native T2S model behind the reference's model-registry surface.

Drop-in counterpart of ``pythia/models/t2s.py`` (``@registry.register_model("t2s") class T2S(BaseModel)``,
:21-376): same constructor/config, ``build()``, ``forward(sample_list) -> dict`` with the 7 keys of
:165-173, ``get_optimizer_parameters(config)`` (:356-376) and the same ``state_dict()`` key schema
(SURVEY.md Appendix D), so checkpoints are interchangeable.  The arithmetic runs on the HIP kernels
of libt2s_hip.so (attention, LayerNorm, GELU, ...) plus library GEMMs; there is no CPU path.

Differences that are deliberate and documented:
  * the [B,1,L,L] additive masks are replaced by compacted key lists (exact: exp(-10000+..) == 0 in fp32);
  * top-k / sort ties are broken lowest-index-first (the reference's ATen order is implementation
    defined, Appendix A Q9); ``sample_list.grounding_noise`` / ``sample_list.grounding_masks`` allow
    the gumbel draws or the masks themselves to be injected for parity tests;
  * dead parameters (Q14) are kept in the state_dict but frozen, so DDP needs no unused-parameter scan.
"""
import math

import os

import torch
import torch.nn.functional as F
from torch import nn

from .gemm_tuning import enable_tuned_gemms
from . import functional as FN
from . import ops
from .base_model import BaseModel
from .registry import registry
from .schema import is_dead_param

HID = 768
LN_EPS = 1e-12


# ------------------------------------------------------------------------------------------------
# parameter holders named exactly like the reference's module tree
# ------------------------------------------------------------------------------------------------
class _Holder(nn.Module):
    pass


class BertLayerParams(nn.Module):
    def __init__(self, hidden=HID, ffn=4 * HID):
        super().__init__()
        self.attention = _Holder()
        att_self = _Holder()
        att_self.query = nn.Linear(hidden, hidden)
        att_self.key = nn.Linear(hidden, hidden)
        att_self.value = nn.Linear(hidden, hidden)
        self.attention.add_module("self", att_self)
        self.attention.output = _Holder()
        self.attention.output.dense = nn.Linear(hidden, hidden)
        self.attention.output.LayerNorm = nn.LayerNorm(hidden, eps=LN_EPS)
        self.intermediate = _Holder()
        self.intermediate.dense = nn.Linear(hidden, ffn)
        self.output = _Holder()
        self.output.dense = nn.Linear(ffn, hidden)
        self.output.LayerNorm = nn.LayerNorm(hidden, eps=LN_EPS)


class BertEncoderParams(nn.Module):
    def __init__(self, num_layers):
        super().__init__()
        self.layer = nn.ModuleList([BertLayerParams() for _ in range(num_layers)])


def _dropout_cfg(module, config):
    """BertConfig defaults (Appendix A, Q1): hidden_dropout_prob = attention_probs_dropout_prob = 0.1 unless the yml
    section overrides them.  Hidden dropout is fused into the residual+LayerNorm kernels, attention-probability
    dropout into the attention kernels."""
    module.hidden_dropout = float(config.get("hidden_dropout_prob", 0.1))
    module.attn_dropout = float(config.get("attention_probs_dropout_prob", 0.1))


def _train_dropout(module):
    """(hidden, attention) dropout probabilities in effect: the configured ones in training, 0 in eval."""
    if not module.training:
        return 0.0, 0.0
    return module.hidden_dropout, module.attn_dropout


def _bert_init(module):
    """BertPreTrainedModel.init_weights (Appendix A, Q3): N(0, 0.02), bias 0, LayerNorm (1, 0)."""
    for m in module.modules():
        if isinstance(m, (nn.Linear, nn.Embedding)):
            m.weight.data.normal_(mean=0.0, std=0.02)
        if isinstance(m, nn.LayerNorm):
            m.bias.data.zero_()
            m.weight.data.fill_(1.0)
        if isinstance(m, nn.Linear) and m.bias is not None:
            m.bias.data.zero_()


class TextBert(nn.Module):
    """t2s.py:521-545 (BertEmbeddings + 3-layer BertEncoder over L=20)."""

    def __init__(self, config):
        super().__init__()
        vocab = config.get("vocab_size", 30522)
        self.embeddings = _Holder()
        self.embeddings.word_embeddings = nn.Embedding(vocab, HID)
        self.embeddings.position_embeddings = nn.Embedding(512, HID)
        self.embeddings.token_type_embeddings = nn.Embedding(2, HID)
        self.embeddings.LayerNorm = nn.LayerNorm(HID, eps=LN_EPS)
        self.encoder = BertEncoderParams(config.get("num_hidden_layers", 12))
        _dropout_cfg(self, config)
        _bert_init(self)

    def load_pretrained(self, state_dict, strict=True):
        """``TextBert.from_pretrained('bert-base-uncased', config=...)`` of the reference (t2s.py:47-56; the loader is the
        third-party ``BertPreTrainedModel.from_pretrained``): from a Hugging Face BERT checkpoint's state_dict take the
        embeddings and the FIRST ``num_hidden_layers`` encoder layers (3 in configs/t2s_abinet.yml); layers 3..11, the pooler
        and the pre-training heads are ignored.  Accepted key spellings: with or without the ``bert.`` prefix, and the
        old TF-style ``LayerNorm.gamma / .beta``.  Returns the list of checkpoint keys that were used."""
        own = self.state_dict()
        picked = {}
        for k, v in state_dict.items():
            k2 = k[5:] if k.startswith("bert.") else k
            if k2.endswith("LayerNorm.gamma"):
                k2 = k2[:-5] + "weight"
            elif k2.endswith("LayerNorm.beta"):
                k2 = k2[:-4] + "bias"
            if k2 in own:
                if tuple(v.shape) != tuple(own[k2].shape):
                    raise ValueError("pretrained tensor %s has shape %s, text_bert expects %s" % (k, tuple(v.shape), tuple(own[k2].shape)))
                picked[k2] = v
        missing = [k for k in own if k not in picked]
        if missing and strict:
            raise KeyError("pretrained BERT state_dict lacks %d text_bert tensors, e.g. %s" % (len(missing), missing[:4]))
        self.load_state_dict(picked, strict=False)
        return sorted(picked)

    def forward(self, txt_inds, txt_mask, dtype):
        e = self.embeddings
        L = txt_inds.size(1)
        x = (e.word_embeddings(txt_inds) + e.position_embeddings.weight[:L].unsqueeze(0)
             + e.token_type_embeddings.weight[0])
        x = FN.layer_norm(x, e.LayerNorm.weight, e.LayerNorm.bias)
        pd, pa = _train_dropout(self)
        if pd > 0:
            x = F.dropout(x, pd, True)                      # BertEmbeddings dropout
        keys = ops.compact_keys(txt_mask > 0)
        return FN.bert_encoder(x, keys, self.encoder.layer, dtype, pd, pa)


class QTV(nn.Module):
    """t2s.py:378-432."""

    def __init__(self, config):
        super().__init__()
        self.encoder = BertEncoderParams(config.get("num_hidden_layers", 12))
        _dropout_cfg(self, config)
        _bert_init(self)

    def forward(self, fwd, dtype):
        txt, obj, ocr = fwd["txt_emb"], fwd["obj_mmt_in"], fwd["ocr_mmt_in"]
        x = torch.cat([txt, obj, ocr], dim=1)
        valid = torch.cat([fwd["txt_mask"] > 0, fwd["obj_mask"] > 0, fwd["ocr_mask"] > 0], dim=1)
        # x + tanh(encoder(x)) per modality (t2s.py:428-432) as one node on the concatenated rows; the three results are row
        # slices of ONE [B, L1, 768] tensor, which the MMT passes take whole (``qtv_out``) as the prefix of their sequence
        y = FN.qtv_encoder(x, ops.compact_keys(valid), self.encoder.layer, dtype, *_train_dropout(self))
        T, Fn = txt.size(1), obj.size(1)
        fwd["qtv_out"] = y
        fwd["txt_emb"], fwd["obj_mmt_in"], fwd["ocr_mmt_in"] = y[:, :T], y[:, T:T + Fn], y[:, T + Fn:]


class _AttentionScoreParams(nn.Module):
    """AttentionScore (spatio_temporal_grounding.py:6-23): linear_q / linear_k exist but are unused."""

    def __init__(self, hidden):
        super().__init__()
        self.linear_q = nn.Linear(hidden, hidden)
        self.linear_k = nn.Linear(hidden, hidden)


class Grounding_Module(nn.Module):
    """t2s.py:434-518 + spatio_temporal_grounding.py (forward only: nothing here receives gradients)."""

    def __init__(self, grounding_config, bert_config):
        super().__init__()
        g = grounding_config
        self.frame_topk, self.ocr_topk = g.frame_topk, g.ocr_topk
        self.frame_num, self.frame_ocr_num, self.hidden_size = g.frame_num, g.ocr_frame_num, g.hidden_size
        h = self.hidden_size
        self.q_linear = nn.Linear(h, h)
        self.frame_attn = nn.Linear(h * 2, 1)
        self.self_attn = nn.Linear(h, 1)
        self.frame_grounding_indicator = _Holder()
        self.frame_grounding_indicator.frame_pos_att = _AttentionScoreParams(h)
        self.frame_grounding_indicator.frame_neg_att = _AttentionScoreParams(h)
        self.ocr_grounding_indicator = _Holder()
        self.ocr_grounding_indicator.ocr_pos_att = _AttentionScoreParams(h)
        self.ocr_grounding_indicator.ocr_neg_att = _AttentionScoreParams(h)
        self.encoder = BertEncoderParams(bert_config.get("num_hidden_layers", 12))     # dead (Q14)

    @torch.no_grad()
    def forward(self, sample_list, fwd):
        ocr_feat, frame_feat = fwd["ocr_mmt_in"].float(), fwd["obj_mmt_in"].float()
        frame_mask = fwd["obj_mask"].float()
        q_feat, q_mask = fwd["txt_emb"].float(), fwd["txt_mask"].float()
        B, N, _ = ocr_feat.shape
        Fn = frame_feat.size(1)
        noise = sample_list.get("grounding_noise", None)
        if noise is None:
            e1 = torch.empty(B, 2, Fn, device=ocr_feat.device).exponential_()
            e2 = torch.empty(B, 2, N, device=ocr_feat.device).exponential_()
        else:
            e1, e2 = noise[0].to(ocr_feat.device).float(), noise[1].to(ocr_feat.device).float()

        # question pooling, t2s.py:453-459,472-473 (Q6): projection = library GEMM, pooling = HIP kernel
        qp = F.linear(q_feat, self.q_linear.weight, self.q_linear.bias).contiguous()
        gq = ops.question_pool(qp, self.self_attn.weight.view(-1), self.self_attn.bias, q_mask)           # [B, 768]

        # stage 1 + 2: temporal scorer, gumbel split, frame top-k, OCR slots of the grounded frames, spatial
        # scorer, per-frame OCR top-k, boxes (spatio_temporal_grounding.py:15-142, t2s.py:486-494) on the HIP kernels
        P, ot, k = self.frame_ocr_num, self.ocr_topk, self.frame_topk
        assert Fn == self.frame_num and N == self.frame_num * P, "OCR slots must equal frame_num * ocr_frame_num"
        # (frame_feat / ocr_feat are row slices of QTV's one output tensor: the scorer kernels take the batch stride)
        f_score = ops.attention_score(gq, frame_feat, frame_mask.contiguous())
        sel = ops.ground_select(f_score, frame_mask.contiguous(), e1.contiguous(), sample_list.frame_id.contiguous(), gq,
                                ocr_feat, e2.contiguous(), sample_list.temporal_id.contiguous(),
                                sample_list.ocr_bbox_coordinates.float().contiguous(), Fn, P, k, ot)
        ground_frame, ground_box = sel["ground_frame"], sel["ground_box"]
        o_score, new_mask = sel["ocr_score"], sel["new_ocr_mask"]
        masks = dict(pos_obj_mask=sel["pos_obj_mask"], neg_obj_mask=sel["neg_obj_mask"],
                     pos_ocr_mask=sel["pos_ocr_mask"], neg_ocr_mask=sel["neg_ocr_mask"])
        inject = sample_list.get("grounding_masks", None)
        if inject is not None:
            masks.update({k_: v.to(ocr_feat.device).float() for k_, v in inject.items() if k_ in masks})
            # the MMT passes rely on the cardinalities the top-k selection guarantees (static key bounds, K / V projected for the
            # key rows only): injected masks (tests) must obey them too - checked here, on the host, because nothing else would
            lim = {"pos_obj_mask": k, "neg_obj_mask": k, "pos_ocr_mask": min(P, ot) * Fn, "neg_ocr_mask": min(P, ot) * k}
            for k_, cap_ in lim.items():
                if k_ in inject and int((masks[k_] > 0).sum(1).max()) > cap_:
                    raise ValueError("injected %s selects more than %d entries in a sample" % (k_, cap_))
            # the outputs the reference derives from its masks follow the injected masks (spatio_temporal_grounding.py:65-66:
            # frame ids at the nonzero positions of the pos frame mask, ascending; :139-140: the boxes under the pos OCR mask)
            if "ground_frame" in inject:
                ground_frame = inject["ground_frame"].to(ocr_feat.device)
            elif "pos_obj_mask" in inject:
                ground_frame = torch.gather(sample_list.frame_id, 1, torch.nonzero(masks["pos_obj_mask"])[:, 1].view(B, -1))
            if "ground_box" in inject:
                ground_box = inject["ground_box"].to(ocr_feat.device)
            elif "pos_ocr_mask" in inject:
                ground_box = torch.masked_select(sample_list.ocr_bbox_coordinates.float(),
                                                 masks["pos_ocr_mask"].unsqueeze(-1).expand(B, -1, 4) > 0).view(B, -1, 4)
        fwd["ground_frame"] = ground_frame
        fwd["ground_bbox"] = ground_box
        fwd["frame_topk"] = torch.tensor(self.frame_topk, device=frame_feat.device)
        fwd["ocr_topk"] = torch.tensor(self.ocr_topk, device=ocr_feat.device)
        fwd["frame_score"], fwd["ocr_score"], fwd["new_ocr_mask"], fwd["global_q"] = f_score, o_score, new_mask, gq
        fwd.update(masks)
        # features are NOT gated, only masks differ (t2s.py:510-518)
        for p in ("pos", "neg"):
            fwd[p + "_obj_mmt_in"] = fwd["obj_mmt_in"]
            fwd[p + "_ocr_mmt_in"] = fwd["ocr_mmt_in"]


class PrevPredEmbeddings(nn.Module):
    """t2s.py:673-723."""

    def __init__(self):
        super().__init__()
        self.position_embeddings = nn.Embedding(100, HID)
        self.token_type_embeddings = nn.Embedding(5, HID)
        self.ans_layer_norm = nn.LayerNorm(HID, eps=LN_EPS)
        self.ocr_layer_norm = nn.LayerNorm(HID, eps=LN_EPS)
        self.emb_layer_norm = nn.LayerNorm(HID, eps=LN_EPS)

    def forward(self, ans_emb, ocr_emb, prev_inds, dtype, emb_dropout=0.0, draws=None, ocr_row0=0):
        """``draws`` = n: a list of n results that share the gathered rows and differ in the dropout draw of the position /
        type embedding only - the three MMT passes of a train step call this with the same inputs (t2s.py:293-313), and each
        separate call costs a [B, N, 768] zero fill + scatter + accumulation in the backward of its OCR-row gather."""
        # LayerNorm is row-wise, so LN(table)[gather] == LN(table[gather]): only the 12 gathered rows per
        # sample are normalised instead of the whole [V, 768] table + [B, N, 768] OCR tensor (t2s.py:702-709).
        B, D = prev_inds.shape
        V = ans_emb.size(0)
        is_ocr = prev_inds.ge(V)
        ans_rows = ans_emb[prev_inds.clamp(max=V - 1)]                                            # [B, D, 768] fp32
        # ``ocr_emb`` may be the whole [question; frames; OCR] prefix with the OCR rows starting at ``ocr_row0``: gathering from the
        # tensor the passes also take whole keeps its gradient one accumulation (a gather from a row slice adds a slice-backward
        # node: a zero fill + copy of the full gradient)
        ocr_idx = (prev_inds - V).clamp(min=0) + ocr_row0
        ocr_rows = torch.gather(ocr_emb, 1, ocr_idx.unsqueeze(-1).expand(-1, -1, HID))
        ans_n = FN.layer_norm(ans_rows, self.ans_layer_norm.weight, self.ans_layer_norm.bias)
        ocr_n = FN.layer_norm(ocr_rows, self.ocr_layer_norm.weight, self.ocr_layer_norm.bias)
        raw = torch.where(is_ocr.unsqueeze(-1), ocr_n, ans_n)
        emb = self.position_embeddings.weight[:D].unsqueeze(0) + self.token_type_embeddings(is_ocr.long())
        emb = FN.layer_norm(emb, self.emb_layer_norm.weight, self.emb_layer_norm.bias)
        if draws is not None:
            return [raw + (F.dropout(emb, emb_dropout, True) if emb_dropout > 0 else emb) for _ in range(draws)]
        if emb_dropout > 0:
            emb = F.dropout(emb, emb_dropout, True)         # emb_dropout, t2s.py:720
        return raw + emb


class MMT(nn.Module):
    """t2s.py:548-633."""

    def __init__(self, config):
        super().__init__()
        self.prev_pred_embeddings = PrevPredEmbeddings()
        self.encoder = BertEncoderParams(config.get("num_hidden_layers", 12))
        self.kept_dec_emb = None        # a list when T2S.keep_intermediates: the decoder-step embeddings of every call
        _dropout_cfg(self, config)
        _bert_init(self)

    def forward(self, txt_emb, txt_mask, obj_emb, obj_mask, ocr_emb, ocr_mask, fixed_ans_emb, prev_inds, dtype,
                max_keys=None):
        pd, pa = _train_dropout(self)
        dec_emb = self.prev_pred_embeddings(fixed_ans_emb, ocr_emb, prev_inds, dtype, pd)
        if self.kept_dec_emb is not None:
            self.kept_dec_emb.append(dec_emb)
        x = torch.cat([txt_emb, obj_emb, ocr_emb, dec_emb], dim=1)
        T, Fn, N, D = txt_emb.size(1), obj_emb.size(1), ocr_emb.size(1), dec_emb.size(1)
        L1 = T + Fn + N
        valid = torch.cat([txt_mask > 0, obj_mask > 0, ocr_mask > 0], dim=1)
        # decoder keys: step j visible to decoder row i iff i >= j; prefix rows never see them (t2s.py:574-618)
        keys = ops.compact_keys(valid, n_dec=D, dec_row0=L1, cap_hint=max_keys)
        keys.bound_is_structural = max_keys is not None          # top-k masks: the bound holds by construction (Grounding_Module)
        out = FN.bert_encoder(x, keys, self.encoder.layer, dtype, pd, pa)
        return FN.split_rows(out, T + Fn, L1)

    def forward_shared_prefix(self, txt_emb, txt_mask, obj_emb, obj_masks, ocr_emb, ocr_masks, fixed_ans_emb, prev_inds, dtype,
                              max_keys=None, prefix=None):
        """The reference's three MMT calls of one train step (ref / pos / neg masks, t2s.py:293-313) over ONE sequence
        [q; frames; OCR | dec(ref) | dec(pos) | dec(neg)]: the prefix rows are identical in the three calls, so they are
        concatenated, cast and projected to layer 0's Q/K/V once, and their input gradients are accumulated in one place
        (functional.SharedPrefixEncoderFn).  Each pass keeps its own key list, its own decoder-step rows (their dropout draw
        differs) and its own dropout seeds; the decoder rows of the other passes are never keys in it.  Returns one
        (ocr_out, dec_out) pair per pass; ``max_keys``: static bound on the visible keys of each pass (or None)."""
        pd, pa = _train_dropout(self)
        T, Fn, N = txt_emb.size(1), obj_emb.size(1), ocr_emb.size(1)
        L1 = T + Fn + N
        # ``prefix``: [q; frames; OCR] as ONE tensor (QTV's output, of which txt_emb / obj_emb / ocr_emb are row slices)
        if prefix is not None:
            decs = self.prev_pred_embeddings(fixed_ans_emb, prefix, prev_inds, dtype, pd, draws=len(obj_masks), ocr_row0=T + Fn)
        else:
            decs = self.prev_pred_embeddings(fixed_ans_emb, ocr_emb, prev_inds, dtype, pd, draws=len(obj_masks))
        if self.kept_dec_emb is not None:
            self.kept_dec_emb.extend(decs)
        D = decs[0].size(1)
        x = torch.cat(([prefix] if prefix is not None else [txt_emb, obj_emb, ocr_emb]) + decs, dim=1)
        keys = []
        for i, (om, cm) in enumerate(zip(obj_masks, ocr_masks)):
            valid = torch.cat([txt_mask > 0, om > 0, cm > 0], dim=1)
            keys.append(ops.compact_keys(valid, n_dec=D, dec_row0=L1 + i * D, cap_hint=None if max_keys is None else max_keys[i]))
            keys[-1].bound_is_structural = max_keys is not None and max_keys[i] is not None
        # per pass: (encoder output fp32 [B, L, 768], its operand-dtype copy or None, first OCR row, end of the prefix, the pass's
        # decoder rows) - the heads read it through functional.pass_head
        outs = FN.shared_prefix_encoder(x, keys, self.encoder.layer, dtype, pd, pa)
        return [(out, out_lo, T + Fn, L1, L1 + i * D, L1 + (i + 1) * D) for i, (out, out_lo) in enumerate(outs)]

    def forward_passes(self, txt_emb, txt_mask, obj_emb, obj_masks, ocr_emb, ocr_masks, fixed_ans_emb, prev_inds, dtype):
        """The reference's three MMT calls of one train step (ref / pos / neg masks, t2s.py:293-313) as ONE encoder call on
        a 3B batch: the passes share every weight and differ only in their key lists (and in the dropout draw of the
        decoder-step embeddings, kept per pass), so stacking them along the batch is the same arithmetic per sample with
        one gradient per weight instead of three accumulated ones and a third of the launches.  Returns one
        (ocr_out, dec_out) pair per pass."""
        pd, pa = _train_dropout(self)
        B, T, Fn, N = txt_emb.size(0), txt_emb.size(1), obj_emb.size(1), ocr_emb.size(1)
        L1 = T + Fn + N
        xs, valids = [], []
        for om, cm in zip(obj_masks, ocr_masks):
            dec_emb = self.prev_pred_embeddings(fixed_ans_emb, ocr_emb, prev_inds, dtype, pd)
            xs.append(torch.cat([txt_emb, obj_emb, ocr_emb, dec_emb], dim=1))
            valids.append(torch.cat([txt_mask > 0, om > 0, cm > 0], dim=1))
        D = xs[0].size(1) - L1
        keys = ops.compact_keys(torch.cat(valids, dim=0), n_dec=D, dec_row0=L1)
        out = FN.bert_encoder(torch.cat(xs, dim=0), keys, self.encoder.layer, dtype, pd, pa)
        return [(out[i * B:(i + 1) * B, T + Fn:L1], out[i * B:(i + 1) * B, L1:]) for i in range(len(xs))]


class OcrPtrNet(nn.Module):
    """t2s.py:636-670 (Q12: the RAW 0/1 mask is added to the scores)."""

    def __init__(self, hidden_size, query_key_size=None):
        super().__init__()
        self.hidden_size = hidden_size
        self.query_key_size = query_key_size or hidden_size
        self.query = nn.Linear(hidden_size, self.query_key_size)
        self.key = nn.Linear(hidden_size, self.query_key_size)

    def forward(self, query_inputs, key_inputs, attention_mask, dtype):
        assert attention_mask.dim() == 2
        # 12 query rows per sample: fp32 GEMM (tiny); N key rows per sample: operand-dtype GEMM
        q = F.linear(query_inputs, self.query.weight, self.query.bias)
        k = F.linear(key_inputs.to(dtype), self.key.weight.to(dtype), self.key.bias.to(dtype))
        return q, k                      # scored by the HIP pointer kernel inside T2S._forward_output


class _Classifier(nn.Module):
    """ClassifierLayer(type="linear") = nn.Linear(768, V) under ``.module`` (layers.py:91-108)."""

    def __init__(self, in_dim, out_dim):
        super().__init__()
        self.module = nn.Linear(in_dim, out_dim)


# ------------------------------------------------------------------------------------------------
@registry.register_model("t2s")
class T2S(BaseModel):
    def __init__(self, config):
        super().__init__(config)
        self.compute_dtype = torch.bfloat16
        cfgreg = registry.get("config")
        self._datasets = (cfgreg["datasets"] if cfgreg is not None and "datasets" in cfgreg else "vtextgqa").split(",")

    # -- construction (t2s.py:31-151) -------------------------------------------------------------
    def build(self):
        c = self.config
        self.finetune_modules = []
        self.text_bert = TextBert(c.text_bert)
        if c.get("text_bert_init_from_bert_base", False):
            # t2s.py:47-56: TextBert.from_pretrained('../../huggingface/bert-base-uncased') + the smaller learning rate.
            # The weights are read from ``text_bert_pretrained_path`` (same default path) when the file is there; offline
            # the text encoder keeps its N(0, 0.02) init and ``text_bert.load_pretrained(state_dict)`` can be called later.
            self._load_bert_base(c.get("text_bert_pretrained_path", "../../huggingface/bert-base-uncased"))
            self.finetune_modules.append({"module": self.text_bert, "lr_scale": c.lr_scale_text_bert})
        self.text_bert_out_linear = nn.Identity()
        self.frame_embeddings = nn.Embedding(4000, 50)
        self.linear_obj_feat_to_mmt_in = nn.Linear(c.obj.mmt_in_dim, HID)
        self.obj_feat_layer_norm = nn.LayerNorm(HID, eps=LN_EPS)
        self.obj_frame_layer_norm = nn.LayerNorm(HID, eps=LN_EPS)
        self.linear_obj_frame_to_mmt_in = nn.Linear(50, HID)
        self.obj_drop_p = c.obj.dropout_prob
        self.linear_ocr_feat_to_mmt_in = nn.Linear(c.ocr.mmt_in_dim, HID)
        self.linear_ocr_bbox_to_mmt_in = nn.Linear(4, HID)
        self.temporal_position_embeddings = nn.Embedding(4000, 50)
        self.track_position_embeddings = nn.Embedding(4000, 50)
        self.ocr_feat_layer_norm = nn.LayerNorm(HID, eps=LN_EPS)
        self.ocr_bbox_layer_norm = nn.LayerNorm(HID, eps=LN_EPS)
        self.ocr_drop_p = c.ocr.dropout_prob
        self.TransLayer = QTV(c.translayers)
        self.Grounding_Module = Grounding_Module(c.grounding, c.encoder)
        self.mmt = MMT(c.mmt)
        self.finetune_modules.append({"module": self.mmt, "lr_scale": c.lr_scale_mmt})
        self.ocr_ptr_net = OcrPtrNet(**c.classifier.ocr_ptr_net)
        num_choices = registry.get(self._datasets[0] + "_num_final_outputs")
        if num_choices is None:
            raise RuntimeError("registry key '%s_num_final_outputs' is not set (t2s.py:138)" % self._datasets[0])
        num_choices -= c.classifier.ocr_max_num
        self.classifier = _Classifier(HID, num_choices)
        self.answer_processor = registry.get(self._datasets[0] + "_answer_processor")
        self.decode_with_prefix_cache = True      # eval: reuse the step-invariant prefix K/V (False = reference's loop)
        # diagnostics for the parity tests: _last_fwd also keeps the pre-QTV encodings (txt_emb0 / obj_in0 / ocr_in0), and per MMT
        # pass the decoder-step embeddings and the encoder outputs ({ref,pos,neg}_dec_emb / _mmt_dec / _mmt_ocr)
        self.keep_intermediates = False
        self.batch_mmt_passes = False             # train: True = the three MMT passes as one 3B-batch encoder call (MMT.forward_passes)
        # train: the three MMT passes over one sequence with a shared prefix (MMT.forward_shared_prefix); T2S_SHARE_MMT_PREFIX=0
        # runs them as three separate encoder calls, the reference's literal structure
        self.share_mmt_prefix = os.environ.get("T2S_SHARE_MMT_PREFIX", "1") != "0"
        for n, p in self.named_parameters():
            if is_dead_param(n):
                p.requires_grad_(False)
        return self

    def _load_bert_base(self, path):
        import os
        cand = [path] if os.path.isfile(path) else [os.path.join(path, f) for f in ("pytorch_model.bin", "model.safetensors")]
        for f in cand:
            if os.path.isfile(f):
                if f.endswith(".safetensors"):
                    from safetensors.torch import load_file
                    sd = load_file(f)
                else:
                    sd = torch.load(f, map_location="cpu", weights_only=True)
                used = self.text_bert.load_pretrained(sd)
                if self.writer is not None:
                    self.writer.write("text_bert initialised from %s (%d tensors)" % (f, len(used)))
                return True
        # the reference's TextBert.from_pretrained raises when the checkpoint is missing (t2s.py:47-56); offline there is none, so
        # this is a loud warning rather than an error unless the config opts out (text_bert_allow_random_init: true)
        msg = ("text_bert_init_from_bert_base: no checkpoint under %s; text_bert keeps its RANDOM init while its learning rate "
               "is the fine-tuning one (lr_scale_text_bert) - call model.text_bert.load_pretrained(state_dict)" % path)
        if not self.config.get("text_bert_allow_random_init", False):
            import warnings
            warnings.warn(msg, RuntimeWarning, stacklevel=2)
        if self.writer is not None:
            self.writer.write(msg)
        return False

    def set_dropout(self, p):
        """Set every dropout probability of the model (hidden, attention-probability, embedding, obj/ocr input) to ``p``:
        the reference's config default is 0.1 everywhere; parity runs use 0."""
        for m in self.modules():
            if hasattr(m, "hidden_dropout"):
                m.hidden_dropout = float(p)
                m.attn_dropout = float(p)
        self.obj_drop_p = self.ocr_drop_p = float(p)
        return self

    def set_compute_dtype(self, dtype):
        assert dtype in (torch.float32, torch.bfloat16)
        self.compute_dtype = dtype
        return self

    # -- forward (t2s.py:153-354) ------------------------------------------------------------------
    def forward(self, sample_list):
        dt = self.compute_dtype
        fwd = {}
        enable_tuned_gemms()                # recorded library-GEMM selections for the benchmark shapes (gemm_tuning.py); idempotent
        with FN.shared_operands():          # one set of operand-dtype weight copies per layer for the whole forward (3 MMT passes)
            self._forward_txt_encoding(sample_list, fwd, dt)
            self._forward_obj_encoding(sample_list, fwd, dt)
            self._forward_ocr_encoding(sample_list, fwd, dt)
            if self.keep_intermediates:
                fwd["txt_emb0"], fwd["obj_in0"], fwd["ocr_in0"] = fwd["txt_emb"], fwd["obj_mmt_in"], fwd["ocr_mmt_in"]
            self.mmt.kept_dec_emb = [] if self.keep_intermediates else None
            self.TransLayer(fwd, dt)
            self.Grounding_Module(sample_list, fwd)
            self._forward_mmt_and_output(sample_list, fwd, dt)
        self._last_fwd = fwd
        return {"ref_scores": fwd["ref_scores"], "pos_scores": fwd["pos_scores"], "neg_scores": fwd["neg_scores"],
                "ground_box": fwd["ground_bbox"], "ground_frame": fwd["ground_frame"],
                "frame_topk": fwd["frame_topk"], "ocr_topk": fwd["ocr_topk"]}

    def _drop(self, x, p):
        return F.dropout(x, p, self.training) if (self.training and p > 0) else x

    def _forward_txt_encoding(self, s, fwd, dt):
        T = s.text.size(1)
        fwd["txt_mask"] = (torch.arange(T, device=s.text.device).unsqueeze(0) < s.text_len.unsqueeze(-1)).float()
        fwd["txt_emb"] = self.text_bert(s.text, fwd["txt_mask"], dt)

    def _forward_obj_encoding(self, s, fwd, dt):
        assert s.video_feat.size(-1) == 1024
        x = FN.embed_rows(s.video_feat, None, s.frame_id, self.frame_embeddings.weight, None, None, dt)
        y = F.linear(x, self.linear_obj_feat_to_mmt_in.weight.to(dt), self.linear_obj_feat_to_mmt_in.bias.to(dt))
        y = FN.layer_norm(y, self.obj_feat_layer_norm.weight, self.obj_feat_layer_norm.bias)
        fwd["obj_mmt_in"] = self._drop(y, self.obj_drop_p)
        fwd["obj_mask"] = s.frame_mask

    def _forward_ocr_encoding(self, s, fwd, dt):
        assert s.context_feature_0.size(-1) == 300 and s.context_feature_1.size(-1) == 604
        x = FN.embed_rows(s.context_feature_0, s.context_feature_1, s.temporal_id, self.temporal_position_embeddings.weight,
                          s.track_id, self.track_position_embeddings.weight, dt)
        a = F.linear(x, self.linear_ocr_feat_to_mmt_in.weight.to(dt), self.linear_ocr_feat_to_mmt_in.bias.to(dt))
        # LN_feat(a) + LN_bbox(Linear_bbox(bbox)) and the dropout in one kernel (the K = 4 box projection is computed per row, fp32)
        fwd["ocr_mmt_in"] = FN.ocr_tail(a, s.ocr_bbox_coordinates, self.linear_ocr_bbox_to_mmt_in, self.ocr_feat_layer_norm,
                                        self.ocr_bbox_layer_norm, self.ocr_drop_p if self.training else 0.0)
        fwd["ocr_mask"] = s.ocr_mask

    def _forward_output(self, ocr_out, dec_out, mask, dt):
        fixed = F.linear(dec_out, self.classifier.module.weight, self.classifier.module.bias)      # fp32, 12 rows/sample
        q, k = self.ocr_ptr_net(dec_out, ocr_out, mask, dt)
        return FN.ptr_logits(fixed, q, k, mask.float())

    def _three_pass(self, fwd, prev_inds, dt):
        g = self.Grounding_Module
        T, Fn, N = fwd["txt_emb"].size(1), fwd["obj_mmt_in"].size(1), fwd["ocr_mmt_in"].size(1)
        D = prev_inds.size(1)
        # static bounds on the number of visible keys of the pos / neg passes (Q10)
        bounds = {"ref": None,
                  "pos": T + g.frame_topk + g.ocr_topk * g.frame_num + D,
                  "neg": T + g.frame_topk + g.ocr_topk * g.frame_topk + D}
        passes = (("ref", fwd["obj_mask"], fwd["ocr_mask"]),
                  ("pos", fwd["pos_obj_mask"], fwd["pos_ocr_mask"]),
                  ("neg", fwd["neg_obj_mask"], fwd["neg_ocr_mask"]))
        if self.training and self.share_mmt_prefix and not self.batch_mmt_passes:
            outs = self.mmt.forward_shared_prefix(fwd["txt_emb"], fwd["txt_mask"], fwd["obj_mmt_in"], [p[1] for p in passes],
                                                  fwd["ocr_mmt_in"], [p[2] for p in passes], self.classifier.module.weight, prev_inds, dt,
                                                  max_keys=[bounds[p[0]] for p in passes], prefix=fwd.get("qtv_out"))
            for (name, _, cm), (out, out_lo, a, b, d0, d1) in zip(passes, outs):
                # pointer-network keys straight from the operand-dtype copy of the encoder output + the decoder rows (one node)
                k_ptr, dec_out = FN.pass_head(out, out_lo, self.ocr_ptr_net.key, a, b, d0, d1)
                fixed = F.linear(dec_out, self.classifier.module.weight, self.classifier.module.bias)      # fp32, 12 rows/sample
                q = F.linear(dec_out, self.ocr_ptr_net.query.weight, self.ocr_ptr_net.query.bias)
                fwd[name + "_scores"] = FN.ptr_logits(fixed, q, k_ptr, cm.float())
                self._keep_pass(fwd, name, out[:, a:b], dec_out, -3 + ("ref", "pos", "neg").index(name))
            return
        if self.training and self.batch_mmt_passes:
            outs = self.mmt.forward_passes(fwd["txt_emb"], fwd["txt_mask"], fwd["obj_mmt_in"], [p[1] for p in passes],
                                           fwd["ocr_mmt_in"], [p[2] for p in passes], self.classifier.module.weight, prev_inds, dt)
            for (name, _, cm), (ocr_out, dec_out) in zip(passes, outs):
                fwd[name + "_scores"] = self._forward_output(ocr_out, dec_out, cm, dt)
            return
        for name, om, cm in passes:
            ocr_out, dec_out = self.mmt(fwd["txt_emb"], fwd["txt_mask"], fwd["obj_mmt_in"], om, fwd["ocr_mmt_in"], cm,
                                        self.classifier.module.weight, prev_inds, dt, max_keys=bounds[name])
            fwd[name + "_scores"] = self._forward_output(ocr_out, dec_out, cm, dt)
            self._keep_pass(fwd, name, ocr_out, dec_out, -1)

    def _keep_pass(self, fwd, name, ocr_out, dec_out, which):
        if self.keep_intermediates:
            fwd[name + "_mmt_ocr"], fwd[name + "_mmt_dec"] = ocr_out.detach(), dec_out.detach()
            fwd[name + "_dec_emb"] = self.mmt.kept_dec_emb[which].detach()

    def _forward_mmt_and_output(self, s, fwd, dt):
        if self.training:
            fwd["prev_inds"] = s.train_prev_inds.clone()
            self._three_pass(fwd, fwd["prev_inds"], dt)
        elif not self.decode_with_prefix_cache:
            # the reference's loop verbatim (t2s.py:315-354): every step recomputes all three full MMT passes
            D = s.train_prev_inds.size(1)
            fwd["prev_inds"] = torch.zeros_like(s.train_prev_inds)
            fwd["prev_inds"][:, 0] = self.answer_processor.BOS_IDX
            for _ in range(D):
                self._three_pass(fwd, fwd["prev_inds"], dt)
                fwd["prev_inds"][:, 1:] = fwd["pos_scores"].argmax(dim=-1)[:, :-1]
        else:
            self._greedy_decode_cached(s, fwd, dt)

    @torch.no_grad()
    def _greedy_decode_cached(self, s, fwd, dt):
        """Greedy decoding (t2s.py:315-354) with prefix reuse: the prefix rows cannot attend to the decoder columns
        (t2s.py:574-618), so their hidden states are step-invariant.  Step 0 runs each of the three passes once over
        the full sequence and keeps every layer's fused QKV buffer and the pointer-network keys; steps 1..D-1
        recompute only the D decoder rows against the cached K/V.  Same outputs, ~D x fewer FLOPs."""
        g = self.Grounding_Module
        txt, obj, ocr = fwd["txt_emb"], fwd["obj_mmt_in"], fwd["ocr_mmt_in"]
        T, Fn, N = txt.size(1), obj.size(1), ocr.size(1)
        L1 = T + Fn + N
        D = s.train_prev_inds.size(1)
        prev = torch.zeros_like(s.train_prev_inds)
        prev[:, 0] = self.answer_processor.BOS_IDX
        layers = self.mmt.encoder.layer
        pn = self.ocr_ptr_net
        passes = (("ref", fwd["obj_mask"], fwd["ocr_mask"], None),
                  ("pos", fwd["pos_obj_mask"], fwd["pos_ocr_mask"], T + g.frame_topk + g.ocr_topk * g.frame_num + D),
                  ("neg", fwd["neg_obj_mask"], fwd["neg_ocr_mask"], T + g.frame_topk + g.ocr_topk * g.frame_topk + D))

        def scores(dec_out, k_ptr, cm):
            fixed = F.linear(dec_out, self.classifier.module.weight, self.classifier.module.bias)
            q = F.linear(dec_out, pn.query.weight, pn.query.bias)
            return FN.ptr_logits(fixed, q, k_ptr, cm.float())

        state = {}
        for name, om, cm, bound in passes:                       # step 0: full sequence once per pass
            dec = self.mmt.prev_pred_embeddings(self.classifier.module.weight, ocr, prev, dt)
            x = torch.cat([txt, obj, ocr, dec], dim=1)
            valid = torch.cat([fwd["txt_mask"] > 0, om > 0, cm > 0], dim=1)
            keys = ops.compact_keys(valid, n_dec=D, dec_row0=L1, cap_hint=bound)
            out, caches = FN.encoder_prefill(x, keys, layers, dt)
            k_ptr = F.linear(out[:, T + Fn:L1].to(dt), pn.key.weight.to(dt), pn.key.bias.to(dt)).contiguous()
            state[name] = (keys, caches, k_ptr, cm)
            fwd[name + "_scores"] = scores(out[:, L1:], k_ptr, cm)
        prev[:, 1:] = fwd["pos_scores"].argmax(dim=-1)[:, :-1]
        for _ in range(1, D):                                    # steps 1..D-1: decoder rows only
            dec = self.mmt.prev_pred_embeddings(self.classifier.module.weight, ocr, prev, dt)
            for name, (keys, caches, k_ptr, cm) in state.items():
                dec_out = FN.encoder_decode_rows(dec, caches, keys, layers, dt, L1)
                fwd[name + "_scores"] = scores(dec_out, k_ptr, cm)
            prev[:, 1:] = fwd["pos_scores"].argmax(dim=-1)[:, :-1]
        fwd["prev_inds"] = prev
        fwd["dec_emb_last"] = dec                                  # decoder-step embeddings of the last step (same for the 3 passes)

    # -- optimizer hook (t2s.py:356-376) -------------------------------------------------------------
    def get_optimizer_parameters(self, config):
        """Group MEMBERSHIP is the reference's (every parameter, the frozen dead ones included: they never get a gradient, so
        Adam skips them there and here), which keeps ``optimizer.state_dict()`` index-compatible with reference checkpoints."""
        groups = []
        base_lr = config.optimizer_attributes.params.lr
        finetune = set()
        for m in self.finetune_modules:
            ps = list(m["module"].parameters())
            groups.append({"params": ps, "lr": base_lr * m["lr_scale"]})
            finetune.update(ps)
        remaining = [p for p in self.parameters() if p not in finetune]
        groups.insert(0, {"params": remaining})
        return groups
