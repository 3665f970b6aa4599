"""Checkpoint key schema of the reference T2S model.

The product module tree reproduces these 220 (at default depth) tensor names exactly so
that ``state_dict()`` / ``load_state_dict()`` exchange checkpoints with the reference
(``pythia/utils/checkpoint.py:98-111`` handles an optional ``module.`` prefix).
"""
from collections import OrderedDict

HIDDEN = 768
FFN = 3072
ID_EMB = 50
ID_VOCAB = 4000


def _bert_layer(prefix, out):
    H = HIDDEN
    for n in ("query", "key", "value"):
        out[prefix + "attention.self.%s.weight" % n] = (H, H)
        out[prefix + "attention.self.%s.bias" % n] = (H,)
    out[prefix + "attention.output.dense.weight"] = (H, H)
    out[prefix + "attention.output.dense.bias"] = (H,)
    out[prefix + "attention.output.LayerNorm.weight"] = (H,)
    out[prefix + "attention.output.LayerNorm.bias"] = (H,)
    out[prefix + "intermediate.dense.weight"] = (FFN, H)
    out[prefix + "intermediate.dense.bias"] = (FFN,)
    out[prefix + "output.dense.weight"] = (H, FFN)
    out[prefix + "output.dense.bias"] = (H,)
    out[prefix + "output.LayerNorm.weight"] = (H,)
    out[prefix + "output.LayerNorm.bias"] = (H,)


def state_dict_schema(num_answers, text_vocab=30522, n_text=3, n_qtv=2, n_ground=2, n_mmt=3,
                      obj_in=1074, ocr_in=1004):
    """name -> shape, in the reference's ``state_dict()`` order (module construction order of
    ``T2S.build`` t2s.py:31-42)."""
    H = HIDDEN
    o = OrderedDict()
    o["text_bert.embeddings.word_embeddings.weight"] = (text_vocab, H)
    o["text_bert.embeddings.position_embeddings.weight"] = (512, H)
    o["text_bert.embeddings.token_type_embeddings.weight"] = (2, H)
    o["text_bert.embeddings.LayerNorm.weight"] = (H,)
    o["text_bert.embeddings.LayerNorm.bias"] = (H,)
    for i in range(n_text):
        _bert_layer("text_bert.encoder.layer.%d." % i, o)
    o["frame_embeddings.weight"] = (ID_VOCAB, ID_EMB)
    o["linear_obj_feat_to_mmt_in.weight"] = (H, obj_in)
    o["linear_obj_feat_to_mmt_in.bias"] = (H,)
    o["obj_feat_layer_norm.weight"] = (H,)
    o["obj_feat_layer_norm.bias"] = (H,)
    o["obj_frame_layer_norm.weight"] = (H,)
    o["obj_frame_layer_norm.bias"] = (H,)
    o["linear_obj_frame_to_mmt_in.weight"] = (H, ID_EMB)
    o["linear_obj_frame_to_mmt_in.bias"] = (H,)
    o["linear_ocr_feat_to_mmt_in.weight"] = (H, ocr_in)
    o["linear_ocr_feat_to_mmt_in.bias"] = (H,)
    o["linear_ocr_bbox_to_mmt_in.weight"] = (H, 4)
    o["linear_ocr_bbox_to_mmt_in.bias"] = (H,)
    o["temporal_position_embeddings.weight"] = (ID_VOCAB, ID_EMB)
    o["track_position_embeddings.weight"] = (ID_VOCAB, ID_EMB)
    o["ocr_feat_layer_norm.weight"] = (H,)
    o["ocr_feat_layer_norm.bias"] = (H,)
    o["ocr_bbox_layer_norm.weight"] = (H,)
    o["ocr_bbox_layer_norm.bias"] = (H,)
    for i in range(n_qtv):
        _bert_layer("TransLayer.encoder.layer.%d." % i, o)
    g = "Grounding_Module."
    o[g + "q_linear.weight"] = (H, H)
    o[g + "q_linear.bias"] = (H,)
    o[g + "frame_attn.weight"] = (1, 2 * H)
    o[g + "frame_attn.bias"] = (1,)
    o[g + "self_attn.weight"] = (1, H)
    o[g + "self_attn.bias"] = (1,)
    for ind, names in (("frame_grounding_indicator", ("frame_pos_att", "frame_neg_att")),
                       ("ocr_grounding_indicator", ("ocr_pos_att", "ocr_neg_att"))):
        for n in names:
            for l in ("linear_q", "linear_k"):
                o["%s%s.%s.%s.weight" % (g, ind, n, l)] = (H, H)
                o["%s%s.%s.%s.bias" % (g, ind, n, l)] = (H,)
    for i in range(n_ground):
        _bert_layer(g + "encoder.layer.%d." % i, o)
    p = "mmt.prev_pred_embeddings."
    o[p + "position_embeddings.weight"] = (100, H)
    o[p + "token_type_embeddings.weight"] = (5, H)
    for n in ("ans", "ocr", "emb"):
        o[p + "%s_layer_norm.weight" % n] = (H,)
        o[p + "%s_layer_norm.bias" % n] = (H,)
    for i in range(n_mmt):
        _bert_layer("mmt.encoder.layer.%d." % i, o)
    o["ocr_ptr_net.query.weight"] = (H, H)
    o["ocr_ptr_net.query.bias"] = (H,)
    o["ocr_ptr_net.key.weight"] = (H, H)
    o["ocr_ptr_net.key.bias"] = (H,)
    o["classifier.module.weight"] = (num_answers, H)
    o["classifier.module.bias"] = (num_answers,)
    return o


# Parameters that exist in checkpoints but never receive a gradient:
# constructed-but-unused modules, plus q_linear/self_attn which are used only upstream of the
# non-differentiable top-k selection.
DEAD_PREFIXES = ("Grounding_Module.encoder.", "Grounding_Module.frame_attn.",
                 "Grounding_Module.frame_grounding_indicator.", "Grounding_Module.ocr_grounding_indicator.",
                 "Grounding_Module.q_linear.", "Grounding_Module.self_attn.",
                 "linear_obj_frame_to_mmt_in.", "obj_frame_layer_norm.")


def is_dead_param(name):
    if name.startswith("module."):
        name = name[len("module."):]
    return name.startswith(DEAD_PREFIXES)
