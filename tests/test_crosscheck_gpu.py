"""Cross-check of the kernels at the shapes of the reference's sibling configurations (BASELINE.json configs[4]):

  * ``transtr_abinet.yml`` / ``t2s_clipocr.yml``: the T2S dims with grounding top-k 1 / 1 (one frame, one OCR token per frame);
  * ``m4c_abinet.yml`` (pythia/models/m4c.py:185-310, 425-584): no id embeddings - 1024-d frame rows and 904-d OCR rows straight
    into the Linear + LayerNorm, ONE frame (``mid_img_feat``), a single MMT pass + classifier / pointer head, pos-BCE only.
    m4c.py reuses the TextBert / MMT / OcrPtrNet / PrevPredEmbeddings code of t2s.py verbatim, so the oracle's restatement of
    those functions is the reference here; the 1024 / 904 input stage is restated inline.
Both against the CPU oracle in fp32 mode (logits < 1e-3) and, for m4c, in bf16 mode (< 1e-2) with gradients."""
import pytest
import torch
import torch.nn.functional as F

from oracle import t2s_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def test_t2s_with_topk_1_1_matches_oracle():
    _need_gpu()
    from vitxt_gqa_amd import build_model, t2s_model_config
    from vitxt_gqa_amd.init import make_state_dict
    from vitxt_gqa_amd.schema import state_dict_schema
    from vitxt_gqa_amd.synth import make_batch, make_noise
    from vitxt_gqa_amd.testing import setup_registry, to_device
    from selection_util import decisive_frames, decisive_ocr_rows, relative_diff
    B, Fn, P, V = 2, 16, 15, 50
    setup_registry(V, Fn * P)
    cfg = t2s_model_config(frame_num=Fn, ocr_frame_num=P)
    cfg.grounding["frame_topk"], cfg.grounding["ocr_topk"] = 1, 1
    cfg.text_bert["vocab_size"] = 60
    for sec in ("text_bert", "translayers", "encoder", "mmt"):
        cfg[sec]["hidden_dropout_prob"] = cfg[sec]["attention_probs_dropout_prob"] = 0.0
    cfg.obj["dropout_prob"] = cfg.ocr["dropout_prob"] = 0.0
    cfg["losses"][1]["weight"] = 100                                  # t2s_clipocr.yml
    model = build_model(cfg)
    sd = make_state_dict(state_dict_schema(V, text_vocab=60), seed=4, attn_gain=4.0)
    model.load_state_dict(sd)
    model = model.set_compute_dtype(torch.float32).to(DEV).train()
    batch = make_batch(B, Fn, P, V=V, seed=9, text_vocab=60)
    e1, e2 = make_noise(B, Fn, P, seed=9)
    s = to_device(batch, DEV)
    s.grounding_noise = (e1, e2)
    out = model(s)
    ocfg = dict(frame_topk=1, ocr_topk=1, frame_num=Fn, ocr_frame_num=P)
    ref = O.t2s_forward({k: v.double() for k, v in sd.items()}, {k: (v.double() if v.is_floating_point() else v) for k, v in batch.items()},
                        ocfg, training=True, expo_frame=e1.double(), expo_ocr=e2.double(), keep=True)
    f, it = model._last_fwd, ref["_inter"]
    assert out["ground_frame"].shape == (B, 1) and out["ground_box"].shape == (B, Fn, 4)
    assert int(out["frame_topk"]) == 1 and int(out["ocr_topk"]) == 1
    assert decisive_frames(it["frame_score"], batch["frame_mask"].double(), e1, 1,
                           tol=max(1e-4, 4 * relative_diff(f["frame_score"].cpu(), it["frame_score"]))).all()
    assert torch.equal(out["ground_frame"].cpu(), ref["ground_frame"])
    assert torch.equal(f["pos_obj_mask"].cpu().double(), it["pos_obj_mask"]) and f["pos_obj_mask"].sum(1).tolist() == [1.0] * B
    ok = decisive_ocr_rows(it["ocr_score"], it["new_ocr_mask"], e2, 1, Fn, P, tol=max(1e-4, 4 * relative_diff(f["ocr_score"].cpu(), it["ocr_score"])))
    okn = ok.unsqueeze(-1).expand(B, Fn, P).reshape(B, Fn * P)
    assert ok.float().mean().item() > 0.8
    assert torch.equal(f["pos_ocr_mask"].cpu().double()[okn], it["pos_ocr_mask"][okn])
    assert torch.equal(f["neg_ocr_mask"].cpu().double()[okn], it["neg_ocr_mask"][okn])
    assert f["pos_ocr_mask"].sum(1).tolist() == [1.0 * Fn] * B                   # one OCR slot per frame, all frames (Q10)
    if ok.all():
        for k in ("ref_scores", "pos_scores", "neg_scores"):
            assert (out[k].double().cpu() - ref[k]).abs().max().item() < 1e-3, k
        loss, _, _ = O.total_loss(ref, batch["targets"].double(), batch["train_loss_mask"].double(), w_nce=100.0)
        got = sum(v.mean() for v in out["losses"].values()).item()
        assert abs(got - loss.item()) < 2e-3 * abs(loss.item())
    else:
        assert (out["ref_scores"].double().cpu() - ref["ref_scores"]).abs().max().item() < 1e-3


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), (torch.bfloat16, 1e-2)])
def test_m4c_shapes_single_pass_matches_oracle(dtype, tol):
    _need_gpu()
    from vitxt_gqa_amd import build_model, functional as FN, ops, t2s_model_config
    from vitxt_gqa_amd.init import make_state_dict
    from vitxt_gqa_amd.losses import POSBCEWithMaskLoss
    from vitxt_gqa_amd.schema import state_dict_schema
    from vitxt_gqa_amd.synth import make_batch
    from vitxt_gqa_amd import SampleList
    from vitxt_gqa_amd.testing import setup_registry
    B, P, V = 3, 45, 70                                   # one frame, 45 OCR tokens (m4c: up to 960)
    N = P
    setup_registry(V, N)
    cfg = t2s_model_config(frame_num=1, ocr_frame_num=P)
    cfg.obj["mmt_in_dim"], cfg.ocr["mmt_in_dim"] = 1024, 904          # m4c_abinet.yml: no id embeddings
    cfg.grounding["frame_topk"], cfg.grounding["ocr_topk"] = 1, 1
    cfg.text_bert["vocab_size"] = 60
    for sec in ("text_bert", "translayers", "encoder", "mmt"):
        cfg[sec]["hidden_dropout_prob"] = cfg[sec]["attention_probs_dropout_prob"] = 0.0
    model = build_model(cfg)
    sd = make_state_dict(state_dict_schema(V, text_vocab=60, obj_in=1024, ocr_in=904), seed=6)
    model.load_state_dict(sd)
    model = model.set_compute_dtype(dtype).to(DEV).train()
    batch = make_batch(B, 1, P, V=V, seed=13, text_vocab=60)
    batch["train_prev_inds"][:, 2] = V + 3                                   # a copied OCR token among the previous predictions
    batch["ocr_mask"][0, 30:] = 0
    dev = {k: v.to(DEV) for k, v in batch.items()}

    # ---- product path, m4c.py:185-310 with this build's kernels
    with FN.shared_operands():
        T = dev["text"].size(1)
        txt_mask = (torch.arange(T, device=DEV).unsqueeze(0) < dev["text_len"].unsqueeze(-1)).float()
        txt = model.text_bert(dev["text"], txt_mask, dtype)
        xo = ops.embed_rows(dev["video_feat"].contiguous(), None, None, None, None, None, dtype)            # L2norm(1024) only
        assert xo.shape == (B, 1, 1024)
        obj = FN.layer_norm(F.linear(xo, model.linear_obj_feat_to_mmt_in.weight.to(dtype), model.linear_obj_feat_to_mmt_in.bias.to(dtype)),
                            model.obj_feat_layer_norm.weight, model.obj_feat_layer_norm.bias)
        xc = ops.embed_rows(dev["context_feature_0"].contiguous(), dev["context_feature_1"].contiguous(), None, None, None, None, dtype)
        assert xc.shape == (B, N, 904)
        a = FN.layer_norm(F.linear(xc, model.linear_ocr_feat_to_mmt_in.weight.to(dtype), model.linear_ocr_feat_to_mmt_in.bias.to(dtype)),
                          model.ocr_feat_layer_norm.weight, model.ocr_feat_layer_norm.bias)
        b = FN.layer_norm(F.linear(dev["ocr_bbox_coordinates"], model.linear_ocr_bbox_to_mmt_in.weight, model.linear_ocr_bbox_to_mmt_in.bias),
                          model.ocr_bbox_layer_norm.weight, model.ocr_bbox_layer_norm.bias)
        ocr = a + b
        ocr_out, dec_out = model.mmt(txt, txt_mask, obj, dev["frame_mask"], ocr, dev["ocr_mask"], model.classifier.module.weight,
                                     dev["train_prev_inds"].clone(), dtype)
        scores = model._forward_output(ocr_out, dec_out, dev["ocr_mask"], dtype)
    loss = POSBCEWithMaskLoss()(SampleList({"targets": dev["targets"], "train_loss_mask": dev["train_loss_mask"]}), {"pos_scores": scores})
    loss.backward()

    # ---- oracle (fp64): the shared functions + the 1024 / 904 input stage restated
    s64 = {k: v.double().requires_grad_(v.is_floating_point() and not O.is_dead(k)) for k, v in sd.items()}
    d64 = {k: (v.double() if v.is_floating_point() else v) for k, v in batch.items()}
    tm = O.get_mask(d64["text_len"], 20).double()
    r_txt = O.text_bert(s64, d64["text"], tm)
    r_obj = O.layer_norm(O.linear(O.l2_normalize(d64["video_feat"]), s64["linear_obj_feat_to_mmt_in.weight"], s64["linear_obj_feat_to_mmt_in.bias"]),
                         s64["obj_feat_layer_norm.weight"], s64["obj_feat_layer_norm.bias"])
    rx = torch.cat([O.l2_normalize(d64["context_feature_0"]), O.l2_normalize(d64["context_feature_1"])], -1)
    r_ocr = (O.layer_norm(O.linear(rx, s64["linear_ocr_feat_to_mmt_in.weight"], s64["linear_ocr_feat_to_mmt_in.bias"]),
                          s64["ocr_feat_layer_norm.weight"], s64["ocr_feat_layer_norm.bias"])
             + O.layer_norm(O.linear(d64["ocr_bbox_coordinates"], s64["linear_ocr_bbox_to_mmt_in.weight"], s64["linear_ocr_bbox_to_mmt_in.bias"]),
                            s64["ocr_bbox_layer_norm.weight"], s64["ocr_bbox_layer_norm.bias"]))
    r_oo, r_do = O.mmt(s64, r_txt, tm, r_obj, d64["frame_mask"], r_ocr, d64["ocr_mask"], d64["train_prev_inds"].clone())
    r_scores = O.forward_output(s64, r_oo, r_do, d64["ocr_mask"])
    r_loss = O.pos_bce_loss(r_scores, d64["targets"], d64["train_loss_mask"])
    r_loss.backward()

    assert scores.shape == (B, 12, V + N)
    assert (scores.double().cpu() - r_scores.detach()).abs().max().item() < tol
    assert abs(loss.item() - r_loss.item()) < (1e-4 if dtype == torch.float32 else 5e-3) * abs(r_loss.item())
    named = dict(model.named_parameters())
    rel = 5e-3 if dtype == torch.float32 else 6e-2
    checked = 0
    for n in ("mmt.encoder.layer.0.attention.self.query.weight", "mmt.encoder.layer.2.output.dense.weight", "text_bert.encoder.layer.1.intermediate.dense.weight",
              "linear_obj_feat_to_mmt_in.weight", "linear_ocr_feat_to_mmt_in.weight", "ocr_ptr_net.key.weight", "classifier.module.weight",
              "mmt.prev_pred_embeddings.ocr_layer_norm.weight"):
        g, r = named[n].grad.double().cpu(), s64[n].grad
        assert (g - r).norm().item() <= rel * r.norm().item() + 1e-12, (n, (g - r).norm().item(), r.norm().item())
        checked += 1
    assert checked == 8
