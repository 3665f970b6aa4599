"""How the bf16 operand mode's logit error grows as attention gets peakier, at the metric's own length (round 6, VERDICT r5 #1: "nobody knows
whether trained weights pass").  The two reference-generated full-length fixtures sit at the ends - attn_gain 1 (reference init: entropy
8.5 nats, bf16 7e-3 - 8.6e-3) and attn_gain 6 (entropy 0.3 - 1.2 nats, bf16 6e-2 - 9.5e-2, the reference's own autocast run 6e-2 - 9.4e-2);
this test fills in the curve between them against the CPU ORACLE (pinned to the reference at both ends: tests/test_oracle_golden.py,
logits 9e-6): B = 1, 100 frames x 100 OCR tokens (L = 10 132), query / key weights scaled by 1, 2, 3, 4 (score sigma ~ 0.3 gain^2 nats).
Asserted: the fp32 parity mode stays below the north star's 1e-3 at EVERY gain (selection masks and noise injected from the oracle's run);
the bf16 mode meets 1e-2 at gain 1.  Printed (pytest -s; profiles/r06_bf16_vs_peakiness.txt): max / RMS bf16 logit error and the oracle's
attention entropy per gain - the error follows the score scale (gain^2), i.e. it is set by how far the weights are from the init
regime, not by the kernels (tools/error_budget.py)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GAINS = (1.0, 2.0, 3.0, 4.0)
F, P, V, SEED = 100, 100, 5000, 31


def test_bf16_logit_error_against_the_oracle_as_attention_gets_peakier():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle import t2s_oracle as O
    from vitxt_gqa_amd.init import make_state_dict
    from vitxt_gqa_amd.schema import state_dict_schema
    from vitxt_gqa_amd.synth import make_batch, make_noise
    from vitxt_gqa_amd.testing import make_model, to_device
    batch = make_batch(1, F, P, V=V, seed=SEED)
    e1, e2 = make_noise(1, F, P, seed=SEED)
    cfg = dict(frame_topk=5, ocr_topk=5, frame_num=F, ocr_frame_num=P)
    schema = state_dict_schema(V)
    rows = []
    assert O.ATTENTION_IMPL == "eager"
    for gain in GAINS:
        sd = make_state_dict(schema, seed=SEED, attn_gain=gain)
        O.ATTENTION_IMPL = "sdpa"          # the eager form needs ~5 GB per [12, L, L] score tensor; pinned to it by tests/test_oracle_golden.py
        try:
            with torch.no_grad():
                ref = O.t2s_forward(sd, batch, cfg, training=True, expo_frame=e1, expo_ocr=e2, keep=True)
        finally:
            O.ATTENTION_IMPL = "eager"
        it = ref["_inter"]
        masks = {k: it[k].float() for k in ("pos_obj_mask", "neg_obj_mask", "pos_ocr_mask", "neg_ocr_mask")}
        # the oracle's own attention entropy in the first MMT layer's regime: scores of the QTV output rows under this gain's weights
        x = torch.cat([it["txt_emb"], it["obj_in"], it["ocr_in"]], 1)[0]
        wq, wk = sd["mmt.encoder.layer.0.attention.self.query.weight"], sd["mmt.encoder.layer.0.attention.self.key.weight"]
        vis = torch.cat([torch.arange(20) < batch["text_len"][0], batch["frame_mask"][0].bool(), batch["ocr_mask"][0].bool()])
        q = (x[::97] @ wq.t()).view(-1, 12, 64).transpose(0, 1)
        k = (x[vis] @ wk.t()).view(-1, 12, 64).transpose(0, 1)
        pr = torch.softmax(q @ k.transpose(1, 2) / 8.0, -1)
        entropy = float(-(pr * torch.log(pr.clamp_min(1e-30))).sum(-1).mean())
        res = {}
        for name, dt in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
            model = make_model(F, P, V, dtype=dt, state_dict=sd).to(DEV).train()
            s = to_device(batch, DEV)
            s.grounding_noise = (e1, e2)
            s.grounding_masks = masks
            with torch.no_grad():
                out = model.forward(s)
            err = torch.cat([(out[k_].float().cpu() - ref[k_]).abs().flatten() for k_ in ("ref_scores", "pos_scores", "neg_scores")])
            res[name] = (err.max().item(), err.pow(2).mean().sqrt().item())
            del model, out
            torch.cuda.empty_cache()
        rows.append((gain, entropy, res))
        print("attn_gain %.0f: oracle attention entropy (mmt layer 0 regime) %.2f nats of %.2f | fp32 mode max %.2e | bf16 mode max %.3e RMS %.3e" % (
            gain, entropy, math.log(int(vis.sum())), res["fp32"][0], res["bf16"][0], res["bf16"][1]), flush=True)
        assert res["fp32"][0] < 1e-3, (gain, res["fp32"])
    assert rows[0][2]["bf16"][0] < 1e-2, rows[0]
    # the error follows the score scale: it does not shrink as the weights leave the init regime
    assert rows[-1][2]["bf16"][1] > rows[0][2]["bf16"][1]
