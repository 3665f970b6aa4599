"""GPU parity of the whole T2S path (model registry surface -> HIP kernels) against the committed golden
fixtures of the reference and against the CPU oracle.  Tolerances are the north-star's: logits within
1e-3 (fp32 mode) / 1e-2 (bf16 mode, reference-std weights), pointer/copy indices bit-exact."""
import pytest
import torch

from golden_util import Fixture

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


ALL_CASES = ["tiny_b2_f6_p8", "cfg1_b2_f20_p30", "ptr_b3_f8_p10"]


def _run(fx, dtype, train=True, inject=True):
    from vitxt_gqa_amd.testing import build_model_for_fixture, to_device
    model = build_model_for_fixture(fx, dtype).to(DEV)
    model.train(train)
    model.keep_intermediates = True
    s = to_device(fx.batch(), DEV)
    s.grounding_noise = (fx["E1"], fx["E2"])
    if inject:
        s.grounding_masks = fx.masks("" if train else "eval_")
    return model, s


@pytest.mark.parametrize("case", ALL_CASES)
def test_forward_fp32_matches_reference(case):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    fx = Fixture(case)
    model, s = _run(fx, torch.float32)
    out = model(s)
    for k in ("ref_scores", "pos_scores", "neg_scores"):
        err = (out[k].float().cpu() - fx[k]).abs().max().item()
        assert err < 1e-3, "%s max abs err %.3e" % (k, err)
    st = fx.meta["row_stride"]
    f = model._last_fwd
    assert (f["txt_emb"].float().cpu() - fx["txt_emb"]).abs().max().item() < 1e-3
    assert (f["ocr_mmt_in"].float().cpu()[:, ::st] - fx["ocr_in"]).abs().max().item() < 1e-3
    assert (f["frame_score"].cpu() - fx["frame_score"]).abs().max().item() < 1e-4
    assert torch.equal(out["frame_topk"].cpu(), fx["frame_topk"]) and torch.equal(out["ocr_topk"].cpu(), fx["ocr_topk"])
    # every intermediate the reference fixture stores (tests/golden/make_golden.py): the encodings before QTV, the pooled
    # question, the decoder-step embeddings (the ptr case feeds OCR copies: prev_inds >= V, OCR-row gather + type embedding 1,
    # t2s.py:690-723), the MMT outputs of the three passes, and - masks injected - the outputs derived from them, EQUAL
    cpu = lambda t: t.float().cpu()
    assert (cpu(f["txt_emb0"]) - fx["txt_emb0"]).abs().max().item() < 1e-3
    assert (cpu(f["obj_in0"]) - fx["obj_in0"]).abs().max().item() < 1e-3
    assert (cpu(f["ocr_in0"])[:, ::st] - fx["ocr_in0"]).abs().max().item() < 1e-3
    assert (cpu(f["obj_mmt_in"]) - fx["obj_in"]).abs().max().item() < 1e-3
    assert (cpu(f["global_q"]) - fx["global_q"].view(fx.B, -1)).abs().max().item() < 1e-3
    assert (cpu(f["ref_dec_emb"]) - fx["dec_emb"]).abs().max().item() < 1e-4
    assert (cpu(f["ref_mmt_ocr"])[:, ::st] - fx["ref_mmt_ocr"]).abs().max().item() < 1e-3
    for p in ("ref", "pos", "neg"):
        assert (cpu(f[p + "_mmt_dec"]) - fx[p + "_mmt_dec"]).abs().max().item() < 1e-3, p
    assert torch.equal(out["ground_frame"].cpu(), fx["ground_frame"])
    assert torch.equal(out["ground_box"].cpu(), fx["ground_box"])
    # the spatial scorer runs on the build's OWN grounded frames: comparable where those are the reference's (tie-free samples)
    same = (f["new_ocr_mask"].cpu().float() == fx["new_ocr_mask"].float()).all(-1)
    if same.any():
        assert (f["ocr_score"].cpu()[same] - fx["ocr_score"][same]).abs().max().item() < 1e-4
    # ... and on the REFERENCE's grounded frames for every sample: the scorer kernel on the fixture's new_ocr_mask
    from vitxt_gqa_amd import ops
    sc = ops.attention_score(f["global_q"], f["ocr_mmt_in"].float().contiguous(), fx["new_ocr_mask"].float().to(DEV).contiguous())
    assert (sc.cpu() - fx["ocr_score"]).abs().max().item() < 1e-4
    if case == "ptr_b3_f8_p10":
        assert (s.train_prev_inds >= fx.V).float().mean().item() > 0.3
    # losses through BaseModel.__call__ (base_model.py:119-149) with the yml weights
    losses = out["losses"]
    assert set(losses) == {"train/vtextgqa/pos_bce_loss", "train/vtextgqa/InfoNCE"}
    assert abs(losses["train/vtextgqa/pos_bce_loss"].item() - fx["loss_bce"].item()) < 1e-3 * fx["loss_bce"].item()
    assert abs(losses["train/vtextgqa/InfoNCE"].item() / 1000 - fx["loss_nce"].item()) < 2e-4


# bf16 logit tolerance.  The north star's 1e-2 holds at reference-std weights: asserted as such on cfg1 (every logit) and on the
# VOCABULARY logits of the ptr fixture.  The ptr fixture scales both pointer projections by 2.5 (so that the two heads compete:
# |logit| up to 6.9; cfg1: up to 9.6), which multiplies a reference-std error of its POINTER logits by 6.25: their effective
# tolerance is 6.25e-2 (stated here and in DESIGN section 2; the measured maxima are printed by the test).
BF16_TOL = {"cfg1_b2_f20_p30": 1e-2, "ptr_b3_f8_p10": 1e-2}


def _bf16_tol(fx, tol):
    """Per-logit tolerance [V + N]: ``tol`` (the north star's 1e-2 at reference-std weights) times the gain the fixture applies to
    the head that produces the logit: the pointer logits of the ptr fixture are q.k with BOTH projections scaled by 2.5, so a
    reference-std error e shows up as 6.25 e there; its vocabulary head is scaled by 0.3 (kept at ``tol``)."""
    g = fx.meta.get("gains") or {}
    t = torch.full((fx.V + fx.F * fx.P,), tol)
    t[fx.V:] *= max(1.0, g.get("ocr_ptr_net.", 1.0) ** 2)
    return t


@pytest.mark.parametrize("case", ["cfg1_b2_f20_p30", "ptr_b3_f8_p10"])
def test_forward_bf16(case):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    fx = Fixture(case)
    model, s = _run(fx, torch.bfloat16)
    out = model(s)
    tolv = _bf16_tol(fx, BF16_TOL[case])
    total = 0
    for k in ("ref_scores", "pos_scores", "neg_scores"):
        err = (out[k].float().cpu() - fx[k]).abs()
        ev, ep = err[..., :fx.V].max().item(), err[..., fx.V:].max().item()
        print("%s %s: max abs logit error - vocabulary %.3e (tolerance %.3g), pointer %.3e (tolerance %.3g)" % (
            case, k, ev, tolv[0].item(), ep, tolv[-1].item()))
        assert (err < tolv).all(), "%s max abs err: vocabulary logits %.3e, pointer logits %.3e" % (k, ev, ep)
        # pointer / copy indices (north star: bit-exact): the argmax of every decoding row equals the reference's wherever the
        # reference's own gap between the two competing logits exceeds what those two logits' tolerances can close (checked per
        # flipped ROW, with the tolerance of the head each of the two logits comes from); the NUMBER of rows that differ is
        # bounded by the number of such near-tie rows of the reference (a deviation from bit-exact stated as a count)
        flips = _index_flips(out[k].float().cpu(), fx[k], tolv)
        near = _near_tie_rows(fx[k], tolv)
        assert flips <= near
        total += flips
        print("%s %s: %d of %d argmax indices differ (reference rows whose winner is within tolerance of another logit: %d)" % (
            case, k, flips, fx[k].shape[0] * fx[k].shape[1], near))
    if case == "ptr_b3_f8_p10":      # teacher-forced rows of this fixture: the winners are OCR tokens AND vocabulary tokens
        am = fx["pos_scores"].argmax(-1)
        assert (am >= fx.V).any() and (am < fx.V).any()


def _near_tie_rows(want, tolv):
    """Rows of the reference in which some other logit j lies within tol[winner] + tol[j] of the winner."""
    top = want.max(-1, keepdim=True)
    wi = want.argmax(-1, keepdim=True)
    close = (top.values - want) < (tolv[wi.squeeze(-1)].unsqueeze(-1) + tolv)
    close.scatter_(-1, wi, False)
    return int(close.any(-1).sum())


def _index_flips(got, want, tolv):
    """Number of rows whose argmax differs; asserts that EACH of them is itself a near-tie of the reference between exactly the two
    indices involved: want[winner] - want[picked] < tol[winner] + tol[picked] (per-logit tolerances: 1e-2 for a vocabulary logit,
    the gain-scaled value for a pointer logit of the ptr fixture)."""
    if not torch.is_tensor(tolv):
        tolv = torch.full((want.shape[-1],), float(tolv) / 2)
    gi, wi = got.argmax(-1), want.argmax(-1)
    bad = gi != wi
    if bad.any():
        gap = want.gather(-1, wi.unsqueeze(-1)).squeeze(-1) - want.gather(-1, gi.unsqueeze(-1)).squeeze(-1)
        allow = tolv[wi] + tolv[gi]
        assert (gap[bad] < allow[bad]).all(), "argmax differs on rows whose reference gap between the two indices is %s (allowed %s)" % (
            gap[bad].tolist(), allow[bad].tolist())
    return int(bad.sum())


def test_own_selection_matches_reference_where_tie_free():
    """Without mask injection (noise injected only): frame scores, ground_frame on tie-free rows, pointer argmax."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    fx = Fixture("cfg1_b2_f20_p30")
    model, s = _run(fx, torch.float32, inject=False)
    out = model(s)
    f = model._last_fwd
    E1 = fx["E1"]
    pos_cnt = ((-torch.log(E1[:, 0])) >= (-torch.log(E1[:, 1]))).sum(1)
    checked = 0
    for b in range(fx.B):
        if pos_cnt[b] >= 5:
            assert torch.equal(out["ground_frame"][b].cpu(), fx["ground_frame"][b])
            assert torch.equal(f["pos_obj_mask"][b].cpu(), fx["pos_obj_mask"][b])
            assert torch.equal(f["new_ocr_mask"][b].cpu(), fx["new_ocr_mask"][b])
            checked += 1
    assert checked > 0
    assert f["pos_ocr_mask"].sum(1).tolist() == [5.0 * fx.F] * fx.B
    assert out["ground_box"].shape == (fx.B, 5 * fx.F, 4)


@pytest.mark.parametrize("case", ["tiny_b2_f6_p8", "ptr_b3_f8_p10"])
def test_gradients_fp32_match_reference(case):
    """ptr case: half of the decoder-step embeddings are OCR copies, so the gradient reaches the OCR rows through the gather of
    PrevPredEmbeddings (t2s.py:702-709) as well as through the encoder."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    fx = Fixture(case)
    model, s = _run(fx, torch.float32)
    out = model(s)
    loss = sum(v.mean() for v in out["losses"].values())            # base_trainer.py:274-278
    assert abs(loss.item() - fx["loss_total"].item()) < 1e-3 * fx["loss_total"].item()
    loss.backward()
    params = dict(model.named_parameters())
    names = fx.meta["grad_names"]
    ref = fx["grad_norms"]
    total = fx["grad_total_norm"].item()
    live = {n for n, p in params.items() if p.grad is not None}
    assert live == set(names), (sorted(live ^ set(names))[:6])
    for n, r in zip(names, ref.tolist()):
        g = params[n].grad.double().norm().item()
        assert abs(g - r) <= 5e-3 * r + 1e-6 * total, "%s grad norm %g vs %g" % (n, g, r)
    for k, v in fx.arr.items():
        if k.startswith("grad:"):
            n = k[5:]
            g = params[n[:-1].split("[:")[0]].grad[:int(n[:-1].split("[:")[1])] if n.endswith("]") else params[n].grad
            sc = v.abs().max().item()
            assert (g.float().cpu() - v).abs().max().item() < 5e-3 * sc + 1e-7 * total, k


@pytest.mark.parametrize("fused_everywhere", [True, False])
def test_gradients_bf16_cfg1_match_reference(fused_everywhere, monkeypatch):
    """Model-level bf16 gradient parity against the REFERENCE's fixture (cfg1: batch 2, 20 frames x 30 OCR, reference-std
    weights): every parameter-gradient norm within 3 %, the total norm within 1 %.  ``fused_everywhere``: the five-product
    fused attention backward (the benchmark's dominant kernel) takes EVERY launch (ops.ATTN_BWD_FUSED_MIN_KEYS = 0; by default
    only launches with >= 2048 keys use it, and the fixtures have L <= 652)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from vitxt_gqa_amd import ops
    if fused_everywhere:
        import vitxt_gqa_amd.functional as FNmod
        monkeypatch.setattr(ops, "ATTN_BWD_FUSED_MIN_KEYS", 0)
        monkeypatch.setattr(ops, "ATTN_BWD_FUSED", True)
        monkeypatch.setattr(FNmod, "PRUNE_KV_MAX_KEYS", 0)      # (the pruned pos / neg launches take the two-kernel form)
    fx = Fixture("cfg1_b2_f20_p30")
    model, s = _run(fx, torch.bfloat16)
    out = model(s)
    loss = sum(v.mean() for v in out["losses"].values())
    assert abs(loss.item() - fx["loss_total"].item()) < 5e-3 * fx["loss_total"].item()
    seen = []
    orig = ops.attn_bwd

    def spy(*a, **k):
        r = orig(*a, **k)
        seen.append(ops.LAST_ATTN_BWD_PRODUCTS)
        return r

    monkeypatch.setattr(ops, "attn_bwd", spy)
    loss.backward()
    if fused_everywhere:
        assert seen and all(p == 5 for p in seen[3:]), seen        # the 3 TextBert layers (L = 20) included or not: all MMT / QTV launches fused
    params = dict(model.named_parameters())
    names, ref, total = fx.meta["grad_names"], fx["grad_norms"], fx["grad_total_norm"].item()
    assert {n for n, p in params.items() if p.grad is not None} == set(names)
    worst, bad = (0.0, ""), []
    sq = 0.0
    for n, r in zip(names, ref.tolist()):
        g = params[n].grad.double().norm().item()
        sq += g * g
        if n.endswith("attention.self.key.bias"):
            # mathematically zero (softmax shift invariance); the reference's fp32 value is rounding noise (1e-6 .. 1e-5), and
            # so is the bf16 one: bounded absolutely, at 1e-5 of the total gradient norm
            if g > 1e-5 * total:
                bad.append((n, g, r))
            continue
        rel = abs(g - r) / (r + 1e-6 * total)
        worst = max(worst, (rel, n))
        if rel >= 3e-2:
            bad.append((n, g, r))
    assert not bad, "gradient norms outside 3 %% of the reference: %s" % bad
    assert abs(sq ** 0.5 - total) < 1e-2 * total, (sq ** 0.5, total)
    for k, v in fx.arr.items():            # the stored gradient tensors themselves
        if k.startswith("grad:"):
            n = k[5:]
            g = params[n[:-1].split("[:")[0]].grad[:int(n[:-1].split("[:")[1])] if n.endswith("]") else params[n].grad
            d = (g.float().cpu() - v).norm().item()
            assert d < 3e-2 * v.norm().item() + 1e-5 * total, (k, d, v.norm().item())
    print("bf16 gradient norms vs reference (fused everywhere: %s): worst %.3f %% (%s), total %.4f %%" % (
        fused_everywhere, 100 * worst[0], worst[1], 100 * abs(sq ** 0.5 - total) / total))


@pytest.mark.parametrize("case", ["tiny_b2_f6_p8", "ptr_b3_f8_p10"])
def test_eval_greedy_decode_indices_exact(case):
    """fp32 mode: the greedy decode's indices are the reference's, bit-exact.  In the ptr case the reference's own sequence
    walks through OCR tokens (83 % of the steps) and vocabulary tokens, 4 of its 36 rows have a top-2 gap below 0.05, and every
    index it picks is fed back through the OCR-copy branch of the decoder-step embeddings (t2s.py:315-354, 690-723)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    fx = Fixture(case)
    model, s = _run(fx, torch.float32, train=False)
    with torch.no_grad():
        out = model(s)
    for k in ("ref", "pos", "neg"):
        assert (out[k + "_scores"].cpu() - fx["eval_%s_scores" % k]).abs().max().item() < 1e-3, k
    assert torch.equal(out["pos_scores"].argmax(-1).cpu(), fx["eval_argmax"])
    f = model._last_fwd
    assert torch.equal(f["prev_inds"][:, 1:].cpu(), fx["eval_argmax"][:, :-1])
    assert (f["dec_emb_last"].float().cpu() - fx["eval_dec_emb_last"]).abs().max().item() < 1e-4
    assert torch.equal(out["ground_frame"].cpu(), fx["eval_ground_frame"])
    assert torch.equal(out["ground_box"].cpu(), fx["eval_ground_box"])


@pytest.mark.parametrize("case,dtype,tol", [("tiny_b2_f6_p8", torch.float32, 1e-3), ("cfg1_b2_f20_p30", torch.bfloat16, 1e-2),
                                            ("ptr_b3_f8_p10", torch.float32, 1e-3), ("ptr_b3_f8_p10", torch.bfloat16, 1e-2)])
def test_cached_decode_equals_reference_loop(case, dtype, tol):
    """Prefix-reuse greedy decoding == the reference's recompute-everything loop (same scores, same indices)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    fx = Fixture(case)
    model, s = _run(fx, dtype, train=False)
    with torch.no_grad():
        model.decode_with_prefix_cache = True
        a = model(s)
        pa = model._last_fwd["prev_inds"].clone()
        model.decode_with_prefix_cache = False
        b = model(s)
        pb = model._last_fwd["prev_inds"].clone()
    if dtype == torch.float32:
        for k in ("ref_scores", "pos_scores", "neg_scores"):
            assert (a[k] - b[k]).abs().max().item() < 1e-4, k
            assert (a[k].cpu() - fx["eval_" + k]).abs().max().item() < tol, k
        assert torch.equal(pa, pb)
        assert torch.equal(a["pos_scores"].argmax(-1).cpu(), fx["eval_argmax"])
    else:
        # Row t of the final scores was computed from the fed tokens 0..t (causal decoder): rows are comparable between two runs
        # up to each sample's first differing fed token; later rows were fed another token.
        def comparable(fed_a, fed_b):
            return (fed_a.cpu() == fed_b.cpu()).long().cumprod(1).bool()

        fed_ref = torch.cat([torch.full_like(fx["eval_argmax"][:, :1], 1), fx["eval_argmax"][:, :-1]], 1)      # BOS, then the reference's picks
        tolv = _bf16_tol(fx, tol)
        tol = tolv.max().item()
        for k in ("ref_scores", "pos_scores", "neg_scores"):
            d = ((a[k] - b[k]).abs().cpu() / tolv).amax(-1)
            assert d[comparable(pa, pb)].max().item() < 2, ("cached vs loop", k, d)
            d = ((a[k].cpu() - fx["eval_" + k]).abs() / tolv).amax(-1)
            assert d[comparable(pa, fed_ref)].max().item() < 1, ("cached vs reference", k, d)
        # bf16 operands (the throughput dtype): greedy-decode indices against the reference's, step by step.  Row t of the
        # final scores is the logit row that decided step t (causal decoder), so up to a sample's FIRST differing step the
        # rows are comparable: every earlier index must be equal, and the differing one must be a near-tie of the reference
        # (top-2 logit gap below what two logits within the 1e-2 bf16 tolerance can close).  Later steps of that sample were
        # fed a different token and are not comparable.
        got, want = a["pos_scores"].argmax(-1).cpu(), fx["eval_argmax"]
        ref_sc = fx["eval_pos_scores"]
        flipped = 0
        for b_ in range(got.shape[0]):
            diff = (got[b_] != want[b_]).nonzero().flatten()
            if diff.numel():
                t = int(diff[0])
                gi, wi = int(got[b_, t]), int(want[b_, t])
                gap = (ref_sc[b_, t, wi] - ref_sc[b_, t, gi]).item()
                assert gap < (tolv[wi] + tolv[gi]).item(), "sample %d step %d: index %d vs %d at reference gap %.3e (allowed %.3e)" % (
                    b_, t, gi, wi, gap, (tolv[wi] + tolv[gi]).item())
                flipped += 1
        # the deviation from "bit-exact" as a number: samples that leave the reference's sequence <= samples whose reference
        # sequence contains a near-tie row at all
        near = sum(1 for b_ in range(got.shape[0]) if _near_tie_rows(ref_sc[b_], tolv) > 0)
        assert flipped <= near
        print("bf16 greedy decode (%s): %d of %d samples leave the reference's index sequence, each at a near-tie (samples with a near-tie row: %d)"
              % (case, flipped, got.shape[0], near))


@pytest.mark.parametrize("B,F,P,V", [(1, 5, 5, 11), (3, 7, 9, 40), (2, 33, 6, 300), (2, 12, 1, 30), (2, 9, 3, 30)])
def test_ragged_inputs_match_oracle(B, F, P, V):
    """Edge cases the dataset produces (Appendix B): text_len 1 and 20, padded frames (frame_mask 0, frame_id 0),
    a sample whose OCR slots are all padding, minimum (F, P) = (frame_topk, ocr_topk), P < ocr_topk (the reference's
    slice then keeps all P slots of a frame: the "100 OCR tokens in total" reading of BASELINE's shape is F=100, P=1),
    odd sizes that do not divide any tile.  Whole model (own selection, noise injected) vs the CPU oracle with the same tie rule, fp32 mode."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle import t2s_oracle as O
    from vitxt_gqa_amd.init import make_state_dict
    from vitxt_gqa_amd.schema import state_dict_schema
    from vitxt_gqa_amd.synth import make_batch, make_noise
    from vitxt_gqa_amd.testing import make_model, to_device
    sd = make_state_dict(state_dict_schema(V, text_vocab=50), seed=B, attn_gain=4.0)
    batch = make_batch(B, F, P, V=V, seed=F, text_vocab=50)
    batch["text_len"][0] = 1
    batch["text_len"][-1] = 20
    if F > 5:                                   # padded trailing frames: mask 0, id 0 (and their OCR slots padded)
        batch["frame_mask"][0, F - 1:] = 0
        batch["frame_id"][0, F - 1:] = 0
        batch["temporal_id"][0, (F - 1) * P:] = 0
        batch["ocr_mask"][0, (F - 1) * P:] = 0
    batch["ocr_mask"][-1] = 0                   # a question whose video has no OCR at all
    e1, e2 = make_noise(B, F, P, seed=7)
    model = make_model(F, P, V, text_vocab=50, dtype=torch.float32, state_dict=sd).to(DEV).train()
    s = to_device(batch, DEV)
    s.grounding_noise = (e1, e2)
    out = model(s)
    cfg = dict(frame_topk=5, ocr_topk=5, frame_num=F, ocr_frame_num=P)
    ref = O.t2s_forward({k: v.double() for k, v in sd.items()},
                        {k: (v.double() if v.is_floating_point() else v) for k, v in batch.items()}, cfg,
                        training=True, expo_frame=e1.double(), expo_ocr=e2.double(), keep=True)
    f = model._last_fwd
    it = ref["_inter"]
    assert (f["frame_score"].double().cpu() - it["frame_score"]).abs().max().item() < 1e-4
    # selection masks are integer work: the frame stage must be EQUAL (asserted to be decisive first), the OCR masks EQUAL on
    # every (sample, frame) row whose decisions are separated by more than the fp32-vs-fp64 score difference (selection_util)
    from selection_util import decisive_frames, decisive_ocr_rows, relative_diff
    tol = max(1e-4, 4 * relative_diff(f["ocr_score"].cpu(), it["ocr_score"]))
    fm = batch["frame_mask"].double()
    ok_f = decisive_frames(it["frame_score"], fm, e1, 5, tol=max(1e-4, 4 * relative_diff(f["frame_score"].cpu(), it["frame_score"])))
    assert ok_f.any(), "every sample's frame selection hinges on rounding; pick another seed"
    assert torch.equal(f["pos_obj_mask"].cpu().double()[ok_f], it["pos_obj_mask"][ok_f])
    assert torch.equal(f["neg_obj_mask"].cpu().double()[ok_f], it["neg_obj_mask"][ok_f])
    assert torch.equal(out["ground_frame"].cpu()[ok_f], ref["ground_frame"][ok_f])
    assert torch.equal(f["new_ocr_mask"].cpu().double()[ok_f], it["new_ocr_mask"][ok_f])
    ok = decisive_ocr_rows(it["ocr_score"], it["new_ocr_mask"], e2, 5, F, P, tol=tol) & ok_f.unsqueeze(-1)
    assert ok.float().mean().item() > 0.4
    okn = ok.unsqueeze(-1).expand(B, F, P).reshape(B, F * P)
    assert torch.equal(f["pos_ocr_mask"].cpu().double()[okn], it["pos_ocr_mask"][okn])
    if not torch.equal(f["neg_ocr_mask"].cpu().double()[okn], it["neg_ocr_mask"][okn]):          # say which rows, with their inputs
        bad = ((f["neg_ocr_mask"].cpu().double() != it["neg_ocr_mask"]) & okn).view(B, F, P).any(-1).nonzero()
        b_, f_ = [int(v) for v in bad[0]]
        sl = slice(f_ * P, (f_ + 1) * P)
        raise AssertionError("neg_ocr_mask differs on decisive row (b=%d, f=%d): gpu %s oracle %s | oracle score %s gpu score %s | new_mask %s | g0-g1 %s" % (
            b_, f_, f["neg_ocr_mask"][b_, sl].tolist(), it["neg_ocr_mask"][b_, sl].tolist(), it["ocr_score"][b_, sl].tolist(),
            f["ocr_score"][b_, sl].tolist(), it["new_ocr_mask"][b_, sl].tolist(), (-torch.log(e2[b_, 0, sl]) + torch.log(e2[b_, 1, sl])).tolist()))
    agree = (f["pos_ocr_mask"].cpu().double() == it["pos_ocr_mask"]).double().mean().item()
    same_neg = (torch.equal(f["neg_ocr_mask"].cpu().double(), it["neg_ocr_mask"])
                and torch.equal(f["neg_obj_mask"].cpu().double(), it["neg_obj_mask"]))
    if agree == 1.0 and same_neg:
        for k in ("ref_scores", "pos_scores", "neg_scores"):
            err = (out[k].double().cpu() - ref[k]).abs().max().item()
            assert err < 1e-3, "%s max abs err %.3e" % (k, err)
        loss, _, _ = O.total_loss(ref, batch["targets"].double(), batch["train_loss_mask"].double())
        got = sum(v.mean() for v in out["losses"].values()).item()
        assert abs(got - loss.item()) < 2e-3 * abs(loss.item())
    else:                                       # masks differ somewhere: the ref pass does not depend on them
        assert (out["ref_scores"].double().cpu() - ref["ref_scores"]).abs().max().item() < 1e-3


def test_no_cpu_fallback():
    """The product path refuses CPU tensors instead of silently computing elsewhere."""
    from vitxt_gqa_amd import ops
    with pytest.raises(RuntimeError):
        ops.gelu_fwd(torch.zeros(8, 8))


def test_batched_mmt_passes_equal_separate_passes():
    """MMT.forward_passes (the three passes stacked along the batch, one encoder call) against the reference's three
    separate calls: same scores and same parameter gradients (fp32 mode, dropout 0)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from vitxt_gqa_amd.synth import make_batch, make_noise
    from vitxt_gqa_amd.testing import make_model, to_device
    F, P, V, B = 7, 9, 40, 3
    model = make_model(F, P, V, text_vocab=50, dtype=torch.float32, attn_gain=4.0).to(DEV).train()
    s = to_device(make_batch(B, F, P, V=V, seed=3, text_vocab=50), DEV)
    s.grounding_noise = tuple(t.to(DEV) for t in make_noise(B, F, P, seed=3))
    res = {}
    model.share_mmt_prefix = False          # "separate" = the reference's literal three encoder calls
    for mode in (False, True):
        model.batch_mmt_passes = mode
        model.zero_grad(set_to_none=True)
        out = model(s)
        sum(l.mean() for l in out["losses"].values()).backward()
        res[mode] = ({k: out[k].detach().clone() for k in ("ref_scores", "pos_scores", "neg_scores")},
                     {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None})
    model.batch_mmt_passes = False
    for k in res[False][0]:
        assert (res[False][0][k] - res[True][0][k]).abs().max().item() < 1e-4, k
    assert res[False][1].keys() == res[True][1].keys()
    for n, g in res[False][1].items():
        d = (g - res[True][1][n]).abs().max().item()
        assert d <= 1e-4 * max(1.0, g.abs().max().item()), (n, d)


@pytest.mark.parametrize("dtype,tol_s,tol_g", [(torch.float32, 1e-4, 1e-4), (torch.bfloat16, 3e-2, 3e-2)])
def test_shared_prefix_passes_equal_separate_passes(dtype, tol_s, tol_g):
    """MMT.forward_shared_prefix (one sequence [prefix | dec(ref) | dec(pos) | dec(neg)]: one concatenation, one operand copy,
    ONE layer-0 QKV projection for the three passes, their input / projection gradients summed in place) against the reference's
    three separate encoder calls (t2s.py:293-313): same scores, same parameter gradients (dropout 0).  fp32 mode: equal up to
    summation order; bf16 mode: the shared form sums the three passes' dQKV / residual gradients in bf16 before the one
    weight-gradient GEMM - compared at gradient-norm level."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from vitxt_gqa_amd.synth import make_batch, make_noise
    from vitxt_gqa_amd.testing import make_model, to_device
    F, P, V, B = 7, 9, 40, 3
    model = make_model(F, P, V, text_vocab=50, dtype=dtype, attn_gain=4.0).to(DEV).train()
    batch = make_batch(B, F, P, V=V, seed=3, text_vocab=50)
    batch["train_prev_inds"][:, 2] = V + 5           # copied OCR tokens: the OCR branch of the decoder-step embeddings
    batch["train_prev_inds"][:, 4] = V + F * P - 1
    batch["ocr_mask"][0, -P:] = 0
    s = to_device(batch, DEV)
    s.grounding_noise = tuple(t.to(DEV) for t in make_noise(B, F, P, seed=3))
    res = {}
    for mode in (False, True):
        model.share_mmt_prefix = mode
        model.zero_grad(set_to_none=True)
        out = model(s)
        sum(l.mean() for l in out["losses"].values()).backward()
        res[mode] = ({k: out[k].detach().float().clone() for k in ("ref_scores", "pos_scores", "neg_scores")},
                     {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None})
    for k in res[False][0]:
        assert (res[False][0][k] - res[True][0][k]).abs().max().item() < tol_s, k
    assert res[False][1].keys() == res[True][1].keys()
    tot = sum(g.double().norm().item() ** 2 for g in res[False][1].values()) ** 0.5
    for n, g in res[False][1].items():
        if dtype == torch.float32:
            d = (g - res[True][1][n]).abs().max().item()
            assert d <= tol_g * max(1.0, g.abs().max().item()), (n, d)
        else:
            d = (g.double() - res[True][1][n].double()).norm().item()
            assert d <= tol_g * g.double().norm().item() + 1e-3 * tot, (n, d, g.double().norm().item())


@pytest.mark.parametrize("share_prefix", [True, False])
def test_pruned_kv_projection_equals_the_full_projection(share_prefix, monkeypatch):
    """The pos / neg MMT passes project K and V only for the rows their top-k masks make keys (functional.PRUNE_KV_MAX_KEYS; masked
    keys contribute exactly nothing, t2s.py:609-618 with exp(-10000) == 0): same scores and same parameter gradients as with the
    full fused QKV projection, on the cfg1 fixture (bf16 operands; the two forms run different library GEMM shapes, so equal up to
    bf16 rounding of the operands' products, not bit for bit), and both within the reference's tolerance."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import vitxt_gqa_amd.functional as FNmod
    fx = Fixture("cfg1_b2_f20_p30")
    res = {}
    for prune in (0, 1024):
        monkeypatch.setattr(FNmod, "PRUNE_KV_MAX_KEYS", prune)
        model, s = _run(fx, torch.bfloat16)
        model.share_mmt_prefix = share_prefix
        seen = []
        orig = FNmod.ops.attn_fwd
        monkeypatch.setattr(FNmod.ops, "attn_fwd", lambda *a, **k: (seen.append(k.get("kv") is not None), orig(*a, **k))[1])
        out = model(s)
        monkeypatch.setattr(FNmod.ops, "attn_fwd", orig)
        # shared prefix: layers 1, 2 of the pos and neg passes; separate calls: all three layers of both
        assert sum(seen) == (0 if prune == 0 else 6), seen          # 3 layers x (pos, neg): layer 0 reads its K | V rows gathered from the shared projection
        sum(v.mean() for v in out["losses"].values()).backward()
        res[prune] = ({k: out[k].detach().float().cpu() for k in ("ref_scores", "pos_scores", "neg_scores")},
                      {n: p.grad.detach().double().cpu() for n, p in model.named_parameters() if p.grad is not None})
    for k in res[0][0]:
        assert (res[0][0][k] - res[1024][0][k]).abs().max().item() < 1e-2, k
        assert (res[1024][0][k] - fx[k]).abs().max().item() < 1e-2, k
    assert res[0][1].keys() == res[1024][1].keys()
    tot = sum(g.norm().item() ** 2 for g in res[0][1].values()) ** 0.5
    for n, g in res[0][1].items():          # two bf16 runs through different GEMM shapes: equal at bf16-noise level ...
        d = (g - res[1024][1][n]).norm().item()
        assert d <= 4e-2 * g.norm().item() + 1e-5 * tot, (n, d, g.norm().item())
    for k, v in fx.arr.items():             # ... and the pruned form is as close to the REFERENCE's gradient tensors as the full form
        if k.startswith("grad:"):
            n = k[5:]
            base, rows = (n[:-1].split("[:")[0], int(n[:-1].split("[:")[1])) if n.endswith("]") else (n, None)
            e = [(res[pr][1][base][:rows].float() - v).norm().item() if rows else (res[pr][1][base].float() - v).norm().item() for pr in (0, 1024)]
            assert e[1] <= 1.5 * e[0] + 1e-6 * tot and e[1] <= 3e-2 * v.norm().item() + 1e-5 * tot, (k, e, v.norm().item())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_recompute_form_gives_the_same_gradients(dtype):
    """functional.RECOMPUTE_ACTIVATIONS (the GELU output and the LN1 operand copy rebuilt in backward instead of stored - what
    bench.py's memory guard switches to on a card without the room) runs the same kernels on the same inputs: same scores, same
    gradients (scores bit for bit; these shapes take the atomics-free attention backward)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import vitxt_gqa_amd.functional as FNmod
    from vitxt_gqa_amd.synth import make_batch, make_noise
    from vitxt_gqa_amd.testing import make_model, to_device
    F, P, V, B = 7, 9, 40, 3
    model = make_model(F, P, V, text_vocab=50, dtype=dtype, attn_gain=4.0).to(DEV).train()
    s = to_device(make_batch(B, F, P, V=V, seed=4, text_vocab=50), DEV)
    s.grounding_noise = tuple(t.to(DEV) for t in make_noise(B, F, P, seed=4))
    res = {}
    try:
        for mode in (False, True):
            FNmod.RECOMPUTE_ACTIVATIONS = mode
            model.zero_grad(set_to_none=True)
            out = model(s)
            sum(l.mean() for l in out["losses"].values()).backward()
            res[mode] = ({k: out[k].detach().clone() for k in ("ref_scores", "pos_scores", "neg_scores")},
                         {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None})
    finally:
        FNmod.RECOMPUTE_ACTIVATIONS = False
    for k in res[False][0]:
        assert torch.equal(res[False][0][k], res[True][0][k]), k
    assert res[False][1].keys() == res[True][1].keys()
    for n, g in res[False][1].items():      # (embedding-table gradients are scatter-adds with float atomics: equal up to their order)
        d = (g.double() - res[True][1][n].double()).abs().max().item()
        assert d <= 1e-5 * max(1.0, g.abs().max().item()), (n, d)


def test_model_call_appends_metrics_from_the_config_list():
    """BaseModel.__call__ (base_model.py:119-149) appends ``metrics`` computed from the model's own outputs by the evaluators the
    yml lists: here textvqa_accuracy / stvqa_anls on the tiny fixture, against a direct evaluation of the same decoded answers."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from vitxt_gqa_amd import registry
    from vitxt_gqa_amd.metrics import STVQAANLSEvaluator, TextVQAAccuracyEvaluator, decode_answers
    fx = Fixture("tiny_b2_f6_p8")
    model, s = _run(fx, torch.float32)

    class Vocab:
        def idx2word(self, i):
            return "w%d" % i

    class AP:
        answer_vocab, EOS_IDX, BOS_IDX = Vocab(), 2, 1

        def get_true_vocab_size(self):
            return fx.V

    registry.register("vtextgqa_answer_processor", AP())
    model.config["metrics"] = ["textvqa_accuracy", "stvqa_anls"]
    model.init_losses_and_metrics()
    N = fx.F * fx.P
    s.context_tokens = [["tok%d_%d" % (b, i) for i in range(N)] for b in range(fx.B)]
    s.gt_answers = [["w5 w9"] * 10, ["tok1_3"] * 10]
    s.dataset_type = "val"
    out = model(s)
    assert set(out["metrics"]) == {"val/vtextgqa/textvqa_accuracy", "val/vtextgqa/stvqa_anls"}
    answers = decode_answers(out["pos_scores"].argmax(-1).cpu(), s.context_tokens, ["w%d" % i for i in range(fx.V)], fx.V, 2)
    entries = [{"pred_answer": a, "gt_answers": g} for a, g in zip(answers, s.gt_answers)]
    assert out["metrics"]["val/vtextgqa/textvqa_accuracy"].item() == pytest.approx(TextVQAAccuracyEvaluator().eval_pred_list([], entries)[1])
    assert out["metrics"]["val/vtextgqa/stvqa_anls"].item() == pytest.approx(STVQAANLSEvaluator().eval_pred_list([], entries)[1])
    assert set(out["losses"]) == {"val/vtextgqa/pos_bce_loss", "val/vtextgqa/InfoNCE"}
