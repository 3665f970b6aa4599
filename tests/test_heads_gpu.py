"""GPU parity of the pointer-network head and the grounding scorer / selection kernels against the CPU oracle's
restatement of the reference (oracle/t2s_oracle.py).  Run on the MI355X box: pytest -m gpu."""
import math

import pytest
import torch

from oracle import t2s_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


@pytest.mark.parametrize("kdt,exact,tol", [(torch.float32, True, 2e-5), (torch.float32, False, 2e-2), (torch.bfloat16, False, 2e-2)])
@pytest.mark.parametrize("B,D,N,V", [(2, 12, 48, 64), (3, 12, 600, 1000), (1, 12, 1001, 7)])
def test_ptr_scores(B, D, N, V, kdt, exact, tol):
    _need_gpu()
    from vitxt_gqa_amd import ops
    g = torch.Generator().manual_seed(N)
    q = torch.randn(B, D, 768, generator=g)
    k = torch.randn(B, N, 768, generator=g).to(kdt)
    mask = (torch.rand(B, N, generator=g) < 0.5).float()
    out = torch.full((B, D, V + N), 7.0)
    ref = (q.double() @ k.double().transpose(1, 2)) / math.sqrt(768) + mask.double().unsqueeze(1)
    got = ops.ptr_scores(q.to(DEV), k.to(DEV), mask.to(DEV), out.to(DEV), V, exact_fp32=exact).cpu()
    assert (got[:, :, :V] == 7.0).all()                       # classifier columns untouched
    assert (got[:, :, V:].double() - ref).abs().max().item() < tol * max(1.0, ref.abs().max().item() / 4)


def test_ptr_logits_autograd():
    _need_gpu()
    from vitxt_gqa_amd import functional as FN
    g = torch.Generator().manual_seed(1)
    B, D, N, V = 2, 12, 100, 30
    fixed = torch.randn(B, D, V, generator=g)
    q = torch.randn(B, D, 768, generator=g)
    k = torch.randn(B, N, 768, generator=g)
    mask = (torch.rand(B, N, generator=g) < 0.5).float()
    dl = torch.randn(B, D, V + N, generator=g)
    fr, qr, kr = [t.double().requires_grad_(True) for t in (fixed, q, k)]
    ref = torch.cat([fr, qr @ kr.transpose(1, 2) / math.sqrt(768) + mask.double().unsqueeze(1)], -1)
    gref = torch.autograd.grad(ref, (fr, qr, kr), dl.double())
    fg, qg, kg = [t.to(DEV).requires_grad_(True) for t in (fixed, q, k)]
    out = FN.ptr_logits(fg, qg, kg, mask.to(DEV))
    assert (out.double().cpu() - ref).abs().max().item() < 1e-4
    out.backward(dl.to(DEV))
    for a, b in zip((fg, qg, kg), gref):
        assert (a.grad.double().cpu() - b).abs().max().item() < 1e-4 * max(1.0, b.abs().max().item())


def test_question_pool_and_scores():
    _need_gpu()
    from vitxt_gqa_amd import ops
    g = torch.Generator().manual_seed(2)
    B, T, M = 3, 20, 333
    sd = {"Grounding_Module.q_linear.weight": torch.randn(768, 768, generator=g) * 0.05,
          "Grounding_Module.q_linear.bias": torch.randn(768, generator=g) * 0.1,
          "Grounding_Module.self_attn.weight": torch.randn(1, 768, generator=g) * 0.1,
          "Grounding_Module.self_attn.bias": torch.randn(1, generator=g)}
    qf = torch.randn(B, T, 768, generator=g)
    qm = O.get_mask(torch.tensor([20, 7, 1]), T)
    ref_gq = O.question_pool({k: v.double() for k, v in sd.items()}, qf.double(), qm.double())       # [B, 1, 768]
    qp = torch.nn.functional.linear(qf, sd["Grounding_Module.q_linear.weight"], sd["Grounding_Module.q_linear.bias"])
    gq = ops.question_pool(qp.to(DEV).contiguous(), sd["Grounding_Module.self_attn.weight"].view(-1).to(DEV),
                           sd["Grounding_Module.self_attn.bias"].to(DEV), qm.to(DEV))
    assert (gq.double().cpu() - ref_gq.squeeze(1)).abs().max().item() < 1e-4
    for kdt, tol in ((torch.float32, 1e-6), (torch.bfloat16, 2e-4)):
        k = (torch.randn(B, M, 768, generator=g) * 0.2).to(kdt)
        mask = (torch.rand(B, M, generator=g) < 0.6).float()
        mask[2] = 0                                             # fully masked sample: every entry -10000
        ref = O.attention_score(ref_gq, k.double(), mask.double())
        got = ops.attention_score(ref_gq.squeeze(1).float().to(DEV).contiguous(), k.to(DEV), mask.to(DEV)).cpu()
        assert (got.double() - ref).abs().max().item() < tol
        assert (got[2] == -10000.0).all()


@pytest.mark.parametrize("B,F,P", [(2, 6, 8), (3, 20, 30), (2, 100, 100)])
def test_ground_select_matches_oracle_rule(B, F, P):
    """Selection kernel == oracle's selection (same injected noise, same lowest-index tie rule), bit-exact masks."""
    _need_gpu()
    from vitxt_gqa_amd import ops
    g = torch.Generator().manual_seed(F)
    N = F * P
    gq = torch.randn(B, 1, 768, generator=g) * 0.3
    frame_feat = torch.randn(B, F, 768, generator=g) * 0.3
    ocr_feat = torch.randn(B, N, 768, generator=g) * 0.3
    frame_mask = torch.ones(B, F)
    frame_mask[0, F - 1] = 0
    frame_id = torch.arange(1, F + 1).repeat(B, 1)
    temporal_id = frame_id.repeat_interleave(P, dim=1)
    bbox = torch.rand(B, N, 4, generator=g)
    e1 = torch.empty(B, 2, F).exponential_(generator=g)
    e2 = torch.empty(B, 2, N).exponential_(generator=g)
    f_score = O.attention_score(gq, frame_feat, frame_mask)
    gf, pos_f, neg_f = O.temporal_grounding(f_score, frame_mask, frame_id, e1, 5)
    newm = O.new_ocr_mask_from_frames(gf, temporal_id)
    o_score = O.attention_score(gq, ocr_feat, newm)
    box, pos_o, neg_o = O.spatial_grounding(o_score, bbox, newm, e2, 5, F, P)
    fs_gpu = ops.attention_score(gq.squeeze(1).to(DEV).contiguous(), frame_feat.to(DEV), frame_mask.to(DEV))
    assert (fs_gpu.cpu() - f_score).abs().max().item() < 1e-6
    # feed the oracle's fp32 frame scores so that the comparison isolates the selection logic
    sel = ops.ground_select(f_score.to(DEV), frame_mask.to(DEV), e1.to(DEV), frame_id.to(DEV), gq.squeeze(1).to(DEV).contiguous(),
                            ocr_feat.to(DEV), e2.to(DEV), temporal_id.to(DEV), bbox.to(DEV), F, P, 5, 5)
    assert torch.equal(sel["pos_obj_mask"].cpu(), pos_f * frame_mask)
    assert torch.equal(sel["neg_obj_mask"].cpu(), neg_f * frame_mask)
    assert torch.equal(sel["ground_frame"].cpu(), gf)
    assert torch.equal(sel["new_ocr_mask"].cpu(), newm)
    assert (sel["ocr_score"].cpu() - o_score).abs().max().item() < 1e-6
    # OCR selection masks are integer work: EQUALITY on every (sample, frame) row whose decisions (gumbel coin flips, the
    # k-th / (k+1)-th boundary of both top-k selections) are separated by more than the score rounding (selection_util);
    # rows that hinge on the last bits of a score are excluded, exact -10000 ties follow the shared lowest-index rule
    from selection_util import decisive_ocr_rows
    ok = decisive_ocr_rows(o_score, newm, e2, 5, F, P)
    assert ok.float().mean().item() > 0.9, "too few decisive rows for the test to mean anything"
    okn = ok.unsqueeze(-1).expand(B, F, P).reshape(B, N)
    assert torch.equal(sel["pos_ocr_mask"].cpu()[okn], pos_o[okn])
    assert torch.equal(sel["neg_ocr_mask"].cpu()[okn], neg_o[okn])
    assert sel["pos_ocr_mask"].sum(1).tolist() == [5.0 * F] * B
    assert sel["ground_box"].shape == (B, 5 * F, 4)
    okb = ok.unsqueeze(-1).expand(B, F, 5).reshape(B, F * 5)
    assert torch.equal(sel["ground_box"].cpu()[okb], box[okb])
    print("decisive rows: %d of %d; all-equal masks: %s" % (int(ok.sum()), ok.numel(), torch.equal(sel["pos_ocr_mask"].cpu(), pos_o)))


def test_embed_rows():
    _need_gpu()
    from vitxt_gqa_amd import functional as FN
    g = torch.Generator().manual_seed(3)
    B, N = 2, 37
    f0 = torch.randn(B, N, 300, generator=g)
    f1 = (torch.rand(B, N, 604, generator=g) < 0.1).float()
    f1[0, 0] = 0                                             # all-zero PHOC row: x / max(||x||, 1e-12) = 0
    id0 = torch.randint(0, 4000, (B, N), generator=g)
    id1 = torch.randint(0, 50, (B, N), generator=g)
    e0 = torch.randn(4000, 50, generator=g)
    e1 = torch.randn(4000, 50, generator=g)
    ref = torch.cat([O.l2_normalize(f0.double()), O.l2_normalize(f1.double()), e0.double()[id0], e1.double()[id1]], -1)
    e0g, e1g = e0.to(DEV).requires_grad_(True), e1.to(DEV).requires_grad_(True)
    for dt, tol in ((torch.float32, 1e-6), (torch.bfloat16, 2e-2)):
        out = FN.embed_rows(f0.to(DEV), f1.to(DEV), id0.to(DEV), e0g, id1.to(DEV), e1g, dt)
        assert out.dtype == dt and out.shape == (B, N, 1004)
        assert (out.double().cpu() - ref).abs().max().item() < tol
    gout = torch.randn(B, N, 1004, generator=g)
    out.float().backward(gout.to(DEV))
    r0 = torch.zeros(4000, 50).index_add_(0, id0.reshape(-1), gout.reshape(-1, 1004)[:, 904:954])
    r1 = torch.zeros(4000, 50).index_add_(0, id1.reshape(-1), gout.reshape(-1, 1004)[:, 954:1004])
    assert (e0g.grad.cpu() - r0).abs().max().item() < 2e-2 and (e1g.grad.cpu() - r1).abs().max().item() < 2e-2
    # frames: single feature + single id table
    v = torch.randn(B, 5, 1024, generator=g)
    fid = torch.arange(1, 6).repeat(B, 1)
    o2 = FN.embed_rows(v.to(DEV), None, fid.to(DEV), e0g, None, None, torch.float32)
    assert (o2.double().cpu() - torch.cat([O.l2_normalize(v.double()), e0.double()[fid]], -1)).abs().max().item() < 1e-6


def test_losses_match_oracle():
    _need_gpu()
    from vitxt_gqa_amd import SampleList
    from vitxt_gqa_amd.losses import POSBCEWithMaskLoss, InfoNCE
    g = torch.Generator().manual_seed(4)
    B, D, C = 3, 12, 1601
    ref, pos, neg = [torch.randn(B, D, C, generator=g) * 2 for _ in range(3)]
    targets = (torch.rand(B, D, C, generator=g) < 0.01).float()
    mask = (torch.rand(B, D, generator=g) < 0.7).float()
    rd, pd, nd = [t.double().requires_grad_(True) for t in (ref, pos, neg)]
    l_ref, a_ref, b_ref = O.total_loss(dict(ref_scores=rd, pos_scores=pd, neg_scores=nd), targets.double(), mask.double())
    g_ref = torch.autograd.grad(l_ref, (rd, pd, nd))
    rg, pg, ng = [t.to(DEV).requires_grad_(True) for t in (ref, pos, neg)]
    s = SampleList({"targets": targets.to(DEV), "train_loss_mask": mask.to(DEV)})
    out = {"ref_scores": rg, "pos_scores": pg, "neg_scores": ng}
    a = POSBCEWithMaskLoss()(s, out)
    b = InfoNCE()(s, out)
    assert abs(a.item() - a_ref.item()) < 1e-4 * abs(a_ref.item())
    assert abs(1000 * b.item() - b_ref.item()) < 1e-3 * abs(b_ref.item()) + 1e-4
    (a + 1000 * b).backward()
    for got, want in zip((rg, pg, ng), g_ref):
        sc = want.abs().max().item()
        assert (got.grad.double().cpu() - want).abs().max().item() < 1e-4 * sc + 1e-9
    # all-zero mask: count clamps to 1, loss 0
    s0 = SampleList({"targets": targets.to(DEV), "train_loss_mask": torch.zeros(B, D, device=DEV)})
    assert POSBCEWithMaskLoss()(s0, out).item() == 0.0
