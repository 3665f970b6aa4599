"""Synthetic T2S batches with the schema ``VTEXTGQADataset`` produces (SURVEY.md section 8d,
Appendix B; reference ``pythia/datasets/videoqa/vtextgqa/dataset.py:83-312``).

There is no dataset offline, so benches and tests use seeded random features of the right
shape/dtype/range.  Constraints kept from the reference: ``N = F*P``; ``temporal_id[f*P+j] ==
frame_id[f]`` (required by t2s.py:490-492); ``frame_id`` in [1, 4000); ``text_len >= 1``;
padded OCR slots still carry non-zero features.
"""
import torch

T_MAX = 20
DEC_STEPS = 12


def make_batch(B, F, P, V=5000, seed=0, text_vocab=30522, ocr_keep=0.7, bos_idx=1, full_targets=True, ocr_prev_frac=0.0,
               text_len=None):
    """``ocr_keep``: density of ``ocr_mask`` - a number, or one per sample (samples with different key counts in one batch);
    ``text_len``: per-sample question lengths instead of the drawn ones.  Neither changes the random stream of the other fields."""
    g = torch.Generator().manual_seed(seed)
    N = F * P
    s = {}
    s["text"] = torch.randint(0, text_vocab, (B, T_MAX), generator=g)
    s["text_len"] = torch.randint(5, T_MAX + 1, (B,), generator=g)
    s["video_feat"] = torch.randn(B, F, 1024, generator=g)
    s["frame_id"] = torch.arange(1, F + 1).repeat(B, 1)
    s["frame_mask"] = torch.ones(B, F, dtype=torch.long)
    s["context_feature_0"] = torch.randn(B, N, 300, generator=g)
    s["context_feature_1"] = (torch.rand(B, N, 604, generator=g) < 0.1).float()
    s["temporal_id"] = s["frame_id"].repeat_interleave(P, dim=1)
    s["track_id"] = torch.randint(0, 50, (B, N), generator=g)
    bb = torch.rand(B, N, 2, 2, generator=g).sort(dim=2).values        # x1<=x2, y1<=y2
    s["ocr_bbox_coordinates"] = torch.stack([bb[:, :, 0, 0], bb[:, :, 0, 1], bb[:, :, 1, 0], bb[:, :, 1, 1]], -1)
    keep = torch.as_tensor(ocr_keep, dtype=torch.float32).reshape(-1, 1)          # [1, 1] or [B, 1]
    s["ocr_mask"] = (torch.rand(B, N, generator=g) < keep).long()
    if text_len is not None:
        s["text_len"] = torch.as_tensor(text_len, dtype=torch.long).reshape(B)
    s["train_prev_inds"] = torch.randint(0, V, (B, DEC_STEPS), generator=g)
    s["train_prev_inds"][:, 0] = bos_idx
    if full_targets:
        s["targets"] = (torch.rand(B, DEC_STEPS, V + N, generator=g) < 1e-3).float()
    s["train_loss_mask"] = torch.ones(B, DEC_STEPS)
    if ocr_prev_frac > 0:
        # teacher-forced OCR copies: previous-step indices in [V, V+N) (PrevPredEmbeddings' OCR branch, t2s.py:690-723);
        # drawn LAST so that every other field is the same stream as without them
        pick = torch.rand(B, DEC_STEPS, generator=g) < ocr_prev_frac
        pick[:, 0] = False
        ocr_ix = V + torch.randint(0, N, (B, DEC_STEPS), generator=g)
        s["train_prev_inds"] = torch.where(pick, ocr_ix, s["train_prev_inds"])
    return s


def make_noise(B, F, P, seed=0):
    """The two exponential draws consumed by the reference's ``F.gumbel_softmax`` calls, in order
    [B,2,F] then [B,2,N] (Appendix A, Q8; Appendix E)."""
    g = torch.Generator().manual_seed(seed + 7919)
    return (torch.empty(B, 2, F).exponential_(generator=g),
            torch.empty(B, 2, F * P).exponential_(generator=g))


def make_token_slots(B, N, width=64, seed=0, max_len=14):
    """Synthetic normalised OCR tokens as NUL-padded byte slots uint8 [B, N, width] (what ``phoc.pack_tokens`` produces
    from real strings): lengths uniform in 1..max_len, symbols uniform in [a-z0-9]."""
    g = torch.Generator().manual_seed(seed + 104729)
    sym = torch.tensor(list(b"abcdefghijklmnopqrstuvwxyz0123456789"), dtype=torch.uint8)
    ch = sym[torch.randint(0, 36, (B, N, width), generator=g)]
    ln = torch.randint(1, max_len + 1, (B, N, 1), generator=g)
    return torch.where(torch.arange(width).view(1, 1, width) < ln, ch, torch.zeros((), dtype=torch.uint8))
