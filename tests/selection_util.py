"""Which selection decisions are robust to score rounding?  (test helper)

The grounding masks are integer work and are compared for EQUALITY - but only where the decision does not hinge on the
last bits of a floating-point score that two correct implementations may round differently (GPU fp32 vs CPU fp32 / fp64
oracle).  A row (sample, frame) of the spatial stage - or a sample of the temporal stage - is *decisive* when

  * every 2-way gumbel split in it has |g0 - g1| above ``tol_g`` (the split compares (s + g0) >= (s + g1) with s ~ 1e-3 and
    g ~ O(1): only the noise decides, up to the rounding of the sum and of log()), and
  * the k-th and (k+1)-th candidates of the top-k are either separated by more than ``tol`` RELATIVE to their size (the scores
    are softmax probabilities: two correct implementations differ by ~1e-6 .. 1e-5 relative, whatever the magnitude) plus an
    absolute ``atol`` (default 3e-45: two fp32 denormal quanta - peaked softmaxes produce values that small, and two of
    them that round to the same fp32 number are a tie on the GPU), or tie EXACTLY at
    the -10000 fill
    (exact ties are resolved by the shared lowest-index rule), for the "largest" and the "smallest"
    selection alike.

Follows spatio_temporal_grounding.py:34-68,79-142 as restated in oracle/t2s_oracle.py (temporal_grounding / spatial_grounding).
"""
import torch

NEG_FILL = -10000.0


def _split(score, expo, mask, tol_g):
    g = -torch.log(expo.double())
    pos = ((score.double() + g[:, 0]) >= (score.double() + g[:, 1])).double() * mask.double()
    neg = (1.0 - ((score.double() + g[:, 0]) >= (score.double() + g[:, 1])).double()) * mask.double()
    # an element whose mask is 0 is a filler whichever way the coin falls
    fragile = ((g[:, 0] - g[:, 1]).abs() <= tol_g * (1.0 + g.abs().amax(1))) & (mask != 0)
    fill = torch.full_like(score.double(), NEG_FILL)
    pos_s = torch.where(pos == 0, fill, score.double() * pos)
    neg_s = torch.where(neg == 0, fill, score.double() * neg)
    return pos_s, neg_s, fragile


def _boundary_ok(vals, k, largest, tol, atol):
    """vals [..., M]: is the k-th / (k+1)-th boundary of the top-k (largest or smallest) decided by more than the tolerance,
    or an exact tie of fillers?"""
    M = vals.shape[-1]
    if k >= M:
        return torch.ones(vals.shape[:-1], dtype=torch.bool)
    # a score below 1e-30 may have been flushed to zero somewhere inside an fp32 softmax / renormalisation: all of them count as 0
    vals = torch.where(vals.abs() < 1e-30, torch.zeros_like(vals), vals)
    s = torch.sort(vals, dim=-1, descending=largest).values
    a, b = s[..., k - 1], s[..., k]
    real = torch.minimum(a.abs(), b.abs()) < 1.0                    # at least one of the two is a score, not the -10000 fill
    scale = torch.where(a.abs() < 1.0, a.abs(), b.abs())
    scale = torch.where((a.abs() < 1.0) & (b.abs() < 1.0), torch.maximum(a.abs(), b.abs()), scale)
    return (real & ((a - b).abs() > tol * scale + atol)) | ((a == NEG_FILL) & (b == NEG_FILL))


def relative_diff(got, want):
    """max |got - want| / |want| over the entries that are scores (not fills, not exact zeros)."""
    m = (want.abs() < 1.0) & (want.abs() > 1e-30)        # normal fp32 range: a denormal's relative error says nothing
    return ((got.double() - want.double()).abs()[m] / want.double().abs()[m]).max().item() if m.any() else 0.0


def decisive_frames(frame_score, frame_mask, expo_frame, topk, tol=1e-4, atol=3e-45, tol_g=1e-6):
    """[B] bool: samples whose temporal selection (pos / neg frame top-k) is robust."""
    pos_s, neg_s, fragile = _split(frame_score, expo_frame, frame_mask, tol_g)
    return (~fragile.any(-1)) & _boundary_ok(pos_s, topk, True, tol, atol) & _boundary_ok(neg_s, topk, False, tol, atol)


def decisive_ocr_rows(ocr_score, new_mask, expo_ocr, topk, F, P, tol=1e-4, atol=3e-45, tol_g=1e-6):
    """[B, F] bool: (sample, frame) rows whose spatial selection (pos / neg OCR top-k of the frame's P slots) is robust."""
    B = ocr_score.shape[0]
    pos_s, neg_s, fragile = _split(ocr_score, expo_ocr, new_mask, tol_g)
    k = min(topk, P)
    return ((~fragile.view(B, F, P).any(-1)) & _boundary_ok(pos_s.view(B, F, P), k, True, tol, atol)
            & _boundary_ok(neg_s.view(B, F, P), k, False, tol, atol))
