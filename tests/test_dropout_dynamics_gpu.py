"""Training dynamics with the in-kernel attention-dropout masks (a stateless hash: rowkey16 * colkey16, csrc/attn_common.h) against
i.i.d. Philox Bernoulli masks.  tools/dropout_dynamics.py runs the full comparison (5 seeds x 200 steps at the cfg1
shape: profiles/r05_dropout_dynamics.txt - largest gap 1.0 pooled standard deviations of a 20-step window mean); this test runs a short
version of the same three arms and asserts that the loss envelopes overlap."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_hash_masks_train_like_iid_masks():
    import dropout_dynamics as DD
    seeds, steps, window = 3, 60, 20
    res = DD.run(seeds, steps, window=window)
    txt, worst = DD.table(res, window=window)
    print(txt)
    h, i, k = res["hash"], res["iid"], res["kernel"]
    assert torch.isfinite(h).all() and torch.isfinite(i).all() and torch.isfinite(k).all()
    # the loss falls (the run optimises), in every arm
    for arm in (h, i, k):
        assert arm[:, -1].mean().item() < 0.8 * arm[:, 0].mean().item()
    # envelopes overlap: window means of the two mask kinds within 3 pooled standard deviations (+ 0.2 % of the loss: with 3 seeds the
    # sample deviation of a window can be tiny), window by window
    for w in range(h.shape[1]):
        sd = ((h[:, w].var() + i[:, w].var()) / 2).sqrt().item()
        gap = abs(h[:, w].mean().item() - i[:, w].mean().item())
        assert gap <= 3 * sd + 2e-3 * h[:, w].mean().item(), "window %d: hash %.3f vs iid %.3f (pooled sd %.3f)" % (w, h[:, w].mean().item(), i[:, w].mean().item(), sd)
    # the product kernels draw the very masks of the "hash" arm: same trajectory up to bf16-vs-fp32 attention arithmetic
    assert (k.mean(0) - h.mean(0)).abs().max().item() <= 0.02 * h.mean(0).max().item()
