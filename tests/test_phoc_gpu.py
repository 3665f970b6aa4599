"""t2s_phoc (vitxt_gqa_amd/csrc/phoc.hip, through the C ABI) against the oracle and the reference-generated golden vectors:
byte work, so the bar is bit-exact.  Edge cases: empty token, single symbol, tokens longer than one 64-lane chunk,
every token length 1..200 (the i/n roundings), strided output rows, a foreign symbol."""
import os
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def test_phoc_kernel_matches_golden_and_oracle():
    _need_gpu()
    from oracle import phoc_oracle as po
    from vitxt_gqa_amd import phoc as P
    d = np.load(os.path.join(HERE, "golden", "phoc_words.npz"))
    raw = [str(t) for t in d["raw"]]
    exp = np.unpackbits(d["bits"], axis=1)[:, :604].astype(np.float32)
    slots = P.pack_tokens(raw, len(raw), width=64)
    out = P.phoc_features(slots).cpu().numpy()
    assert out.dtype == np.float32 and np.array_equal(out, exp)
    # every length 1..200 and random words up to 250 symbols (4 lane chunks), width 256
    rnd = random.Random(9)
    alpha = "abcdefghijklmnopqrstuvwxyz0123456789"
    common = "th he in er an re es on st nt en at ed nd to or ea ti ar te ng al it as is ha et se ou of le".split()
    words = ["a" * n for n in range(1, 201)] + ["".join(rnd.choice(alpha) for _ in range(rnd.randint(1, 250))) for _ in range(3000)]
    words += ["".join(rnd.choice(common) for _ in range(rnd.randint(1, 60))) for _ in range(1000)]
    slots = P.pack_tokens(words, len(words), width=256)
    out = P.phoc_features(slots).cpu().numpy()
    assert np.array_equal(out, po.build_phoc_batch(slots))
    assert set(np.unique(out)) == {0.0, 1.0}


def test_phoc_kernel_strided_rows_batch_shape_and_foreign_symbol():
    _need_gpu()
    from oracle import phoc_oracle as po
    from vitxt_gqa_amd import ops, phoc as P
    B, N = 3, 7
    toks = [["stop", "EXIT", "24/7", "", "Coca-Cola", "x" * 40, "the"] for _ in range(B)]
    slots = np.stack([P.pack_tokens(t, N, width=48) for t in toks])
    dev = torch.from_numpy(slots).cuda()
    wide = torch.full((B, N, 640), -7.0, device="cuda")            # rows of a wider buffer: only the first 604 are written
    out = ops.phoc(dev, out=wide[..., :604])
    assert out.data_ptr() == wide.data_ptr()
    assert np.array_equal(wide[..., :604].cpu().numpy().reshape(-1, 604), po.build_phoc_batch(slots.reshape(-1, 48)))
    assert (wide[..., 604:] == -7.0).all()
    bad = slots.copy()
    bad[1, 2, 0] = ord("-")
    with pytest.raises(RuntimeError):
        P.phoc_features(bad)                                       # the host wrapper refuses, like the reference raises
    o = ops.phoc(torch.from_numpy(bad).cuda())                     # the raw ABI poisons exactly that row
    assert torch.isnan(o[1, 2]).all() and torch.isfinite(o[0]).all() and torch.isfinite(o[1, 3:]).all()
    assert ops.phoc(torch.zeros(0, 16, dtype=torch.uint8, device="cuda")).shape == (0, 604)
