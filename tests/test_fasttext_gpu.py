"""t2s_fasttext_rows (table resident in HBM, gather + average on the GPU) bit-exact against the CPU restatement of the fastText
lookup (oracle/fasttext_oracle.py), dim 300 as wiki.en.bin and a small odd dim, UTF-8 / multi-word / out-of-vocabulary / empty tokens."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from fasttext_util import TOKENS, write_model  # noqa: E402
from oracle import fasttext_oracle as FO  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dim,bucket", [(300, 2000), (20, 64)])
def test_fasttext_rows_bit_exact(tmp_path, dim, bucket):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from vitxt_gqa_amd.fasttext import FastTextTable
    path = str(tmp_path / "m.bin")
    write_model(path, dim=dim, bucket=bucket, seed=dim)
    m = FO.read_model(path)
    tab = FastTextTable.load(path, device="cuda:0")
    rng = np.random.default_rng(1)
    alphabet = list("abcdefghijklmnopqrstuvwxyz0123456789-'") + ["é", "ü", "路"]
    rand_tokens = ["".join(rng.choice(alphabet, size=int(rng.integers(1, 15)))) + (" " + "".join(rng.choice(alphabet, size=3)) if rng.random() < 0.2 else "")
                   for _ in range(150)]
    batch = TOKENS + [rand_tokens[:100], rand_tokens[100:]]
    L = 100
    got = tab.features(batch, L).cpu().numpy()
    assert got.shape == (len(batch), L, dim)
    for b, toks in enumerate(batch):
        want = FO.tokens_to_features(m, toks, L)
        assert np.array_equal(got[b], want), (b, np.abs(got[b] - want).max())
    # a preallocated output (e.g. the arena field the model reads) and the all-empty batch
    out = torch.full((2, 5, dim), 7.0, device="cuda:0")
    assert tab.features([[], []], 5, out=out).abs().max().item() == 0


def test_golden_vectors_of_the_real_fasttext_library_on_the_gpu():
    """The HIP gather + average against vectors of the REAL fasttext module, when tests/golden/make_fasttext_golden.py has been run
    where that wheel exists (it is in neither the reference checkout nor the build image); skips otherwise."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    vec, binf = os.path.join(golden, "fasttext_vectors.npz"), os.path.join(golden, "fasttext_tiny.bin")
    if not (os.path.exists(vec) and os.path.exists(binf)):
        pytest.skip("no fasttext golden vectors: run tests/golden/make_fasttext_golden.py where the fasttext wheel is installed")
    from vitxt_gqa_amd.fasttext import FastTextTable
    z = np.load(vec, allow_pickle=True)
    tab = FastTextTable.load(binf, device="cuda:0")
    toks = [str(t) for t in z["tokens"]]
    got = tab.features([toks], len(toks)).cpu().numpy()[0]
    assert np.allclose(got, z["token_vectors"], atol=1e-6)
