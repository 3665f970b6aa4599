"""FastText lookup, host side (no GPU): the oracle's restatement of the third-party algorithm (oracle/fasttext_oracle.py) is
pinned by the published FNV-1a test vectors and by hand-built cases; the product's host code (vitxt_gqa_amd/fasttext.py: .bin
reader, rolling n-gram hashing, CSR encoder) must agree with it.  Reference call sites: processors.py:361-491, vocab.py:375-381."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from fasttext_util import TOKENS, WORDS, write_model  # noqa: E402
from oracle import fasttext_oracle as FO  # noqa: E402


def test_fnv1a_published_vectors_and_signed_byte_quirk():
    # FNV-1a 32-bit test vectors (Fowler / Noll / Vo reference suite)
    assert FO.fnv1a(b"") == 0x811C9DC5 and FO.fnv1a(b"a") == 0xE40C292C and FO.fnv1a(b"foobar") == 0xBF9CF968
    # fastText widens each byte as a SIGNED char before the xor: bytes >= 0x80 differ from plain FNV-1a
    plain = ((2166136261 ^ 0xC3) * 16777619) & 0xFFFFFFFF
    assert FO.fnv1a(b"\xc3") == ((2166136261 ^ 0xFFFFFFC3) * 16777619) & 0xFFFFFFFF != plain


def test_subwords_by_hand():
    # '<st>' with minn 3, maxn 6: n-grams '<st', 'st>', '<st>' (no 1-grams, nothing longer than the 4 characters)
    got = FO.compute_subwords("<st>", 3, 6, 100, 1000)
    want = [100 + FO.fnv1a(g.encode()) % 1000 for g in ("<st", "<st>", "st>")]
    assert got == want
    # UTF-8: n-grams are over CHARACTERS; 'é' is two bytes and never split
    got = FO.compute_subwords("<é>", 1, 2, 0, 1 << 32)
    want = [FO.fnv1a(g.encode("utf-8")) for g in ("<é", "é", "é>")]      # 1-grams '<' and '>' are skipped
    assert got == want


def test_reader_and_host_encoder_agree_with_the_oracle(tmp_path):
    from vitxt_gqa_amd import fasttext as FT
    path = str(tmp_path / "tiny.bin")
    mat = write_model(path, dim=20, bucket=500)
    m = FO.read_model(path)
    assert m["words"] == WORDS and m["dim"] == 20 and m["bucket"] == 500 and np.array_equal(m["matrix"], mat)
    tab = FT.FastTextTable.load(path, device="cpu")                       # reader + dictionary only: no kernel is launched here
    assert tab.words == WORDS and tab.dim == 20 and torch.equal(tab.table, torch.from_numpy(mat))
    for toks in TOKENS:
        for tok in toks:
            for w in tok.split(" "):
                assert tab.subword_ids(w) == FO.subword_ids(m, w), repr(w)
    assert tab.subword_ids("</s>") == [0]
    ids, wend, off = tab.encode(TOKENS, max_length=12)
    assert off.numel() == len(TOKENS) * 12 + 1 and int(off[-1]) == ids.numel() == wend.numel()
    # slot of 'the  of' (two spaces): words 'the', '', 'of' -> three word ends; the empty word has the n-gram '<>' ... of length 2 < minn: no rows
    s = TOKENS[0].index("the  of")
    assert int(wend[off[s]:off[s + 1]].sum()) == 3 and (ids[off[s]:off[s + 1]] == -1).sum() == 1
    assert int(off[12 + 3]) == int(off[12 + 12]) and int(off[24]) == int(off[36])          # slots past a sample's tokens are empty
    # the oracle on the same tokens (shape + padding rule of processors.py:478-491)
    f = FO.tokens_to_features(m, TOKENS[0], 12)
    assert f.shape == (12, 20) and np.abs(f[:11]).sum() > 0 and not f[11].any()
    one = FO.token_vector(m, "main st")
    assert np.allclose(one, (FO.get_word_vector(m, "main") + FO.get_word_vector(m, "st")) / 2)
    with pytest.raises(ValueError):
        FT.FastTextTable(WORDS, torch.zeros(3, 20), 500, 3, 6, device="cpu")


GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_golden_vectors_of_the_real_fasttext_library():
    """Consumes tests/golden/fasttext_vectors.npz + fasttext_tiny.bin when a maintainer has generated them with the REAL fasttext
    module (tests/golden/make_fasttext_golden.py - the wheel exists neither in the reference checkout nor in the build image).
    Until then this test skips and the FastText row stays "parity unpinned" (DESIGN section 2)."""
    vec, binf = os.path.join(GOLDEN, "fasttext_vectors.npz"), os.path.join(GOLDEN, "fasttext_tiny.bin")
    if not (os.path.exists(vec) and os.path.exists(binf)):
        pytest.skip("no fasttext golden vectors: run tests/golden/make_fasttext_golden.py where the fasttext wheel is installed")
    from vitxt_gqa_amd import fasttext as FT
    z = np.load(vec, allow_pickle=True)
    m = FO.read_model(binf)
    tab = FT.FastTextTable.load(binf, device="cpu")
    for w, want in zip(z["words"], z["word_vectors"]):
        assert np.allclose(FO.get_word_vector(m, str(w)), want, atol=1e-6), repr(w)            # the oracle vs the library
        assert tab.subword_ids(str(w)) == FO.subword_ids(m, str(w)), repr(w)                   # the product's host encoder vs the oracle
    for t, want in zip(z["tokens"], z["token_vectors"]):
        assert np.allclose(FO.token_vector(m, str(t)), want, atol=1e-6), repr(t)
