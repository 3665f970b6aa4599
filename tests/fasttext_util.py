"""Writes a small synthetic fastText v12 ``.bin`` (``FastText::saveModel`` layout) for the tests: there is no fasttext library
and no wiki.en.bin offline, so the file format, the dictionary and the matrix are exercised on a model of our own making."""
import struct

import numpy as np

WORDS = ["</s>", "the", "exit", "stop", "main", "st", "42", "<pad>", "café", "路", "a", "of", "station", "street", "stop-sign"]


def write_model(path, dim=300, bucket=2000, minn=3, maxn=6, seed=0, words=WORDS):
    rng = np.random.default_rng(seed)
    mat = (rng.standard_normal((len(words) + bucket, dim)) * 0.3).astype("<f4")
    with open(path, "wb") as f:
        f.write(struct.pack("<ii", 793712314, 12))
        f.write(struct.pack("<12i", dim, 5, 5, 5, 5, 1, 1, 2, bucket, minn, maxn, 100))      # ws, epoch, minCount, neg, wordNgrams, loss, model, ...
        f.write(struct.pack("<d", 1e-4))
        f.write(struct.pack("<3i", len(words), len(words), 0))
        f.write(struct.pack("<qq", 12345, -1))
        for i, w in enumerate(words):
            f.write(w.encode("utf-8") + b"\x00")
            f.write(struct.pack("<qb", 1000 - i, 0))
        f.write(struct.pack("<?", False))
        f.write(struct.pack("<qq", mat.shape[0], mat.shape[1]))
        f.write(mat.tobytes())
        f.write(struct.pack("<?", False))                                  # output matrix (unused by the lookup)
        f.write(struct.pack("<qq", len(words), dim))
        f.write(np.zeros((len(words), dim), dtype="<f4").tobytes())
    return mat


TOKENS = [["exit", "main st", "42", "café", "zzzqqq", "路路", "stop-sign", "a", "", "x", "the  of"], ["<pad>"] * 3, []]
