#!/usr/bin/env python3
"""Generates tests/golden/fasttext_vectors.npz + tests/golden/fasttext_tiny.bin from the REAL `fasttext` module - the
third-party dependency whose algorithm oracle/fasttext_oracle.py restates (pythia/datasets/processors.py:361-491 loads the model
with ``fasttext.load_model``, pythia/utils/vocab.py:375-381 averages ``model.get_word_vector`` over the space-separated words).

That module is absent from the reference checkout and from the build image (no network), so this script cannot run there and the
FastText row stays "parity unpinned" until someone runs it where the wheel exists:

    pip install fasttext            # or fasttext-wheel; any 0.9.x
    python tests/golden/make_fasttext_golden.py
    python -m pytest tests/test_fasttext_cpu.py -k golden      # oracle + host encoder vs the library's own vectors
    python -m pytest tests/test_fasttext_gpu.py -k golden      # the HIP kernel vs the same vectors (on a GPU box)

It trains a tiny skip-gram model (dim 16, bucket 2000, minn 3, maxn 6 - the .bin v12 layout of wiki.en.bin, which is what the
reference's config names) on a synthetic corpus written below, saves the .bin (a few hundred KB) and stores, for a list of probe
tokens - in-vocabulary words, out-of-vocabulary words (pure n-gram vectors), multi-word OCR tokens, non-ASCII tokens, the empty
word between two spaces - the library's ``get_word_vector`` of every word and the reference's token rule (mean over words).
Fixtures are data: the probe tokens, the library's vectors and the model file; no reference source text.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

PROBES = ["the", "street", "main st", "exit 12", "zzzyqx", "a", "st", "the  of", "café", "über straße", "OPEN", "24/7",
          "northbound", "x" * 40, "coca cola zero", "</s>"]


def corpus(path, rng):
    words = ["the", "of", "and", "street", "main", "exit", "north", "south", "bound", "open", "closed", "coffee", "café", "shop",
             "st", "ave", "road", "12", "24", "7", "stop", "bus", "station", "market", "straße", "über", "cola", "coca", "zero"]
    with open(path, "w", encoding="utf-8") as f:
        for _ in range(4000):
            f.write(" ".join(rng.choice(words, size=rng.integers(4, 12))) + "\n")


def main():
    try:
        import fasttext
    except ImportError:
        print("make_fasttext_golden.py: the `fasttext` module is not installed here - nothing written (see the docstring)", file=sys.stderr)
        return 2
    rng = np.random.default_rng(0)
    txt = os.path.join(HERE, "_fasttext_corpus.txt")
    corpus(txt, rng)
    model = fasttext.train_unsupervised(txt, model="skipgram", dim=16, bucket=2000, minn=3, maxn=6, epoch=2, minCount=1, thread=1, seed=0)
    os.remove(txt)
    bin_path = os.path.join(HERE, "fasttext_tiny.bin")
    model.save_model(bin_path)
    model = fasttext.load_model(bin_path)                      # what FastTextProcessor does (processors.py:409-420)
    word_vecs, token_vecs, words = [], [], []
    for tok in PROBES:
        parts = tok.split(" ")
        vs = [model.get_word_vector(w) for w in parts]
        token_vecs.append(np.mean(vs, axis=0))                 # WordToVectorDict.__getitem__ (vocab.py:375-381)
        for w, v in zip(parts, vs):
            words.append(w)
            word_vecs.append(v)
    np.savez_compressed(os.path.join(HERE, "fasttext_vectors.npz"), tokens=np.array(PROBES, dtype=object), token_vectors=np.stack(token_vecs),
                        words=np.array(words, dtype=object), word_vectors=np.stack(word_vecs),
                        meta=np.array([fasttext.__name__, getattr(fasttext, "__version__", "unknown")], dtype=object))
    print("wrote fasttext_tiny.bin (%d bytes) and fasttext_vectors.npz (%d tokens, %d words)" % (os.path.getsize(bin_path), len(PROBES), len(words)))
    return 0


if __name__ == "__main__":
    sys.exit(main())
