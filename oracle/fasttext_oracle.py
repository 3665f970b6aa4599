"""CPU restatement (numpy, pure-Python loops: small cases only) of the FastText word-vector lookup that produces
``context_feature_0`` - TEST INFRASTRUCTURE: only tests/ may import this file; the product path is vitxt_gqa_amd/fasttext.py
+ csrc/fasttext.hip and fails loudly without the HIP extension.

Reference call site: ``FastTextProcessor._map_strings_to_indices`` (pythia/datasets/processors.py:478-491) ->
``WordToVectorDict.__getitem__`` (pythia/utils/vocab.py:375-381): ``np.mean([model.get_word_vector(w) for w in token.split(" ")], axis=0)``.
``model`` is the THIRD-PARTY ``fasttext`` Python module (``fasttext.load_model`` on ``wiki.en.bin``, processors.py:465-476), which is
absent from /root/reference and from this image and whose version the reference pins nowhere (requirement.txt lists none); the
published algorithm of fastText v0.9.x (``FastText::getWordVector`` / ``Dictionary::getSubwords`` / ``Dictionary::computeSubwords`` /
``Dictionary::hash``, and the ``.bin`` layout of ``FastText::saveModel``) is restated here.  PARITY UNPINNED by reference outputs:
no fasttext build exists here to generate vectors; the hash is pinned by the published FNV-1a test vectors (tests/test_fasttext_cpu.py).
"""
import struct

import numpy as np

MAGIC, VERSION = 793712314, 12
EOS, BOW, EOW = "</s>", "<", ">"


def fnv1a(data: bytes) -> int:
    """Dictionary::hash: 32-bit FNV-1a where each byte is first widened as a SIGNED char (uint32_t(int8_t(c)))."""
    h = 2166136261
    for c in data:
        h ^= (c - 256 if c >= 128 else c) & 0xFFFFFFFF
        h = (h * 16777619) & 0xFFFFFFFF
    return h


def read_model(path):
    """FastText::loadModel for a non-quantized model: args, dictionary, input matrix [nwords + bucket, dim]."""
    with open(path, "rb") as f:
        magic, version = struct.unpack("<ii", f.read(8))
        assert magic == MAGIC and version == VERSION, "not a fastText v12 .bin"
        dim, ws, epoch, min_count, neg, word_ngrams, loss, model, bucket, minn, maxn, lr_update = struct.unpack("<12i", f.read(48))
        (t,) = struct.unpack("<d", f.read(8))
        size, nwords, nlabels = struct.unpack("<3i", f.read(12))
        ntokens, prune = struct.unpack("<qq", f.read(16))
        words = []
        for _ in range(size):
            b = bytearray()
            while True:
                c = f.read(1)
                if c == b"\x00":
                    break
                b += c
            count, typ = struct.unpack("<qb", f.read(9))
            words.append((bytes(b).decode("utf-8"), count, typ))
        for _ in range(max(prune, 0)):
            f.read(8)
        (quant,) = struct.unpack("<?", f.read(1))
        assert not quant, "quantized models are not supported"
        m, n = struct.unpack("<qq", f.read(16))
        mat = np.frombuffer(f.read(m * n * 4), dtype="<f4").reshape(m, n).copy()
    return dict(dim=dim, bucket=bucket, minn=minn, maxn=maxn, nwords=nwords, words=[w for w, _, ty in words if ty == 0], matrix=mat)


def compute_subwords(word_bow_eow: str, minn, maxn, nwords, bucket):
    """Dictionary::computeSubwords: character n-grams (n = minn..maxn) of '<word>' over UTF-8 characters, skipping the
    1-grams '<' and '>'; id = nwords + hash(ngram) % bucket."""
    b = word_bow_eow.encode("utf-8")
    out = []
    i = 0
    while i < len(b):
        if (b[i] & 0xC0) == 0x80:          # continuation byte: n-grams start at character boundaries
            i += 1
            continue
        j, n = i, 1
        ngram = bytearray()
        while j < len(b) and n <= maxn:
            ngram.append(b[j])
            j += 1
            while j < len(b) and (b[j] & 0xC0) == 0x80:
                ngram.append(b[j])
                j += 1
            if n >= minn and not (n == 1 and (i == 0 or j == len(b))):
                out.append(nwords + fnv1a(bytes(ngram)) % bucket)
            n += 1
        i += 1
    return out


def subword_ids(m, word):
    """Dictionary::getSubwords(word): [word id] + n-gram ids for an in-vocabulary word (EOS: the word id alone), n-gram ids only
    otherwise."""
    w2i = m.setdefault("_w2i", {w: i for i, w in enumerate(m["words"])})
    wid = w2i.get(word, -1)
    if wid >= 0 and word == EOS:
        return [wid]
    grams = compute_subwords(BOW + word + EOW, m["minn"], m["maxn"], m["nwords"], m["bucket"]) if (m["maxn"] > 0 and word != EOS) else []
    return ([wid] if wid >= 0 else []) + grams


def get_word_vector(m, word):
    """FastText::getWordVector: zero; add the rows of the subword ids in order (float32); scale by float32(1.0 / count)."""
    ids = subword_ids(m, word)
    v = np.zeros(m["dim"], dtype=np.float32)
    for i in ids:
        v = (v + m["matrix"][i]).astype(np.float32)
    if ids:
        v = (v * np.float32(1.0 / len(ids))).astype(np.float32)
    return v


def token_vector(m, token):
    """WordToVectorDict.__getitem__ (vocab.py:379-381)."""
    return np.mean([get_word_vector(m, w) for w in token.split(" ")], axis=0)


def tokens_to_features(m, tokens, max_length):
    """FastTextProcessor._map_strings_to_indices (processors.py:478-491): [max_length, dim] fp32, rows past the tokens = PAD_INDEX (0)."""
    out = np.zeros((max_length, m["dim"]), dtype=np.float32)
    for i, tok in enumerate(tokens[:max_length]):
        out[i] = token_vector(m, tok)
    return out
