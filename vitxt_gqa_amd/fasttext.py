"""FastText OCR-token vectors (``context_feature_0``) with the table resident in HBM.

The reference looks every OCR token up on the host through the third-party ``fasttext`` module (``FastTextProcessor``,
pythia/datasets/processors.py:361-491; ``WordToVectorDict``, pythia/utils/vocab.py:375-381) and ships the [B, N, 300] fp32 rows
over PCIe (0.77 GB per B=64 batch of the 100 x 100 shape: the largest host->device field left).  Here the model's input
matrix ((nwords + bucket) x 300 fp32: 4.8 GB for wiki.en.bin - trivial beside 288 GB of HBM3E) is uploaded ONCE; per batch
the host computes only the subword row ids of every token (string hashing: a few dozen int32 per token) and the
``t2s_fasttext_rows`` kernel gathers and averages the rows on the GPU, in fastText's own order of operations.

Host side = this file: the ``.bin`` reader (``FastText::loadModel`` layout, non-quantized models), ``Dictionary::getSubwords``
(FNV-1a hashing of the UTF-8 character n-grams of ``<word>``), the CSR batch encoder.  There is no CPU lookup path here
(oracle/fasttext_oracle.py is the checker used by the tests)."""
import struct

import numpy as np
import torch

from . import hipext as X

MAGIC, VERSION = 793712314, 12
EOS = "</s>"


def _fnv1a_ngrams(word_bytes, minn, maxn):
    """Hashes of the character n-grams of the UTF-8 byte string ``<word>`` (Dictionary::computeSubwords + Dictionary::hash: each
    byte enters the 32-bit FNV-1a as a SIGNED char).  Rolling per start position: the hash of an n-gram extends that of its
    (n-1)-gram by the bytes of one more character."""
    b = word_bytes
    n_b = len(b)
    starts = [i for i in range(n_b) if (b[i] & 0xC0) != 0x80]
    out = []
    for si, i in enumerate(starts):
        h = 2166136261
        j = i
        for n in range(1, maxn + 1):
            if si + n - 1 >= len(starts):
                break
            end = starts[si + n] if si + n < len(starts) else n_b
            while j < end:
                c = b[j]
                h ^= (c | 0xFFFFFF00) if c >= 128 else c
                h = (h * 16777619) & 0xFFFFFFFF
                j += 1
            if n >= minn and not (n == 1 and (i == 0 or end == n_b)):
                out.append(h)
    return out


class FastTextTable:
    """A fastText model's dictionary on the host and its input matrix on the device."""

    def __init__(self, words, matrix, bucket, minn, maxn, device="cuda:0"):
        self.words = list(words)
        self.w2i = {w: i for i, w in enumerate(self.words)}
        self.nwords, self.bucket, self.minn, self.maxn = len(self.words), int(bucket), int(minn), int(maxn)
        m = torch.as_tensor(matrix)
        if m.shape[0] != self.nwords + self.bucket or m.dtype != torch.float32 or m.shape[1] % 4:
            raise ValueError("input matrix must be fp32 [nwords + bucket, dim] with dim a multiple of 4, got %s %s" % (tuple(m.shape), m.dtype))
        self.dim = int(m.shape[1])
        self.table = m.contiguous().to(device)          # resident in HBM from here on
        self._cache = {}

    @classmethod
    def load(cls, path, device="cuda:0"):
        """``FastText::loadModel`` for a non-quantized ``.bin`` (e.g. wiki.en.bin, processors.py:392-408)."""
        with open(path, "rb") as f:
            magic, version = struct.unpack("<ii", f.read(8))
            if magic != MAGIC or version != VERSION:
                raise ValueError("%s is not a fastText v12 .bin model" % path)
            dim, _ws, _ep, _mc, _neg, _wn, _loss, _model, bucket, minn, maxn, _lru = struct.unpack("<12i", f.read(48))
            f.read(8)
            size, nwords, _nlabels = struct.unpack("<3i", f.read(12))
            _ntokens, prune = struct.unpack("<qq", f.read(16))
            words = []
            for _ in range(size):
                raw = bytearray()
                while True:
                    c = f.read(1)
                    if c == b"\x00" or c == b"":
                        break
                    raw += c
                _count, typ = struct.unpack("<qb", f.read(9))
                if typ == 0:
                    words.append(bytes(raw).decode("utf-8"))
            f.read(8 * max(prune, 0))
            (quant,) = struct.unpack("<?", f.read(1))
            if quant:
                raise ValueError("quantized fastText models are not supported")
            m, n = struct.unpack("<qq", f.read(16))
            if m != nwords + bucket or n != dim:
                raise ValueError("input matrix is %d x %d, expected %d x %d" % (m, n, nwords + bucket, dim))
            mat = np.fromfile(f, dtype="<f4", count=m * n).reshape(m, n)
        return cls(words, torch.from_numpy(mat), bucket, minn, maxn, device)

    def subword_ids(self, word):
        """``Dictionary::getSubwords``: the word's own row (if in the vocabulary) followed by its n-gram rows."""
        ids = self._cache.get(word)
        if ids is None:
            wid = self.w2i.get(word, -1)
            if wid >= 0 and word == EOS:
                ids = [wid]
            else:
                grams = [self.nwords + h % self.bucket for h in _fnv1a_ngrams(("<" + word + ">").encode("utf-8"), self.minn, self.maxn)] \
                    if (self.maxn > 0 and word != EOS) else []
                ids = ([wid] if wid >= 0 else []) + grams
            if len(self._cache) < 1 << 20:
                self._cache[word] = ids
        return ids

    def encode(self, token_lists, max_length):
        """Batch of token lists -> CSR arrays for ``features``: one slot per (sample, position); a token is split at spaces into
        words (vocab.py:381), each word into subword rows.  Returns (ids int32 [nnz], word_end uint8 [nnz], offsets int32
        [B * max_length + 1]); slots past a sample's tokens are empty (zeros, PAD_INDEX, processors.py:482-486)."""
        ids, wend, off = [], [], [0]
        for tokens in token_lists:
            for i in range(max_length):
                if i < len(tokens):
                    for w in tokens[i].split(" "):
                        sub = self.subword_ids(w)
                        if sub:
                            ids.extend(sub)
                            wend.extend([0] * (len(sub) - 1) + [1])
                        else:          # a word without any row is a zero vector but still counts in the mean over words
                            ids.append(-1)
                            wend.append(1)
                off.append(len(ids))
        return (torch.tensor(ids, dtype=torch.int32), torch.tensor(wend, dtype=torch.uint8), torch.tensor(off, dtype=torch.int32))

    def features(self, token_lists, max_length, out=None):
        """[B, max_length, dim] fp32 on the device == ``FastTextProcessor`` applied per sample (processors.py:478-491)."""
        B = len(token_lists)
        ids, wend, off = self.encode(token_lists, max_length)
        dev = self.table.device
        ids, wend, off = ids.to(dev), wend.to(dev), off.to(dev)
        if out is None:
            out = torch.empty(B, max_length, self.dim, dtype=torch.float32, device=dev)
        assert out.shape == (B, max_length, self.dim) and out.dtype == torch.float32 and out.is_contiguous()
        if ids.numel() == 0:
            return out.zero_()
        X.check(X.lib().t2s_fasttext_rows(X.ptr(self.table), self.table.shape[0], self.dim, X.ptr(ids), X.ptr(wend), X.ptr(off),
                                          B * max_length, X.ptr(out), X.stream()), "t2s_fasttext_rows")
        return out
