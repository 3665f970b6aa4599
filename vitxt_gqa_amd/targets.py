"""Teacher-forcing targets for the iterative answer decoder: the producer of the boundary inputs ``targets``
(``answers_scores``), ``train_prev_inds`` and ``train_loss_mask``.  Host-side Python, as in the
reference.

Mirrors ``M4CAnswerProcessor`` (``pythia/datasets/processors.py:987-1156``), the answer vocabulary ``VocabDict``
(``pythia/utils/text_utils.py:88-152``) and the ``simple_word`` preprocessor (``text_utils.py:71-78``,
``processors.py:785-799``).  Index space: fixed vocabulary first, then the sample's OCR tokens (``vocab_size + i``).
"""
from collections import defaultdict

import numpy as np
import torch


def word_tokenize(word, remove=(",", "?")):
    """Lower-case, drop commas / question marks, split a possessive 's off (text_utils.py:71-78)."""
    word = word.lower()
    for ch in remove:
        word = word.replace(ch, "")
    return word.replace("'s", " 's").strip()


class AnswerVocab:
    """One word per line; ``<unk>`` is prepended when the file lacks it (text_utils.py:109-127)."""
    UNK, PAD = "<unk>", "<pad>"

    def __init__(self, words):
        words = [w.strip() for w in words]
        if self.UNK not in words:
            words = [self.UNK] + words
        self.words = words
        self.index = {w: i for i, w in enumerate(words)}
        self.unk_index = self.index[self.UNK]

    @classmethod
    def from_file(cls, path):
        with open(path) as f:
            return cls(f.readlines())

    def __len__(self):
        return len(self.words)

    def __getitem__(self, i):
        return self.words[i]

    def word2idx(self, w):
        return self.index.get(w, self.unk_index)


def match_sequences(answer, vocab_index, ocr_positions, vocab_size, max_match_num=20):
    """All ways to spell ``answer`` (already preprocessed) word by word from the fixed vocabulary and / or OCR tokens, as
    tuples of indices, in the reference's order and truncated to ``max_match_num`` after every word
    (processors.py:1013-1053).  A word that matches nothing makes the whole answer unspellable."""
    seqs = [()]
    words = answer.split()
    if not words:
        return []
    for w in words:
        hits = ([vocab_index[w]] if w in vocab_index else []) + [vocab_size + i for i in ocr_positions[w]]
        if not hits:
            return []
        seqs = [s + (h,) for s in seqs for h in hits][:max_match_num]
    return seqs


class AnswerTargetBuilder:
    def __init__(self, vocab, max_length, max_copy_steps=12, num_answers=10, preprocess=word_tokenize):
        self.vocab = vocab
        self.PAD_IDX, self.BOS_IDX, self.EOS_IDX = vocab.word2idx("<pad>"), vocab.word2idx("<s>"), vocab.word2idx("</s>")
        if self.PAD_IDX != 0 or vocab.unk_index in (self.PAD_IDX, self.BOS_IDX, self.EOS_IDX):
            raise ValueError("the answer vocabulary must hold <pad> (index 0), <s> and </s>")
        self.max_length, self.max_copy_steps, self.num_answers = max_length, max_copy_steps, num_answers
        self.preprocess = preprocess

    def get_vocab_size(self):
        return len(self.vocab) + self.max_length

    def get_true_vocab_size(self):
        return len(self.vocab)

    def __call__(self, item, rng=np.random):
        """item: {"answers": [str] * num_answers, "context_tokens": [str]} -> the reference's ``answer_info`` dict.  The one
        random draw (which spelling to teach, processors.py:1129) comes from ``rng.choice`` exactly as in the reference,
        so the same numpy seed gives the same sample."""
        answers = [self.preprocess(a) for a in item["answers"]]
        assert len(answers) == self.num_answers
        # soft score of every distinct answer: mean over the annotators of min(1, #other annotators agreeing / 3)
        score_of = {}
        for a in set(answers):
            accs = []
            for i in range(len(answers)):
                agree = sum(1 for j, other in enumerate(answers) if j != i and other == a)
                accs.append(min(1, agree / 3))
            score_of[a] = sum(accs) / len(accs)
        V = len(self.vocab)
        scores = torch.zeros(self.max_copy_steps, self.get_vocab_size(), dtype=torch.float)
        ocr_positions = defaultdict(list)
        for i, tok in enumerate(item["context_tokens"]):
            ocr_positions[tok].append(i)
        all_seqs = []
        for a in answers:
            seqs = match_sequences(a, self.vocab.index, ocr_positions, V)
            all_seqs.extend(seqs)
            for s in seqs:                         # step 0: the best score among the answers starting with that token
                scores[0, s[0]] = max(scores[0, s[0]], score_of[a])
        prev = torch.zeros(self.max_copy_steps, dtype=torch.long)
        mask = torch.zeros(self.max_copy_steps, dtype=torch.float)
        seq = ()
        if all_seqs:
            seq = all_seqs[rng.choice(len(all_seqs))]
            steps = min(1 + len(seq), self.max_copy_steps)
            mask[:steps] = 1.
            prev[0] = self.BOS_IDX
            for t in range(1, steps):
                prev[t] = seq[t - 1]
                scores[t, seq[t] if t < len(seq) else self.EOS_IDX] = 1.
        return {"answers": answers, "answers_scores": scores, "sampled_idx_seq": seq, "train_prev_inds": prev, "train_loss_mask": mask}
