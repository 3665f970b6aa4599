"""Generates tests/golden/targets.json from the REFERENCE's M4CAnswerProcessor (pythia/datasets/processors.py:987-1156)
on synthetic questions.  Runs only in the authoring container; third-party modules the import chain wants but the
processor never touches are stubbed (the same way tests/golden/make_golden.py shims pytorch_transformers)."""
import json
import os
import random
import sys
import tempfile
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, "/root/reference")
import transformers  # noqa: E402

pt, tb = types.ModuleType("pytorch_transformers"), types.ModuleType("pytorch_transformers.tokenization_bert")
tb.BertTokenizer = transformers.BertTokenizer
sys.modules["pytorch_transformers"], sys.modules["pytorch_transformers.tokenization_bert"] = pt, tb
for m in ("editdistance", "demjson", "lmdb", "cv2", "torchtext", "torchtext.vocab", "git", "tensorboardX", "nltk", "fasttext", "fastText"):
    if m not in sys.modules:
        try:
            __import__(m)
        except Exception:
            sys.modules[m] = types.ModuleType(m)
from pythia.datasets.processors import M4CAnswerProcessor  # noqa: E402
from pythia.utils.configuration import ConfigNode  # noqa: E402

words = ["<pad>", "<s>", "</s>", "<unk>", "yes", "no", "red", "blue", "stop", "bus", "the", "coca", "cola", "'s", "joe", "diner", "main", "street", "24", "7"]
rnd = random.Random(5)
with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as f:
    f.write("\n".join(words) + "\n")
    vocab_path = f.name
MAXLEN, STEPS, NANS = 30, 12, 10
proc = M4CAnswerProcessor(ConfigNode({"vocab_file": vocab_path, "preprocessor": {"type": "simple_word", "params": {}},
                                      "context_preprocessor": {"type": "simple_word", "params": {}},
                                      "max_length": MAXLEN, "max_copy_steps": STEPS, "num_answers": NANS}))
ocr_pool = ["pepsi", "cola", "coca", "joe", "diner", "exit", "main", "st.", "24", "7", "open", "zero", "stop", "x1", "the"]
ans_pool = ["Coca Cola", "coca cola", "Joe's Diner", "stop", "Stop?", "pepsi, zero", "yes", "main street", "24 7", "unseen words here",
            "the the the the the the the the the the the the the the", "exit", "open 24 7", "blue bus"]
cases = []
for i in range(40):
    ctx = [rnd.choice(ocr_pool) for _ in range(rnd.randint(0, 25))]
    base = rnd.choice(ans_pool)
    answers = [base if rnd.random() < 0.6 else rnd.choice(ans_pool) for _ in range(NANS)]
    np.random.seed(1000 + i)
    out = proc({"answers": list(answers), "context_tokens": list(ctx)})
    nz = out["answers_scores"].nonzero().tolist()
    cases.append({"answers": answers, "context_tokens": ctx, "seed": 1000 + i, "proc_answers": out["answers"],
                  "scores_nz": [[r, c, float(out["answers_scores"][r, c])] for r, c in nz],
                  "sampled_idx_seq": list(out["sampled_idx_seq"]), "train_prev_inds": out["train_prev_inds"].tolist(),
                  "train_loss_mask": out["train_loss_mask"].tolist()})
json.dump({"vocab": words, "max_length": MAXLEN, "max_copy_steps": STEPS, "num_answers": NANS, "bos": proc.BOS_IDX, "eos": proc.EOS_IDX,
           "vocab_size": proc.get_vocab_size(), "true_vocab_size": proc.get_true_vocab_size(), "cases": cases},
          open(os.path.join(ROOT, "tests", "golden", "targets.json"), "w"))
print(len(cases), "cases;", sum(1 for c in cases if c["sampled_idx_seq"]), "spellable; vocab", proc.get_true_vocab_size())
os.unlink(vocab_path)
