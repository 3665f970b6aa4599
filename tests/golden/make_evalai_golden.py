"""Generates tests/golden/evalai.json from the REFERENCE's EvalAI answer normalisation and accuracy evaluators
(pythia/utils/m4c_evaluators.py:5-274).  Runs only in the authoring container (needs /root/reference)."""
import json
import os
import random
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.modules.setdefault("editdistance", types.ModuleType("editdistance"))
sys.path.insert(0, "/root/reference")
from pythia.utils import m4c_evaluators as R  # noqa: E402

rnd = random.Random(2024)
proc = R.EvalAIAnswerProcessor()
table = dict(proc.CONTRACTIONS)                       # before any call (the reference grows NUMBER_MAP while it runs)
words = ("stop Exit the a an one two ten none zero dont Dont couldnt've she's let's somebody'd yall'd've whats 24/7 1,000 3.5 "
         "st. u.s.a. mr. coca-cola (sale) [50%] off? yes! no, maybe; \"quoted\" joe's its' it's o'clock oclock a.m. 7 8,5 "
         "e=mc2 a_b x>y <tag> user@mail `code` back\\slash tab\there new\nline ... . , ? ! thered've Ive im").split(" ")


def phrase():
    return " ".join(rnd.choice(words) for _ in range(rnd.randint(1, 4)))


strings = sorted(set([phrase() for _ in range(300)] + words + ["", " ", "." * 40, "1,000,000 dollars.", "a the an", "The U.S.A."]))
processed = {s: proc(s) for s in strings}

tv_entries = []
for _ in range(80):
    pool = [phrase() for _ in range(rnd.randint(1, 4))]
    gts = [rnd.choice(pool) for _ in range(10)]
    pred = rnd.choice(pool + [phrase()])
    if rnd.random() < 0.3:
        pred = pred.upper()
    tv_entries.append({"pred_answer": pred, "gt_answers": gts})
tv_scores, tv_acc = R.TextVQAAccuracyEvaluator().eval_pred_list([], [dict(e) for e in tv_entries])
st_scores, st_acc = R.STVQAAccuracyEvaluator().eval_pred_list([], [dict(e) for e in tv_entries])
json.dump({"contractions": table, "processed": processed,
           "entries": tv_entries, "textvqa": {"scores": tv_scores, "accuracy": tv_acc}, "stvqa": {"scores": st_scores, "accuracy": st_acc}},
          open(os.path.join(ROOT, "tests", "golden", "evalai.json"), "w"))
print("%d contractions, %d strings, textvqa acc %.4f, stvqa acc %.4f" % (len(table), len(processed), tv_acc, st_acc))
