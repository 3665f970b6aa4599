"""Generates tests/golden/metrics.json from the REFERENCE's evaluators (pythia/utils/m4c_evaluators.py) on synthetic
prediction lists.  Runs only in the authoring container.  The reference's ANLS evaluator needs the third-party
``editdistance`` package (absent): a plain Wagner-Fischer table stands in for ``editdistance.eval`` here."""
import json
import os
import random
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _edit(a, b):
    d = [[i + j if i * j == 0 else 0 for j in range(len(b) + 1)] for i in range(len(a) + 1)]
    for i in range(1, len(a) + 1):
        for j in range(1, len(b) + 1):
            d[i][j] = min(d[i - 1][j] + 1, d[i][j - 1] + 1, d[i - 1][j - 1] + (a[i - 1] != b[j - 1]))
    return d[len(a)][len(b)]


ed = types.ModuleType("editdistance")
ed.eval = _edit
sys.modules["editdistance"] = ed
sys.path.insert(0, "/root/reference")
from pythia.utils import m4c_evaluators as R  # noqa: E402

rnd = random.Random(77)
words = "stop exit open closed coca cola pepsi 24/7 the of sale 50% off main street st. no parking bus taxi hotel 2019 a1".split()


def phrase():
    return " ".join(rnd.choice(words) for _ in range(rnd.randint(1, 3)))


def typo(s):
    s = list(s)
    for _ in range(rnd.randint(0, 3)):
        if s and rnd.random() < 0.7:
            s[rnd.randrange(len(s))] = rnd.choice("abcdefgh ")
        else:
            s.insert(rnd.randrange(len(s) + 1), rnd.choice("xyz"))
    return "".join(s)


anls_list = []
for _ in range(60):
    gts = [phrase() for _ in range(rnd.randint(1, 4))]
    pred = typo(rnd.choice(gts)) if rnd.random() < 0.7 else phrase()
    anls_list.append({"pred_answer": rnd.choice(["", " "]) + pred.upper() if rnd.random() < 0.2 else pred, "gt_answers": gts})
anls_scores, anls_acc = R.STVQAANLSEvaluator().eval_pred_list([], [dict(e) for e in anls_list])


def ground_entry():
    fps = rnd.choice([24, 25, 30, 29.97])
    w, h = rnd.choice([(1280, 720), (640, 360), (1920, 1080)])
    spans = []
    for _ in range(rnd.randint(1, 3)):
        t0 = rnd.uniform(0, 8)
        t1 = t0 + rnd.uniform(0.2, 3)
        st, edf = int(t0 * fps) + 1, int(t1 * fps) + 1
        bb = {}
        for f in range(st - 1, edf):
            if rnd.random() < 0.8:
                x1, y1 = rnd.randint(0, w - 50), rnd.randint(0, h - 30)
                bb[str(f)] = [x1, y1, x1 + rnd.randint(5, 200), y1 + rnd.randint(5, 80)]
        spans.append({"temporal_gt": [t0, t1], "bbox_gt": bb})
    topk, otk = 5, 5
    frames = sorted(rnd.sample(range(1, int(12 * fps)), topk))
    if rnd.random() < 0.7:                      # make some predictions land inside a span, near a gt box
        sp = rnd.choice(spans)
        st, edf = int(sp["temporal_gt"][0] * fps) + 1, int(sp["temporal_gt"][1] * fps) + 1
        frames[rnd.randrange(topk)] = rnd.randint(st, edf)
        frames = sorted(frames)
    boxes = []
    for f in frames:
        for _ in range(otk):
            gt = None
            for sp in spans:
                gt = sp["bbox_gt"].get(str(f - 1), gt)
            if gt is not None and rnd.random() < 0.5:
                j = lambda v, s: min(max(v + rnd.randint(-s, s), 0), 10 ** 6)
                x1, y1, x2, y2 = j(gt[0], 20), j(gt[1], 10), j(gt[2], 20), j(gt[3], 10)
                x1, x2, y1, y2 = min(x1, x2), max(x1, x2), min(y1, y2), max(y1, y2)
            else:
                x1, y1 = rnd.randint(0, w - 50), rnd.randint(0, h - 30)
                x2, y2 = x1 + rnd.randint(5, 200), y1 + rnd.randint(5, 80)
            boxes.append([x1 / w, y1 / h, x2 / w, y2 / h])
    return {"pred_frame": frames, "pred_box": boxes, "frame_topk": topk, "ocr_topk": otk, "st_gt": spans, "video_fps": fps, "width": w, "height": h}


ground = [ground_entry() for _ in range(50)]
out = {"anls": {"entries": anls_list, "scores": anls_scores, "accuracy": anls_acc},
       "ground": {"entries": ground, "temporal_accuracy": R.TempGroundAccuracyEvaluator().eval_pred_list([dict(e) for e in ground])}}
for thr in (0.3, 0.5):
    sc, acc = R.BoxGroundAccuracyEvaluator().eval_pred_list([], [dict(e) for e in ground], threshold=thr)
    out["ground"]["iou@%s" % thr] = {"scores": sc, "accuracy": acc}
ev = R.BoxGroundAccuracyEvaluator()
pairs = [([rnd.randint(0, 100), rnd.randint(0, 100), rnd.randint(100, 300), rnd.randint(100, 300)],
          [rnd.randint(0, 200), rnd.randint(0, 200), rnd.randint(200, 400), rnd.randint(200, 400)]) for _ in range(40)]
out["iou_pairs"] = [{"a": a, "b": b, "iou": ev.calculate_iou(a, b)} for a, b in pairs]
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "metrics.json"), "w"))
print("anls acc %.4f, temporal %.4f, iou@0.3 %.4f (%d scores), iou@0.5 %.4f" % (anls_acc, out["ground"]["temporal_accuracy"],
      out["ground"]["iou@0.3"]["accuracy"], len(out["ground"]["iou@0.3"]["scores"]), out["ground"]["iou@0.5"]["accuracy"]))
