"""Validation metrics that consume the T2S outputs (second half): answer ANLS and the
temporal / spatial grounding accuracies.  Host-side Python, as in the reference.

Mirrors, with the reference's quirks kept (they change the numbers):
* ``STVQAANLSEvaluator``        pythia/utils/m4c_evaluators.py:277-298  (edit distance: own Levenshtein instead of the
                                 ``editdistance`` package, which is not installed here)
* ``TempGroundAccuracyEvaluator`` :301-326
* ``BoxGroundAccuracyEvaluator``  :329-405  (pixel-inclusive ``+1`` IoU; ``check_iou`` appends a 1 for EVERY matching
                                 (gt span, predicted frame) pair, a 0 only if the LAST checked pair failed, so the
                                 denominator is the number of appended scores, not of questions)
* metric wrappers ``stvqa_anls`` / ``IOU@t`` pythia/modules/metrics.py:224-339: the decoding of ``pos_scores`` into answer
  strings (argmax, OCR-copy indices >= vocabulary size, stop at EOS) and the per-question grounding entries.  The
  reference reads the grounding annotation from a hard-coded .npy path; here it is handed to the constructor.
``textvqa_accuracy`` (EvalAI answer normalisation tables, m4c_evaluators.py:5-259) is not rebuilt.
"""
import torch


def levenshtein(a, b):
    """Edit distance with unit costs (what ``editdistance.eval`` returns for two strings)."""
    if len(a) < len(b):
        a, b = b, a
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


def anls(pred, gt):
    """1 - normalised edit distance, zeroed below 0.5 (m4c_evaluators.py:282-287).  Both empty -> ZeroDivisionError, as
    in the reference."""
    s1, s2 = pred.lower().strip(), gt.lower().strip()
    sim = 1 - levenshtein(s1, s2) / max(len(s1), len(s2))
    return sim if sim >= .5 else 0.


class STVQAANLSEvaluator:
    def eval_pred_list(self, pred_scores, pred_list):
        for entry in pred_list:
            pred_scores.append(max(anls(entry["pred_answer"], gt) for gt in entry["gt_answers"]))
        return pred_scores, sum(pred_scores) / len(pred_scores)


def _frame_span(t_span, fps):
    lo, hi = t_span["temporal_gt"]
    return int(lo * fps) + 1, int(hi * fps) + 1


class TempGroundAccuracyEvaluator:
    def eval_pred_list(self, pred_list):
        hits = []
        for entry in pred_list:
            hit = 0
            for t_span in entry["st_gt"]:
                st, ed = _frame_span(t_span, entry["video_fps"])
                if any(st <= f <= ed for f in entry["pred_frame"]):
                    hit = 1
                    break
            hits.append(hit)
        return sum(hits) / len(hits)


def box_iou(b1, b2):
    """IoU of two (x1, y1, x2, y2) pixel boxes with inclusive corners (the ``+ 1`` of m4c_evaluators.py:347-351)."""
    iw = max(0, min(b1[2], b2[2]) - max(b1[0], b2[0]) + 1)
    ih = max(0, min(b1[3], b2[3]) - max(b1[1], b2[1]) + 1)
    inter = iw * ih
    a1 = (b1[2] - b1[0] + 1) * (b1[3] - b1[1] + 1)
    a2 = (b2[2] - b2[0] + 1) * (b2[3] - b2[1] + 1)
    return inter / (a1 + a2 - inter)


class BoxGroundAccuracyEvaluator:
    def check_iou(self, pred_scores, bbox_iou_value, gt_bbox, pred_bboxs, threshold=None):
        best, last = 0, -1
        for pb in pred_bboxs:
            assert pb[0] <= pb[2] and pb[1] <= pb[3]
            last = box_iou(gt_bbox, pb)
            best = max(best, last)
        ok = best > threshold
        bbox_iou_value.append(last)                      # the reference records the LAST IoU, not the best
        if ok:
            pred_scores.append(1)
        return pred_scores, bbox_iou_value, ok

    def eval_pred_list(self, pred_scores, pred_list, threshold=None):
        for entry in pred_list:
            w, h = entry["width"], entry["height"]
            boxes = [[b[0] * w, b[1] * h, b[2] * w, b[3] * h] for b in entry["pred_box"]]
            k = entry["ocr_topk"]
            ious, ok = [], False
            for t_span in entry["st_gt"]:
                st, ed = _frame_span(t_span, entry["video_fps"])
                gts = t_span["bbox_gt"]
                for i, f in enumerate(entry["pred_frame"]):
                    if st <= int(f) <= ed and str(int(f - 1)) in gts:
                        gt = gts[str(int(f - 1))]
                        assert gt[0] <= gt[2] and gt[1] <= gt[3]
                        pred_scores, ious, ok = self.check_iou(pred_scores, ious, gt, boxes[i * k:(i + 1) * k], threshold)
            if not ok:
                pred_scores.append(0)
        return pred_scores, sum(pred_scores) / len(pred_scores)


# ---- metric wrappers over (sample_list, model_output) ----------------------------------------------------------------
def decode_answers(pred_inds, context_tokens, answer_vocab, vocab_size, eos_idx, word_tokenize=lambda w: w):
    """Indices [B, T] -> answer strings: ids >= vocab_size copy OCR token ``id - vocab_size``; decoding stops at EOS;
    words joined by spaces with " 's" glued back (metrics.py:196-213)."""
    out = []
    for b, row in enumerate(pred_inds.tolist()):
        words = []
        for idx in row:
            if idx >= vocab_size:
                words.append(word_tokenize(context_tokens[b][idx - vocab_size]))
            elif idx == eos_idx:
                break
            else:
                words.append(answer_vocab[idx])
        out.append(" ".join(words).replace(" 's", "'s"))
    return out


class STVQAANLS:
    name = "stvqa_anls"

    def __init__(self, answer_vocab, eos_idx, word_tokenize=lambda w: w):
        self.vocab, self.eos, self.tok = answer_vocab, eos_idx, word_tokenize
        self.evaluator = STVQAANLSEvaluator()

    def calculate(self, sample_list, model_output):
        pred = model_output["pos_scores"].argmax(dim=-1)
        answers = decode_answers(pred, sample_list["context_tokens"], self.vocab, len(self.vocab), self.eos, self.tok)
        entries = [{"pred_answer": a, "gt_answers": g} for a, g in zip(answers, sample_list["gt_answers"])]
        _, acc = self.evaluator.eval_pred_list([], entries)
        return torch.tensor(acc, device=model_output["pos_scores"].device)


class BoxGroundAccuracy:
    """``IOU@0.3`` / ``IOU@0.5`` (metrics.py:233-339).  ``ground_info``: list of dicts with question_id, spatial_temporal_gt,
    fps, width, height (the content of the reference's grounding .npy)."""

    def __init__(self, ground_info, threshold):
        self.name = "IOU@%s" % threshold
        self.threshold = threshold
        self.by_id = {g["question_id"]: g for g in ground_info if "question_id" in g}
        self.evaluator = BoxGroundAccuracyEvaluator()

    def entries(self, sample_list, model_output):
        frames = model_output["ground_frame"].detach().cpu().numpy().tolist()
        boxes = model_output["ground_box"].detach().cpu().tolist()
        ft, ot = int(model_output["frame_topk"]), int(model_output["ocr_topk"])
        out = []
        for i, qid in enumerate(sample_list["question_id"]):
            g = self.by_id[qid]
            out.append({"pred_frame": frames[i], "pred_box": boxes[i], "frame_topk": ft, "ocr_topk": ot,
                        "st_gt": g["spatial_temporal_gt"], "video_fps": g["fps"], "width": g["width"], "height": g["height"]})
        return out

    def calculate(self, sample_list, model_output):
        _, acc = self.evaluator.eval_pred_list([], self.entries(sample_list, model_output), threshold=self.threshold)
        return torch.tensor(acc, device=model_output["ground_frame"].device)
