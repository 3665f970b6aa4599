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
* ``EvalAIAnswerProcessor`` / ``TextVQAAccuracyEvaluator`` / ``STVQAAccuracyEvaluator`` :5-274 and the ``textvqa_accuracy``
  wrapper metrics.py:175-222: the VQA-challenge answer normalisation (lower-case, punctuation, number words, articles,
  contractions) and the leave-one-out soft accuracy over 10 human answers.  The contraction table is generated from its
  rule (see ``_contractions``) instead of being listed; ``tests/golden/evalai.json`` pins table and outputs.
"""
import re

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


# ---- VQA-challenge ("EvalAI") answer normalisation and accuracies --------------------------------------------------------
def _contractions():
    """Misspelt -> apostrophised contraction (m4c_evaluators.py:12-133).  The table follows one rule: a form with ONE
    apostrophe is looked up without it ("dont" -> "don't"); a form with several is looked up with exactly one of them
    missing ("couldnt've", "couldn'tve" -> "couldn't've").  Three entries of the original table break the rule and are
    kept as they are (identity for let's / she's, and the inverted somebody'd -> somebodyd)."""
    single = ("'twas I'm I've ain't aren't can't could've couldn't didn't doesn't don't hadn't hasn't haven't he'd he's how'd "
              "how'll how's isn't it'd it'll ma'am might've mightn't must've mustn't needn't not've o'clock oughtn't shan't "
              "should've shouldn't somebody'll somebody's someone'd someone'll someone's something'd something'll that's "
              "there'd there're there's they'd they'll they're they've wasn't we've weren't what'll what're what's what've "
              "when's where'd where's where've who'd who'll who's who've why'll why're why's won't would've wouldn't y'all "
              "you'd you'll you're you've").split()
    several = ("couldn't've hadn't've he'd've I'd've it'd've mightn't've she'd've shouldn't've somebody'd've someone'd've "
               "something'd've there'd've they'd've we'd've who'd've wouldn't've you'd've y'all'll y'all'd've 'ow's'at").split()
    table = {w.replace("'", ""): w for w in single}
    for w in several:
        for m in re.finditer("'", w):
            table[w[:m.start()] + w[m.end():]] = w
    table.update({"let's": "let's", "she's": "she's", "somebody'd": "somebodyd"})
    return table


class EvalAIAnswerProcessor:
    """``processor(answer) -> normalised answer``; quirks of the original kept because they change scores: the period
    rule strips at most 32 periods (``re.UNICODE`` passed in the ``count`` slot, :192), the punctuation tests look at the
    INPUT string while replacing in the output (:186-191), and the capitalised contraction keys can never match the
    lower-cased words."""
    CONTRACTIONS = _contractions()
    NUMBER_MAP = dict(zip("none zero one two three four five six seven eight nine ten".split(), "0 0 1 2 3 4 5 6 7 8 9 10".split()))
    ARTICLES = ("a", "an", "the")
    PUNCTUATIONS = tuple(';/[]"{}()=+\\_-><@`,?!')
    _period = re.compile(r"\.(?!\d)")                 # ":135" is "(?!<=\d)(\.)(?!\d)": its first group can never fail
    _digit_comma = re.compile(r"(?<=\d)(\,)+(?=\d)")

    def word_tokenize(self, word):
        word = word.lower().replace(",", "").replace("?", "").replace("'s", " 's")
        return word.strip()

    def process_punctuation(self, text):
        out = text
        number_comma = self._digit_comma.search(text) is not None
        for p in self.PUNCTUATIONS:
            glued = (p + " " in text) or (" " + p in text) or number_comma
            out = out.replace(p, "" if glued else " ")
        return self._period.sub("", out, 32)

    def process_digit_article(self, text):
        words = [self.NUMBER_MAP.get(w, w) for w in text.lower().split()]
        return " ".join(self.CONTRACTIONS.get(w, w) for w in words if w not in self.ARTICLES)

    def __call__(self, item):
        item = self.word_tokenize(item).replace("\n", " ").replace("\t", " ").strip()
        return self.process_digit_article(self.process_punctuation(item))


class TextVQAAccuracyEvaluator:
    """Soft accuracy against 10 human answers: a predicted answer scores the mean over the 10 leave-one-out subsets of
    min(1, matches / 3) (m4c_evaluators.py:223-259)."""

    def __init__(self):
        self.answer_processor = EvalAIAnswerProcessor()

    def _compute_answer_scores(self, raw_answers):
        answers = [self.answer_processor(a) for a in raw_answers]
        assert len(answers) == 10
        scores = {}
        for ua in set(answers):
            n = answers.count(ua)
            # leaving out one of the n matching answers leaves n - 1 matches, leaving out another leaves n
            accs = [min(1, float(n - (a == ua)) / 3) for a in answers]
            scores[ua] = sum(accs) / len(accs)
        return scores

    def eval_pred_list(self, pred_scores, pred_list):
        for entry in pred_list:
            pred = self.answer_processor(entry["pred_answer"])
            pred_scores.append(self._compute_answer_scores(entry["gt_answers"]).get(pred, 0.))
        return pred_scores, sum(pred_scores) / len(pred_scores)


class STVQAAccuracyEvaluator:
    """1 if the normalised prediction equals any normalised ground truth (m4c_evaluators.py:262-274)."""

    def __init__(self):
        self.answer_processor = EvalAIAnswerProcessor()

    def eval_pred_list(self, pred_scores, pred_list):
        for entry in pred_list:
            pred = self.answer_processor(entry["pred_answer"])
            pred_scores.append(1. if pred in [self.answer_processor(a) for a in entry["gt_answers"]] else 0.)
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


class TextVQAAccuracy:
    """``textvqa_accuracy`` (metrics.py:175-222): decode ``pos_scores`` and score against the 10 human answers."""
    name = "textvqa_accuracy"

    def __init__(self, answer_vocab, eos_idx, word_tokenize=lambda w: w):
        self.vocab, self.eos, self.tok = answer_vocab, eos_idx, word_tokenize
        self.evaluator = TextVQAAccuracyEvaluator()

    def calculate(self, sample_list, model_output):
        pred = model_output["pos_scores"].argmax(dim=-1)
        answers = decode_answers(pred, sample_list["context_tokens"], self.vocab, len(self.vocab), self.eos, self.tok)
        entries = [{"pred_answer": a, "gt_answers": g} for a, g in zip(answers, sample_list["gt_answers"])]
        _, acc = self.evaluator.eval_pred_list([], entries)
        return torch.tensor(acc, device=model_output["pos_scores"].device)


class STVQAANLS(TextVQAAccuracy):
    """``stvqa_anls`` (metrics.py:224-231): the same decoding, scored by ANLS."""
    name = "stvqa_anls"

    def __init__(self, answer_vocab, eos_idx, word_tokenize=lambda w: w):
        super().__init__(answer_vocab, eos_idx, word_tokenize)
        self.evaluator = STVQAANLSEvaluator()


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


# ---- the ``Metrics`` container that ``BaseModel.__call__`` runs (pythia/modules/metrics.py:56-134, base_model.py:78-149) -----
def dec_bytes2obj(byte_tensor):
    """``pythia/utils/objects_to_byte_tensor.py:33-43``: a pickled object in a uint8 tensor, 2-byte big-endian length first
    (how ``context_tokens_enc`` / ``gt_answers_enc`` travel through the collate function)."""
    import pickle
    b = byte_tensor.tolist() if torch.is_tensor(byte_tensor) else list(byte_tensor)
    n = int(b[0]) * 256 + int(b[1])
    return pickle.loads(bytes(int(x) for x in b[2:2 + n]))


def enc_obj2bytes(obj, max_size=16384):
    """Inverse of ``dec_bytes2obj`` (objects_to_byte_tensor.py:11-30)."""
    import pickle
    enc = pickle.dumps(obj)
    if len(enc) > max_size:
        raise ValueError("objects too large: object size %d, max size %d" % (len(enc), max_size))
    t = torch.zeros(max_size, dtype=torch.uint8)
    t[0], t[1] = len(enc) // 256, len(enc) % 256
    t[2:2 + len(enc)] = torch.tensor(list(enc), dtype=torch.uint8)
    return t


def _batch_objects(sample_list, enc_key, plain_key):
    """Per-sample python objects of a batch: the pickled byte tensors the reference's dataset emits (``*_enc``), or plain lists."""
    if enc_key in sample_list:
        enc = sample_list[enc_key]
        enc = enc.cpu() if torch.is_tensor(enc) else enc
        return [dec_bytes2obj(row) for row in enc]
    return sample_list[plain_key]


class _RegistryAnswerMetric:
    """``textvqa_accuracy`` / ``stvqa_anls`` as the reference registers them: no constructor arguments, the answer processor
    (vocabulary, EOS index, true vocabulary size) comes from ``registry.get(dataset_name + "_answer_processor")`` at call time
    (metrics.py:175-231)."""
    evaluator_cls = None

    def __init__(self):
        self.evaluator = self.evaluator_cls()

    def calculate(self, sample_list, model_output, *args, **kwargs):
        from .registry import registry
        ap = registry.get(sample_list["dataset_name"] + "_answer_processor")
        vocab_size = ap.get_true_vocab_size()
        pred = model_output["pos_scores"].argmax(dim=-1)
        tokens = _batch_objects(sample_list, "context_tokens_enc", "context_tokens")
        gts = _batch_objects(sample_list, "gt_answers_enc", "gt_answers")
        tok = getattr(ap, "word_tokenize", None) or (lambda w: w)
        answers = decode_answers(pred, tokens, _Idx2Word(ap.answer_vocab), vocab_size, ap.EOS_IDX, tok)
        entries = [{"pred_answer": a, "gt_answers": g} for a, g in zip(answers, gts)]
        _, acc = self.evaluator.eval_pred_list([], entries)
        return torch.tensor(acc, device=model_output["pos_scores"].device)

    _calculate_with_checks = calculate
    __call__ = calculate


class _Idx2Word:
    def __init__(self, vocab):
        self.v = vocab

    def __getitem__(self, i):
        return self.v.idx2word(i) if hasattr(self.v, "idx2word") else self.v[i]


def _register_default_metrics():
    from .registry import registry

    @registry.register_metric("textvqa_accuracy")
    class RegisteredTextVQAAccuracy(_RegistryAnswerMetric):
        name = "textvqa_accuracy"
        evaluator_cls = TextVQAAccuracyEvaluator

    @registry.register_metric("stvqa_anls")
    class RegisteredSTVQAANLS(_RegistryAnswerMetric):
        name = "stvqa_anls"
        evaluator_cls = STVQAANLSEvaluator

    for thr in (0.3, 0.5):
        def make(thr=thr):
            class RegisteredIOU:
                """``IOU@t`` (metrics.py:233-339): the grounding annotation is taken from ``registry.get("ground_annotation")``
                (a list of dicts; the reference reads a hard-coded .npy path, metrics.py:250-254)."""
                name = "IOU@%s" % thr

                def calculate(self, sample_list, model_output, *args, **kwargs):
                    info = registry.get("ground_annotation")
                    if info is None:
                        raise RuntimeError("metric IOU@%s needs registry key 'ground_annotation' (list of dicts with question_id, "
                                           "spatial_temporal_gt, fps, width, height)" % thr)
                    return BoxGroundAccuracy(info, thr).calculate(sample_list, model_output)

                _calculate_with_checks = calculate
            return RegisteredIOU
        registry.register_metric("IOU@%s" % thr)(make())


class Metrics:
    """``pythia/modules/metrics.py:56-134``: built from the yml ``metrics`` list (names or {type, params} dicts); called by
    ``BaseModel.__call__`` with (sample_list, model_output); returns {"<dataset_type>/<dataset_name>/<metric>": 1-element float
    tensor}.  Kept quirks: nothing is computed when the batch has no ``targets``; after the first TRAIN batch only
    textvqa_accuracy / stvqa_anls survive (Appendix A, Q16: the reference reassigns ``self.metrics`` permanently); the values
    are also published as ``registry["metrics.<dataset_name>.<dataset_type>"]``."""

    def __init__(self, metric_list):
        from .registry import registry
        _register_default_metrics()
        if not isinstance(metric_list, (list, tuple)):
            metric_list = [metric_list]
        self.metrics = {}
        for metric in metric_list:
            params = {}
            if isinstance(metric, dict) or hasattr(metric, "keys"):
                if "type" not in metric:
                    raise ValueError("Metric {} needs to have 'type' attribute".format(metric))
                params = dict(metric.get("params", {}))
                metric = metric["type"]
            elif not isinstance(metric, str):
                raise TypeError("Metric {} has inappropriate type 'dict' or 'str' allowed".format(metric))
            cls = registry.get_metric_class(metric)
            if cls is None:
                raise ValueError("No metric named {} registered to registry".format(metric))
            self.metrics[metric] = cls(**params)

    def __call__(self, sample_list, model_output, *args, **kwargs):
        from .registry import registry
        values = {}
        if "targets" not in sample_list:
            return values
        dataset_type, dataset_name = sample_list["dataset_type"], sample_list["dataset_name"]
        with torch.no_grad():
            if dataset_type == "train":
                self.metrics = {k: v for k, v in self.metrics.items() if k in {"textvqa_accuracy", "stvqa_anls"}}
            for name, obj in self.metrics.items():
                key = "{}/{}/{}".format(dataset_type, dataset_name, name)
                v = obj._calculate_with_checks(sample_list, model_output, *args, **kwargs)
                v = v.float() if torch.is_tensor(v) else torch.tensor(v, dtype=torch.float)
                values[key] = v.view(1) if v.dim() == 0 else v
        registry.register("{}.{}.{}".format("metrics", dataset_name, dataset_type), values)
        return values
