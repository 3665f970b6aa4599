"""Generates tests/golden/dataset_samples.npz from the REFERENCE's sample builder
(VTEXTGQADataset.add_sample_details / add_answer_info / sample_frames, pythia/datasets/videoqa/vtextgqa/dataset.py:83-311,389-400)
run on synthetic videos.  Runs only in the authoring container.  The reference reads frames and features from hard-coded
absolute paths: ``glob.glob`` and ``np.load`` are patched to serve the synthetic video from memory, the dataset object is a
bare namespace carrying only the attributes those two methods touch, and the token / FastText / PHOC processors are small
deterministic stand-ins (their real counterparts are pinned elsewhere: targets.json, phoc_words.npz)."""
import importlib.machinery
import json
import os
import random
import sys
import types
from unittest import mock

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, "/root/reference")
import transformers  # noqa: E402


def stub(name):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    sys.modules[name] = m
    return m


stub("pytorch_transformers")
stub("pytorch_transformers.tokenization_bert").BertTokenizer = transformers.BertTokenizer
for name in ("editdistance", "demjson", "lmdb", "cv2", "torchtext", "torchtext.vocab", "git", "tensorboardX", "nltk", "fasttext", "fastText"):
    if name not in sys.modules:
        try:
            __import__(name)
        except Exception:
            stub(name)
from pythia.common.sample import Sample  # noqa: E402
from pythia.datasets.processors import CopyProcessor  # noqa: E402
from pythia.datasets.videoqa.vtextgqa import dataset as D  # noqa: E402
from pythia.utils.configuration import ConfigNode  # noqa: E402

rnd = random.Random(31)
words = "stop exit open coca cola 24/7 Main St. bus taxi joe's".split()


def make_video(n_frames, max_det, drop_last_key):
    info = {}
    for fr in range(1, n_frames + 1):
        dets = []
        for _ in range(rnd.randint(0, max_det)):
            x, y = rnd.uniform(0, 900), rnd.uniform(0, 500)
            w, h = rnd.uniform(5, 300), rnd.uniform(5, 100)
            j = lambda: rnd.uniform(-4, 4)          # noqa: E731  (slightly rotated quadrilaterals)
            pts = [x + j(), y + j(), x + w + j(), y + j(), x + w + j(), y + h + j(), x + j(), y + h + j()]
            dets.append({"points": pts, "ocr": rnd.choice(words), "ID": rnd.randint(0, 60)})
        info[str(fr)] = dets
    if drop_last_key:                                # fewer OCR entries than frames: the `len(ocr_info) >= frame_idx` branch
        del info[str(n_frames)]
    return info


def run_case(n_frames, F, P, max_det, drop_last_key, seed):
    rnd.seed(seed)
    info = make_video(n_frames, max_det, drop_last_key)
    feats = {i: np.random.RandomState(seed * 1000 + i).randn(1, 1024).astype(np.float32) for i in range(1, n_frames + 1)}
    rec = {"question_id": seed, "video_id": "vid%d" % seed, "question": "what is on the sign %d" % seed,
           "video_width": rnd.choice([640, 1280, 1920]), "video_height": rnd.choice([360, 720, 1080]),
           "answers": [rnd.choice(words) for _ in range(rnd.randint(1, 4))]}
    me = types.SimpleNamespace(
        num_frames=F, frame_ocr_num=P, ocr_info_dir=["/nonexistent/ocr"],
        text_processor=lambda d: {"token_inds": torch.arange(20), "token_num": torch.tensor(7)},
        ocr_token_processor=lambda d: {"text": d["text"].lower()},
        context_processor=lambda d: {"text": torch.zeros(len(d["tokens"]), 1), "tokens": d["tokens"], "length": torch.tensor(len(d["tokens"]))},
        phoc_processor=lambda d: {"text": torch.zeros(len(d["tokens"]), 1), "tokens": d["tokens"], "length": torch.tensor(len(d["tokens"]))},
        copy_processor=CopyProcessor(ConfigNode({"max_length": F * P})))

    def fake_load(path, allow_pickle=False):
        base = os.path.basename(path)
        if path.startswith("/nonexistent/ocr"):
            return np.array(info, dtype=object)
        return feats[int(base[:-4])]

    with mock.patch.object(D.glob, "glob", lambda pat: ["f%05d.jpg" % i for i in range(n_frames)]), mock.patch.object(D.np, "load", fake_load):
        s = D.VTEXTGQADataset.add_sample_details(me, dict(rec), Sample())
    random.seed(seed)
    answers = list(rec["answers"])
    cap = {}
    me.answer_processor = lambda a: cap.update(a) or {"answers_scores": 0, "sampled_idx_seq": 0, "train_prev_inds": 0, "train_loss_mask": 0}
    me.config = types.SimpleNamespace(fast_read=False)
    D.VTEXTGQADataset.add_answer_info(me, {"answers": answers}, s)
    out = {k: np.asarray(s[k]) for k in ("frame_id", "frame_mask", "video_feat", "temporal_id", "track_id", "ocr_mask", "ocr_bbox_coordinates", "frame_num")}
    meta = {"n_frames": n_frames, "F": F, "P": P, "seed": seed, "record": rec, "info": info, "context_tokens": list(s.context_tokens),
            "answers_10": cap["answers"], "answers_after_shuffle": answers}
    return out, meta


cases = [(3, 6, 4, 3, False), (6, 6, 4, 6, False), (7, 6, 4, 2, False), (25, 6, 4, 5, False), (13, 6, 2, 4, True), (64, 20, 3, 0, False),
         (1, 5, 5, 7, False), (12, 5, 1, 3, False), (5, 6, 3, 4, True)]
arrays, metas = {}, []
for ci, (n, F, P, md, drop) in enumerate(cases):
    out, meta = run_case(n, F, P, md, drop, 40 + ci)
    for k, v in out.items():
        arrays["c%d_%s" % (ci, k)] = v
    metas.append(meta)
sf = [[n, m, D.sample_frames(list(range(1, n + 1)), m)] for n in range(0, 40) for m in (1, 5, 6, 16)]
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "dataset_samples.npz"), meta=json.dumps({"cases": metas, "sample_frames": sf}), **arrays)
print("wrote %d cases, %d arrays" % (len(cases), len(arrays)))
