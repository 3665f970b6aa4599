"""On-disk formats and per-question sample assembly on the producer side of the T2S boundary.
Host-side numpy, as in the reference; everything here ends in the 12 ``sample_list`` fields ``T2S.forward`` consumes
 plus ``targets`` / ``train_prev_inds`` / ``train_loss_mask`` (``targets.py``).

Mirrors ``VTEXTGQADataset`` (``pythia/datasets/videoqa/vtextgqa/dataset.py``):

* files -- the question list ``imdb*.npy`` (pickled object array whose first entry is a header, ``:34``), one
  ``<video>.npy`` per video holding the pickled dict ``{"<frame number>": [{"points": 8 numbers, "ocr": str, "ID": int},
  ...]}`` (``:99-100``) and one ``<frame number>.npy`` with a ``[1, 1024]`` ViT-L CLS row per extracted frame (``:267-272``);
* ``sample_frames`` (``:389-400``) -- uniform subsampling by an integer stride, which covers only the first
  ``stride * num_frames`` frames of a longer video;
* ``add_sample_details`` (``:83-283``) -- per sampled frame the first P detections (padded with ``<pad>`` slots whose
  ``temporal_id`` is still the frame number, ``:142-147``), boxes from the corner points normalised by the video size,
  frame padding with id 0 / mask 0;
* ``add_answer_info`` (``:286-311``) -- two of the shuffled answers stretched to the 10 slots the answer processor wants.

The reference fills every tensor element by element in Python loops; here each field is one vectorised numpy expression
written into a caller-provided row (so ``staging.BatchStager.collate`` can hand out views of the pinned batch arena and
the sample never exists as separate tensors).  ``build_sample`` without ``out`` allocates its own arrays.
"""
import os
import random

import numpy as np

PAD_TOKEN = "<pad>"


# ---- files -----------------------------------------------------------------------------------------------------------
def load_imdb(path):
    """Question records (dicts with question_id, video_id, question, answers, video_width, video_height); entry 0 of the
    file is a header and is dropped (dataset.py:34)."""
    return np.load(path, allow_pickle=True)[1:]


def load_ocr_info(ocr_dir, video):
    """``{frame number as str: list of detections}`` of one video (dataset.py:99-100)."""
    return np.load(os.path.join(ocr_dir, video + ".npy"), allow_pickle=True).item()


def load_vit_rows(feat_dir, video, frame_numbers):
    """[len(frame_numbers), 1024] float rows, one ``<n>.npy`` of shape [1, 1024] per frame (dataset.py:267-272,279)."""
    rows = [np.load(os.path.join(feat_dir, video, "%d.npy" % n), allow_pickle=True) for n in frame_numbers]
    return np.concatenate(rows, axis=0) if rows else np.zeros((0, 1024), np.float32)


def save_ocr_info(ocr_dir, video, info):
    np.save(os.path.join(ocr_dir, video + ".npy"), np.array(info, dtype=object), allow_pickle=True)


def save_imdb(path, records, header=None):
    arr = np.empty(len(records) + 1, dtype=object)
    arr[0] = header if header is not None else {"dataset_name": "vtextgqa"}
    for i, r in enumerate(records):
        arr[i + 1] = r
    np.save(path, arr, allow_pickle=True)


# ---- sample assembly -------------------------------------------------------------------------------------------------
def sample_frames(n_frames, num_frames):
    """Frame numbers (1-based) the reference keeps out of ``n_frames`` extracted frames (dataset.py:389-400)."""
    if n_frames <= num_frames:
        return np.arange(1, n_frames + 1)
    return 1 + (n_frames // num_frames) * np.arange(num_frames)


def detection_box(points):
    """Axis-aligned box of an 8-number corner list, with the reference's choice of corners (dataset.py:125-129):
    x1 from corners 0/3, y1 from corners 0/1, x2 from corners 1/2, y2 from corners 2/3."""
    p = points
    return [min(p[0], p[6]), min(p[1], p[3]), max(p[2], p[4]), max(p[5], p[7])]


def frame_detections(ocr_info, frame_number):
    """Detections of a sampled frame.  The reference looks the frame up only while ``len(ocr_info) >= frame number`` and
    otherwise falls back to the previous frame's entry (dataset.py:120-123); a missing key raises KeyError there too."""
    key = frame_number if len(ocr_info) >= frame_number else frame_number - 1
    return ocr_info[str(key)]


def alloc_sample(num_frames, frame_ocr_num):
    F, N = num_frames, num_frames * frame_ocr_num
    return {"frame_id": np.zeros(F, np.int64), "frame_mask": np.zeros(F, np.int64), "video_feat": np.zeros((F, 1024), np.float32),
            "temporal_id": np.zeros(N, np.int64), "track_id": np.zeros(N, np.int64), "ocr_mask": np.zeros(N, np.int64),
            "ocr_bbox_coordinates": np.zeros((N, 4), np.float32)}


def build_sample(record, ocr_info, n_frames, vit_rows, num_frames, frame_ocr_num, out=None):
    """Geometry / id / mask fields of one question.

    record: imdb entry (video_width, video_height); ocr_info: the video's detection dict; n_frames: number of extracted
    frames of the video; vit_rows: callable(frame numbers) -> [n, 1024] rows (``load_vit_rows`` bound to a directory).
    Returns (fields, ocr_tokens): ``fields`` are rows shaped like ``alloc_sample`` (written into ``out`` when given) plus
    the scalar ``frame_num``; ``ocr_tokens`` is the list of the ``frame_num * P`` raw token strings of the sampled frames
    (``<pad>`` in empty slots) for the token / FastText / PHOC processors, which pad to N themselves."""
    F, P = num_frames, frame_ocr_num
    f = out if out is not None else alloc_sample(F, P)
    for a in f.values():
        a[...] = 0
    numbers = sample_frames(n_frames, F)
    n = len(numbers)
    f["frame_id"][:n] = numbers
    f["frame_mask"][:n] = 1
    tokens = [PAD_TOKEN] * (n * P)                        # padded FRAMES get no token slots (:215); the processors pad
    temporal = f["temporal_id"].reshape(F, P)
    track, mask, box = f["track_id"].reshape(F, P), f["ocr_mask"].reshape(F, P), f["ocr_bbox_coordinates"].reshape(F, P, 4)
    temporal[:n] = numbers[:, None]                       # padded slots of a real frame still carry its number (:142-147)
    for i, number in enumerate(numbers.tolist()):
        dets = frame_detections(ocr_info, number)[:P]
        k = len(dets)
        if k:
            tokens[i * P:i * P + k] = [d["ocr"] for d in dets]
            track[i, :k] = [d["ID"] for d in dets]
            mask[i, :k] = 1
            box[i, :k] = np.asarray([detection_box(d["points"]) for d in dets], np.float32)
    # float32 boxes times float64 reciprocals, rounded back to float32 (:197-203)
    scale = np.array([1. / record["video_width"], 1. / record["video_height"]] * 2)
    box[:n] = (box[:n].astype(np.float32) * scale).astype(np.float32)
    if n:
        f["video_feat"][:n] = vit_rows(numbers.tolist())
    fields = dict(f)
    fields["frame_num"] = np.int64(n)
    return fields, tokens


def training_answers(answers, rng=random):
    """The 10 answer slots handed to the answer processor (dataset.py:286-293): the record's list is shuffled IN PLACE
    (the reference mutates its imdb entry), the first two survive, one answer fills all 10 slots, two fill 5 + 5."""
    rng.shuffle(answers)
    two = answers[:2]
    return two * 10 if len(two) == 1 else [two[0]] * 5 + [two[1]] * 5
