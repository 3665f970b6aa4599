"""Helpers shared by the parity tests: load a committed golden fixture (tests/golden/*.npz, written
by tests/golden/make_golden.py from the reference itself) and regenerate its weights / inputs."""
import json
import os

import numpy as np
import torch

from vitxt_gqa_amd.init import fingerprint, make_state_dict
from vitxt_gqa_amd.schema import state_dict_schema
from vitxt_gqa_amd.synth import make_batch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Fixture:
    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, name + ".npz"))
        self.meta = json.loads(bytes(z["meta"]).decode())
        self.arr = {k: torch.from_numpy(np.asarray(z[k])) for k in z.files if k != "meta"}
        m = self.meta
        self.B, self.F, self.P, self.V = m["B"], m["F"], m["P"], m["V"]
        self.cfg = dict(frame_topk=5, ocr_topk=5, frame_num=self.F, ocr_frame_num=self.P)

    def __getitem__(self, k):
        return self.arr[k]

    def state_dict(self, dtype=torch.float32):
        m = self.meta
        sd = make_state_dict(state_dict_schema(m["V"], text_vocab=m["text_vocab"]), seed=m["seed"],
                             attn_gain=m["attn_gain"], gains=m.get("gains") or None)
        fp = fingerprint(sd, list(m["weight_fingerprint"]))
        for k, v in m["weight_fingerprint"].items():
            assert np.allclose(fp[k], v, rtol=1e-9, atol=1e-9), \
                "weight generator mismatch for %s (numpy stream differs from the fixture's)" % k
        return {k: v.to(dtype) for k, v in sd.items()}

    def batch(self):
        m = self.meta
        if "in:text" in self.arr:
            return {k[3:]: v for k, v in self.arr.items() if k.startswith("in:")}
        b = make_batch(m["B"], m["F"], m["P"], V=m["V"], seed=m["seed"], text_vocab=m["text_vocab"],
                       ocr_prev_frac=m.get("ocr_prev_frac", 0.0), ocr_keep=m.get("ocr_keep", 0.7), text_len=m.get("text_len"))
        for k, v in m["input_fingerprint"].items():
            assert abs(float(b[k].double().sum()) - v) <= 1e-6 * max(1.0, abs(v)), \
                "synthetic input generator mismatch for %s" % k
        return b

    def masks(self, prefix=""):
        return {k: self.arr[prefix + k] for k in ("pos_obj_mask", "neg_obj_mask", "pos_ocr_mask", "neg_ocr_mask")}
