"""Deterministic, name-keyed weight initialisation.

Follows the reference's init rules: BERT-family modules built through
``BertPreTrainedModel.init_weights`` (``text_bert``, ``TransLayer``, ``mmt`` incl.
``prev_pred_embeddings``; t2s.py:382,527,554) get Linear/Embedding ~ N(0, 0.02), bias 0,
LayerNorm (1, 0); everything constructed directly on ``T2S`` / ``Grounding_Module`` /
``OcrPtrNet`` keeps torch defaults (Linear: U(+-1/sqrt(fan_in)) for weight and bias;
Embedding: N(0, 1)).

Each tensor is drawn from its own numpy PCG64 stream seeded by (seed, crc32(name)), so the
values do not depend on construction order, torch's RNG, or which other tensors exist.  This
lets tests regenerate the exact weights the golden fixtures were produced with instead of
committing 100s of MB of checkpoints.
"""
import math
import zlib

import numpy as np
import torch

_BERT_FAMILY = ("text_bert.", "TransLayer.", "mmt.")


def _rng(seed, name):
    return np.random.default_rng([int(seed), zlib.crc32(name.encode())])


def _gain(name, gains):
    g = 1.0
    for sub, v in (gains or {}).items():
        if sub in name:
            g *= v
    return g


def init_tensor(name, shape, seed=0, attn_gain=1.0, gains=None):
    """Return the float32 numpy init of parameter ``name``.  ``gains`` maps a name substring to a multiplier applied to
    the drawn values (fixtures use it to make the pointer head compete with the vocabulary head: ``ocr_ptr_net.`` up,
    ``classifier.module.weight`` down); the streams themselves do not depend on it."""
    g = _gain(name, gains)
    if g != 1.0:
        return (init_tensor(name, shape, seed, attn_gain) * np.float32(g)).astype(np.float32)
    r = _rng(seed, name)
    leaf = name.rsplit(".", 1)[-1]
    is_ln = "LayerNorm" in name or "layer_norm" in name
    if is_ln:
        return np.ones(shape, np.float32) if leaf == "weight" else np.zeros(shape, np.float32)
    if name.startswith(_BERT_FAMILY):
        if leaf == "bias":
            return np.zeros(shape, np.float32)
        w = (r.standard_normal(shape) * 0.02).astype(np.float32)
        if attn_gain != 1.0 and (".attention.self.query." in name or ".attention.self.key." in name):
            w *= attn_gain
        return w
    if name.endswith("embeddings.weight"):            # nn.Embedding default N(0,1)
        return r.standard_normal(shape).astype(np.float32)
    # nn.Linear default: weight U(+-1/sqrt(fan_in)); bias U(+-1/sqrt(fan_in))
    if leaf == "weight":
        bound = 1.0 / math.sqrt(shape[1])
        return r.uniform(-bound, bound, shape).astype(np.float32)
    fan_in = _linear_fan_in(name)
    bound = 1.0 / math.sqrt(fan_in)
    return r.uniform(-bound, bound, shape).astype(np.float32)


_FAN_IN = {"linear_obj_feat_to_mmt_in": 1074, "linear_obj_frame_to_mmt_in": 50,
           "linear_ocr_feat_to_mmt_in": 1004, "linear_ocr_bbox_to_mmt_in": 4,
           "frame_attn": 1536, "intermediate.dense": 768, "output.dense": 3072}


def _linear_fan_in(name):
    for k, v in _FAN_IN.items():
        if k in name and not (k == "output.dense" and "attention.output.dense" in name):
            return v
    return 768


def make_state_dict(schema, seed=0, attn_gain=1.0, dtype=torch.float32, gains=None):
    """schema: name -> shape (``schema.state_dict_schema``).  Returns name -> CPU tensor."""
    return {k: torch.from_numpy(init_tensor(k, tuple(s), seed, attn_gain, gains)).to(dtype) for k, s in schema.items()}


def fingerprint(sd, names=None):
    """Small, order-independent fingerprint used by the golden fixtures to detect a generator
    mismatch (different numpy stream) before blaming the kernels."""
    out = {}
    for k in (names or sorted(sd)):
        t = sd[k].double().flatten()
        out[k] = (float(t.sum()), float((t * t).sum()), float(t[:: max(1, t.numel() // 7)][:7].sum()))
    return out
