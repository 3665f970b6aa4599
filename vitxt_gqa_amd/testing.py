"""Small helpers shared by tests, smoke() and bench.py (product-side: no oracle import here)."""
import torch

from . import build_model, registry, t2s_model_config
from .init import make_state_dict
from .sample import SampleList
from .schema import state_dict_schema


class _Writer:
    def write(self, *a, **k):
        pass


class _AnswerProcessor:
    BOS_IDX = 1


def setup_registry(V, N, dataset="vtextgqa"):
    registry.register("writer", _Writer())
    registry.register("config", {"datasets": dataset, "training_parameters": {"evalai_inference": False}})
    registry.register(dataset + "_num_final_outputs", V + N)
    registry.register(dataset + "_answer_processor", _AnswerProcessor())


def make_model(F, P, V, text_vocab=30522, seed=0, attn_gain=1.0, dtype=torch.bfloat16, state_dict=None, dropout=0.0):
    """T2S with name-seeded reference-style init (or the given state_dict); every dropout probability = ``dropout``
    (0 for parity runs; the reference default is 0.1)."""
    setup_registry(V, F * P)
    cfg = t2s_model_config(frame_num=F, ocr_frame_num=P)
    cfg.text_bert["vocab_size"] = text_vocab
    cfg.obj["dropout_prob"] = dropout
    cfg.ocr["dropout_prob"] = dropout
    for sec in ("text_bert", "translayers", "encoder", "mmt"):
        cfg[sec]["hidden_dropout_prob"] = dropout
        cfg[sec]["attention_probs_dropout_prob"] = dropout
    model = build_model(cfg)
    sd = state_dict if state_dict is not None else make_state_dict(
        state_dict_schema(V, text_vocab=text_vocab), seed=seed, attn_gain=attn_gain)
    model.load_state_dict(sd)
    return model.set_compute_dtype(dtype)


def build_model_for_fixture(fx, dtype):
    m = fx.meta
    return make_model(m["F"], m["P"], m["V"], text_vocab=m["text_vocab"], dtype=dtype, state_dict=fx.state_dict())


def to_device(batch, device):
    s = SampleList(batch).to(device, non_blocking=False)
    s.dataset_name, s.dataset_type = "vtextgqa", "train"
    return s
