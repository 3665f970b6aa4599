"""ConfigNode-like attribute dict (``pythia/utils/configuration.py:17-93``: supports ``cfg.key``,
``cfg["key"]``, ``**cfg`` and ``.get``) and the T2S hyper-parameters of ``configs/t2s_abinet.yml:60-135``
(SURVEY.md Appendix C).  Only config VALUES are in scope; the YAML include/override machinery is not."""
import copy


class ConfigNode(dict):
    def __init__(self, init=None):
        super().__init__()
        for k, v in (init or {}).items():
            self[k] = _wrap(v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = _wrap(v)

    def __deepcopy__(self, memo):
        return ConfigNode(copy.deepcopy(dict(self), memo))


def _wrap(v):
    if isinstance(v, dict) and not isinstance(v, ConfigNode):
        return ConfigNode(v)
    if isinstance(v, list):
        return [_wrap(x) for x in v]
    return v


def t2s_model_config(frame_num=64, ocr_frame_num=15, **overrides):
    """model_attributes.t2s of configs/t2s_abinet.yml:60-110 with (F, P) as free parameters."""
    n = frame_num * ocr_frame_num
    cfg = {
        "model": "t2s",
        "lr_scale_frcn": 0.1, "lr_scale_text_bert": 0.1, "lr_scale_mmt": 1.0,
        "text_bert_init_from_bert_base": False,      # no pretrained weights offline
        "text_bert": {"num_hidden_layers": 3},
        "obj": {"mmt_in_dim": 1074, "dropout_prob": 0.1},
        "ocr": {"mmt_in_dim": 1004, "dropout_prob": 0.1},
        "translayers": {"hidden_size": 768, "num_hidden_layers": 2},
        "grounding": {"frame_topk": 5, "ocr_topk": 5, "max_ocr_num": n, "frame_num": frame_num,
                      "ocr_frame_num": ocr_frame_num, "hidden_size": 768},
        "encoder": {"hidden_size": 768, "num_hidden_layers": 2},
        "mmt": {"hidden_size": 768, "num_hidden_layers": 3},
        "classifier": {"type": "linear", "ocr_max_num": n,
                       "ocr_ptr_net": {"hidden_size": 768, "query_key_size": 768}, "params": {}},
        "losses": [{"type": "pos_bce_loss", "weight": 1.0, "params": {}},
                   {"type": "InfoNCE", "weight": 1000, "params": {}}],
        "metrics": [],
    }
    node = ConfigNode(cfg)
    for k, v in overrides.items():
        node[k] = _wrap(v)
    return node


def training_config():
    """optimizer_attributes + training_parameters of configs/t2s_abinet.yml:112-135 (Appendix A, Q17)."""
    return ConfigNode({
        "optimizer_attributes": {"type": "Adam", "params": {"eps": 1.0e-08, "lr": 1e-4, "weight_decay": 0}},
        "training_parameters": {"clip_norm_mode": "all", "clip_gradients": True, "max_grad_l2_norm": 0.25,
                                "lr_scheduler": True, "lr_steps": [10000, 20000], "lr_ratio": 0.1,
                                "use_warmup": True, "warmup_factor": 0.2, "warmup_iterations": 1000,
                                "max_iterations": 24000, "batch_size": 48, "evalai_inference": False},
    })
