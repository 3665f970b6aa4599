"""``text_bert`` initialised from bert-base-uncased (reference: t2s.py:47-56, ``TextBert.from_pretrained`` + lr x 0.1 group,
t2s.py:356-376): the loader takes the embeddings and the first ``num_hidden_layers`` layers of a Hugging Face BERT
state_dict.  No network / checkpoint offline, so the dict is synthetic with the real key names and shapes."""
import os

import pytest
import torch

from vitxt_gqa_amd import build_model, t2s_model_config, training_config
from vitxt_gqa_amd.testing import setup_registry


def _hf_bert_state_dict(n_layers=12, vocab=30522, old_names=False, prefix="bert."):
    g = torch.Generator().manual_seed(5)
    sd = {}

    def t(*shape):
        return torch.randn(*shape, generator=g) * 0.02

    e = prefix + "embeddings."
    sd[e + "word_embeddings.weight"] = t(vocab, 768)
    sd[e + "position_embeddings.weight"] = t(512, 768)
    sd[e + "token_type_embeddings.weight"] = t(2, 768)
    ln_w, ln_b = ("gamma", "beta") if old_names else ("weight", "bias")
    sd[e + "LayerNorm." + ln_w] = t(768) + 1
    sd[e + "LayerNorm." + ln_b] = t(768)
    sd[e + "position_ids"] = torch.arange(512).unsqueeze(0)                      # a buffer newer checkpoints carry
    for i in range(n_layers):
        l = prefix + "encoder.layer.%d." % i
        for n in ("query", "key", "value"):
            sd[l + "attention.self.%s.weight" % n] = t(768, 768)
            sd[l + "attention.self.%s.bias" % n] = t(768)
        sd[l + "attention.output.dense.weight"] = t(768, 768)
        sd[l + "attention.output.dense.bias"] = t(768)
        sd[l + "attention.output.LayerNorm." + ln_w] = t(768) + 1
        sd[l + "attention.output.LayerNorm." + ln_b] = t(768)
        sd[l + "intermediate.dense.weight"] = t(3072, 768)
        sd[l + "intermediate.dense.bias"] = t(3072)
        sd[l + "output.dense.weight"] = t(768, 3072)
        sd[l + "output.dense.bias"] = t(768)
        sd[l + "output.LayerNorm." + ln_w] = t(768) + 1
        sd[l + "output.LayerNorm." + ln_b] = t(768)
    sd[prefix + "pooler.dense.weight"] = t(768, 768)
    sd[prefix + "pooler.dense.bias"] = t(768)
    sd["cls.predictions.bias"] = t(vocab)
    return sd


def _model(from_pretrained, path=None):
    setup_registry(40, 12)
    cfg = t2s_model_config(frame_num=3, ocr_frame_num=4)
    cfg["text_bert_init_from_bert_base"] = from_pretrained
    if path is not None:
        cfg["text_bert_pretrained_path"] = path
    return build_model(cfg)


@pytest.mark.parametrize("old_names,prefix", [(False, "bert."), (True, "bert."), (False, "")])
def test_load_pretrained_takes_embeddings_and_first_three_layers(old_names, prefix):
    m = _model(False)
    hf = _hf_bert_state_dict(old_names=old_names, prefix=prefix)
    used = m.text_bert.load_pretrained(hf)
    own = m.text_bert.state_dict()
    assert sorted(used) == sorted(own) and len(own) == 5 + 3 * 16
    ln_w = "gamma" if old_names else "weight"
    assert torch.equal(own["embeddings.word_embeddings.weight"], hf[prefix + "embeddings.word_embeddings.weight"])
    assert torch.equal(own["embeddings.LayerNorm.weight"], hf[prefix + "embeddings.LayerNorm." + ln_w])
    for i in range(3):
        assert torch.equal(own["encoder.layer.%d.output.dense.weight" % i], hf[prefix + "encoder.layer.%d.output.dense.weight" % i])
        assert torch.equal(own["encoder.layer.%d.attention.output.LayerNorm.weight" % i],
                           hf[prefix + "encoder.layer.%d.attention.output.LayerNorm.%s" % (i, ln_w)])
    # a truncated checkpoint is refused, a wrong shape too
    short = {k: v for k, v in hf.items() if "layer.2." not in k}
    with pytest.raises(KeyError):
        m.text_bert.load_pretrained(short)
    bad = dict(hf)
    bad[prefix + "embeddings.word_embeddings.weight"] = torch.zeros(100, 768)
    with pytest.raises(ValueError):
        m.text_bert.load_pretrained(bad)


def test_build_reads_the_checkpoint_file_and_keeps_the_small_learning_rate(tmp_path):
    hf = _hf_bert_state_dict()
    d = tmp_path / "bert-base-uncased"
    os.makedirs(d)
    torch.save(hf, str(d / "pytorch_model.bin"))
    m = _model(True, str(d))
    assert torch.equal(m.text_bert.embeddings.word_embeddings.weight, hf["bert.embeddings.word_embeddings.weight"])
    assert torch.equal(m.text_bert.encoder.layer[2].intermediate.dense.bias, hf["bert.encoder.layer.2.intermediate.dense.bias"])
    groups = m.get_optimizer_parameters(training_config())
    # [rest @lr], [text_bert @0.1 lr], [mmt @1.0 lr]  (t2s.py:356-376 with configs/t2s_abinet.yml lr_scale_text_bert 0.1)
    assert len(groups) == 3 and "lr" not in groups[0]
    assert groups[1]["lr"] == pytest.approx(1e-5) and [id(p) for p in groups[1]["params"]] == [id(p) for p in m.text_bert.parameters()]
    assert groups[2]["lr"] == pytest.approx(1e-4) and [id(p) for p in groups[2]["params"]] == [id(p) for p in m.mmt.parameters()]
    tb = {id(p) for p in m.text_bert.parameters()} | {id(p) for p in m.mmt.parameters()}
    assert not ({id(p) for p in groups[0]["params"]} & tb)
    # offline (no file): the flag still builds a model (random init) and still scales the learning rate
    m2 = _model(True, str(tmp_path / "nowhere"))
    assert len(m2.get_optimizer_parameters(training_config())) == 3
