"""Staged inputs on the GPU: one pinned arena, one async H2D copy, device views -- must equal the reference's
field-by-field .to(device) (sample.py:296-326) bit for bit, also through the prefetch ring, and feed the model."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_staged_batch_equals_to_device_and_drives_the_model():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from vitxt_gqa_amd.staging import ArenaLayout, BatchStager
    from vitxt_gqa_amd.synth import make_batch, make_noise
    from vitxt_gqa_amd.testing import make_model, to_device
    F, P, V, B = 6, 8, 64, 2
    batches = [make_batch(B, F, P, V=V, seed=s, text_vocab=1000) for s in range(4)]
    st = BatchStager(ArenaLayout.from_batch(batches[0]), device=DEV, depth=2)
    assert st.slots[0].host_arena.is_pinned()
    model = make_model(F, P, V, text_vocab=1000, dtype=torch.float32).to(DEV).eval()
    model.train()
    outs = []
    for i, d in enumerate(st.prefetch(batches)):
        for k, v in batches[i].items():
            assert d[k].device.type == "cuda" and torch.equal(d[k].cpu(), v), (i, k)
        d["dataset_name"], d["dataset_type"] = "vtextgqa", "train"
        d["grounding_noise"] = tuple(t.to(DEV) for t in make_noise(B, F, P, seed=i))
        ref = to_device(batches[i], DEV)
        ref["grounding_noise"] = d["grounding_noise"]
        with torch.no_grad():
            a = model(d)["ref_scores"]
            b = model(ref)["ref_scores"]
        assert torch.equal(a, b)
        outs.append(a.sum().item())
    assert len(outs) == 4


def test_compact_wire_phoc_rows_built_on_device():
    """OCR tokens travel as 64-byte slots; context_feature_1 is a device-only arena field filled by t2s_phoc on the copy
    stream.  Must equal the oracle's rows for the same tokens (what the reference's PhocProcessor would have shipped)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle import phoc_oracle as po
    from vitxt_gqa_amd.staging import ArenaLayout, BatchStager, phoc_expander
    from vitxt_gqa_amd.synth import make_batch, make_token_slots
    B, F, P = 2, 4, 5
    batches = []
    for s in range(3):
        b = make_batch(B, F, P, V=32, seed=s, text_vocab=100)
        del b["context_feature_1"]
        b["ocr_token_slots"] = make_token_slots(B, F * P, seed=s)
        batches.append(b)
    st = BatchStager(ArenaLayout.from_batch(batches[0]), device=DEV, depth=2,
                     device_only={"context_feature_1": ((B, F * P, 604), torch.float32)}, post_upload=[phoc_expander()])
    for i, d in enumerate(st.prefetch(batches)):
        exp = po.build_phoc_batch(batches[i]["ocr_token_slots"].numpy().reshape(-1, 64)).reshape(B, F * P, 604)
        assert torch.equal(d["context_feature_1"].cpu(), torch.from_numpy(exp))
        assert torch.equal(d["context_feature_0"].cpu(), batches[i]["context_feature_0"])
