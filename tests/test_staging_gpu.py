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
