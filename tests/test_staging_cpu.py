"""Input staging (vitxt_gqa_amd/staging.py): arena layout, in-place collate and the prefetch ring on the CPU path.
Reference behaviour being replaced: BatchCollator + SampleList stacking (pythia/common/batch_collator.py:5-15,
sample.py:78-118) and the per-field .to(device) (sample.py:296-326)."""
import torch

from vitxt_gqa_amd.staging import ALIGN, ArenaLayout, BatchStager
from vitxt_gqa_amd.synth import make_batch


def _samples(batch):
    B = batch["text"].size(0)
    return [{k: v[i] for k, v in batch.items()} for i in range(B)]


def test_layout_alignment_and_views():
    b = make_batch(3, 4, 5, V=50, seed=1)
    lay = ArenaLayout.from_batch(b)
    assert list(lay.fields) == [k for k in b]
    end = 0
    for name, (off, nb, shape, dtype) in lay.fields.items():
        assert off % ALIGN == 0 and off >= end
        assert shape == tuple(b[name].shape) and dtype == b[name].dtype
        end = off + nb
    assert lay.nbytes % ALIGN == 0 and lay.nbytes >= end
    arena = torch.zeros(lay.nbytes, dtype=torch.uint8)
    views = lay.views(arena)
    views["text_len"].copy_(b["text_len"])
    assert torch.equal(lay.views(arena)["text_len"], b["text_len"])          # views alias the arena


def test_collate_matches_stacking_and_keeps_extras():
    b = make_batch(4, 3, 2, V=40, seed=2)
    st = BatchStager(ArenaLayout.from_batch(b), device="cpu", depth=2)
    samples = _samples(b)
    for i, s in enumerate(samples):
        s["question_id"] = 100 + i
    out = st.upload(st.collate(samples))
    for k, v in b.items():
        assert out[k].dtype == v.dtype and torch.equal(out[k], v), k
    assert out["question_id"] == [100, 101, 102, 103]
    assert out.get_batch_size() == 4


def test_prefetch_ring_reuses_slots_in_order():
    batches = [make_batch(2, 3, 2, V=30, seed=s) for s in range(5)]
    st = BatchStager(ArenaLayout.from_batch(batches[0]), device="cpu", depth=2)
    seen = []
    for i, d in enumerate(st.prefetch(batches)):
        for k, v in batches[i].items():
            assert torch.equal(d[k], v), (i, k)
        seen.append(d["text"].data_ptr())
    assert len(seen) == 5
    assert len(set(seen)) == 2                                             # two arenas, alternating
    assert seen[0] == seen[2] == seen[4] and seen[1] == seen[3]


def test_collate_rejects_wrong_batch_size():
    b = make_batch(2, 3, 2, V=30, seed=0)
    st = BatchStager(ArenaLayout.from_batch(b), device="cpu")
    import pytest
    with pytest.raises(ValueError):
        st.collate(_samples(make_batch(3, 3, 2, V=30, seed=0)))


def test_prefetch_surfaces_loader_errors_and_needs_two_slots():
    """A failure in the loader thread (fill / collate / upload) is re-raised in the consumer instead of replaying the previous,
    already released batch; a ring of one slot cannot prefetch."""
    import pytest
    from vitxt_gqa_amd.staging import ArenaLayout, BatchStager
    good = {"a": torch.arange(6, dtype=torch.float32).view(2, 3)}
    st = BatchStager(ArenaLayout.from_batch(good), device="cpu", depth=2)
    bad = {"a": torch.zeros(2, 4)}                      # wrong shape: copy_ into the pinned view raises in the loader thread
    seen = []
    with pytest.raises(RuntimeError, match="loader thread failed"):
        for d in st.prefetch([good, bad, good]):
            seen.append(d["a"].clone())
    assert len(seen) == 1 and torch.equal(seen[0], good["a"])
    with pytest.raises(ValueError):
        next(BatchStager(ArenaLayout.from_batch(good), device="cpu", depth=1).prefetch([good]))
