"""The train step as a whole (BaseTrainer._forward_pass / _backward, base_trainer.py:251-278: forward, both losses, backward,
global-norm clip 0.25, Adam) on the GPU: parameters after one step against the REFERENCE's own updated parameters (golden
fixture), a multi-step loss trajectory against the CPU oracle, and the throughput configuration (bf16 operands, dropout
0.1) actually learning a fixed batch."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from golden_util import Fixture  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _setup(fx, dtype, dropout=None):
    from vitxt_gqa_amd import training_config
    from vitxt_gqa_amd.optim import build_optimizer
    from vitxt_gqa_amd.testing import build_model_for_fixture, to_device
    model = build_model_for_fixture(fx, dtype).to(DEV).train()
    if dropout is not None:
        model.set_dropout(dropout)
    s = to_device(fx.batch(), DEV)
    s.grounding_noise = (fx["E1"], fx["E2"])
    s.grounding_masks = fx.masks()
    cfg = training_config()
    return model, s, cfg, build_optimizer(model, cfg)


def test_one_step_updates_match_reference():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from vitxt_gqa_amd.optim import train_step
    fx = Fixture("tiny_b2_f6_p8")
    model, s, cfg, opt = _setup(fx, torch.float32)
    before = {k: v.detach().clone() for k, v in model.named_parameters()}
    loss, norm, _ = train_step(model, opt, None, s, cfg)
    assert abs(loss.item() - fx["loss_total"].item()) < 1e-3 * fx["loss_total"].item()
    assert abs(norm.item() - fx["grad_total_norm"].item()) < 5e-3 * fx["grad_total_norm"].item()
    params = dict(model.named_parameters())
    checked = 0
    for k, v in fx.arr.items():
        if k.startswith("after:"):
            got = params[k[6:]].detach()[:8].cpu()
            moved = (before[k[6:]].cpu()[:8] - v).abs().max().item()
            assert moved > 5e-5                                   # Adam's first step moves every live element by ~lr = 1e-4
            assert (got - v).abs().max().item() < 5e-6, k
            checked += 1
    assert checked == 3


def test_four_step_trajectory_matches_oracle():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle import t2s_oracle as O
    from vitxt_gqa_amd.optim import train_step
    fx = Fixture("tiny_b2_f6_p8")
    model, s, cfg, opt = _setup(fx, torch.float32)
    sd = fx.state_dict(torch.float64)
    for k, v in sd.items():
        v.requires_grad_(not O.is_dead(k))
    b = {k: (v.double() if v.is_floating_point() else v) for k, v in fx.batch().items()}
    masks = {k: v.double() for k, v in fx.masks().items()}
    st, ref, got = {}, [], []
    for step in range(1, 5):
        ref.append(O.train_step(sd, b, fx.cfg, st, step, expo_frame=fx["E1"].double(), expo_ocr=fx["E2"].double(), inject_masks=masks)[:2])
        loss, norm, _ = train_step(model, opt, None, s, cfg)
        got.append((loss.item(), norm.item()))
    for (rl, rn), (gl, gn) in zip(ref, got):
        assert abs(gl - rl) < 2e-3 * abs(rl), (ref, got)
        assert abs(gn - rn) < 1e-2 * rn, (ref, got)
    assert ref[-1][0] < ref[0][0]                                  # and the oracle itself went downhill
    # parameters after 4 steps: compare the displacement from the start, relative to its own size (not the key biases:
    # softmax is shift invariant, their true gradient is 0 and Adam normalises whatever rounding noise is left)
    start = fx.state_dict(torch.float64)
    for name in ("mmt.encoder.layer.2.output.dense.weight", "ocr_ptr_net.query.weight", "TransLayer.encoder.layer.0.attention.self.query.bias",
                 "linear_obj_feat_to_mmt_in.weight"):
        d_ref = sd[name].detach() - start[name]
        d_got = dict(model.named_parameters())[name].detach().double().cpu() - start[name]
        assert (d_got - d_ref).norm().item() < 0.05 * d_ref.norm().item(), name


def test_bf16_dropout_training_learns_a_fixed_batch():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from vitxt_gqa_amd.optim import train_step
    fx = Fixture("tiny_b2_f6_p8")
    torch.manual_seed(11)
    model, s, cfg, opt = _setup(fx, torch.bfloat16, dropout=0.1)
    losses = []
    for _ in range(40):
        loss, _, out = train_step(model, opt, None, s, cfg)
        losses.append(loss.item())
        assert torch.isfinite(loss)
    first, last = sum(losses[:5]) / 5, sum(losses[-5:]) / 5
    assert last < 0.8 * first, losses


def test_fused_clip_adam_matches_torch_clip_and_adam():
    """FusedClipAdam.step_clipped (t2s_grad_sqnorm / t2s_clip_coef / t2s_adam_step) against torch.nn.utils.clip_grad_norm_ +
    torch.optim.Adam over several steps on tensors of awkward sizes (ragged tails, more than one 64 Ki chunk, two param groups
    with different learning rates, a parameter without a gradient), and its state_dict loads into a plain torch Adam."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from vitxt_gqa_amd.optim import FusedClipAdam
    g = torch.Generator().manual_seed(3)
    shapes = [(768, 768), (3072,), (5, 7), (1,), (200003,), (64, 1025)]
    pa = [torch.nn.Parameter(torch.randn(*s_, generator=g).to(DEV)) for s_ in shapes]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    dead_a, dead_b = torch.nn.Parameter(torch.ones(3, device=DEV)), torch.nn.Parameter(torch.ones(3, device=DEV))
    oa = FusedClipAdam([{"params": pa[:4] + [dead_a]}, {"params": pa[4:], "lr": 3e-4}], lr=1e-3, eps=1e-8)
    ob = torch.optim.Adam([{"params": pb[:4] + [dead_b]}, {"params": pb[4:], "lr": 3e-4}], lr=1e-3, eps=1e-8)
    for step in range(4):
        for x, y in zip(pa, pb):
            gr = torch.randn(x.shape, generator=g).to(DEV) * (10.0 if step == 1 else 0.01)     # step 1 clips hard, the others do not
            x.grad, y.grad = gr.clone(), gr.clone()
        na = oa.step_clipped(0.25)
        nb = torch.nn.utils.clip_grad_norm_(pb + [dead_b], 0.25)
        ob.step()
        assert abs(na.item() - nb.item()) < 1e-5 * nb.item()
        for x, y in zip(pa, pb):
            assert (x.grad - y.grad).abs().max().item() <= 1e-6 * y.grad.abs().max().item() + 1e-12        # clipped in place, as the reference
            assert (x - y).abs().max().item() < 2e-6, step
    assert torch.equal(dead_a, dead_b)
    sd = oa.state_dict()
    assert set(sd["state"]) == set(ob.state_dict()["state"]) and sd["state"][0]["step"].item() == 4.0
    oc = torch.optim.Adam([{"params": pb[:4] + [dead_b]}, {"params": pb[4:], "lr": 3e-4}], lr=1e-3, eps=1e-8)
    oc.load_state_dict(sd)                                                  # the reference's optimizer can resume from it
    assert 4 not in sd["state"]                                             # the parameter without a gradient has no state, as in torch
    assert torch.allclose(oc.state_dict()["state"][5]["exp_avg"], ob.state_dict()["state"][5]["exp_avg"], atol=1e-7)
