"""Round 5: a timed-out dQ hand-off must be LOUD, and the fused backward's LDS-DMA staging must not depend on
how the hardware range-checks a scalar buffer offset.

* dq_mode 0x101 is the hand-off with a dead predecessor (no block publishes its flags, spin limit 0): every successor's wait times
  out at once.  Asserted: the launch ends; the status word is set; the dQ rows of every pair with more than one key block are NaN
  (never a silently wrong number) while dK / dV stay finite; FusedClipAdam's step on such gradients is a no-op on the device and the
  NEXT step raises ops.HandoffTimeout; optim.raise_if_handoff_failed() raises at a synchronisation point and clears the word.
* The running dQ sums stay in the L2 of ONE XCD (plain stores, sc1 loads); the kernel checks the premise - every workgroup of an XCD
  group (equal blockIdx % 8) on the same XCD - and reports a violation as status bit 1.  Asserted: on this card the check is silent;
  dq_mode 0x401 (the check is fed alternating XCD numbers) sets bit 1, the optimizer step behind it is a no-op and
  optim.raise_if_handoff_failed() raises ops.HandoffPlacement; dq_mode 0x201 (write-through sums, the form for a device that places
  workgroups differently) gives bit-identical gradients and never runs the check.
* Lq % 64 != 0 with q / out / dout as views of larger buffers whose bytes right behind the last sample's rows are NaN: the staged rows
  behind Lq must read as zeros (the tile's row offset rides in the range-checked vector offset): finite gradients, equal to the
  two-kernel form's."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _case(B=2, L1=1500, n_dec=12, seed=5, frac=0.9):
    from vitxt_gqa_amd import ops
    L = L1 + n_dec
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(B, L, 2304, generator=g) * 0.7).to(DEV).to(torch.bfloat16)
    dout = torch.randn(B, L, 768, generator=g).to(DEV).to(torch.bfloat16)
    valid = (torch.rand(B, L1, generator=g) < frac).to(DEV)
    valid[:, 0] = True
    keys = ops.compact_keys(valid, n_dec=n_dec, dec_row0=L1)
    out, lse = ops.attn_fwd(x, keys)
    return x, dout, keys, out, lse


def test_a_timed_out_handoff_poisons_dq_and_sets_the_sticky_status():
    from vitxt_gqa_amd import ops, optim
    ops.reset_fused_status()
    x, dout, keys, out, lse = _case()
    good = ops.attn_bwd(x, out, dout, lse, keys, fused=True, dq_mode=1)
    assert ops.fused_handoff_status() == 0 and torch.isfinite(good.float()).all()
    bad = ops.attn_bwd(x, out, dout, lse, keys, fused=True, dq_mode=0x101)
    torch.cuda.synchronize()
    assert ops.fused_handoff_status() & 1, "the timeout must reach the sticky status word"
    dq = bad[..., :768].float()
    assert torch.isnan(dq).any(), "a chain whose hand-off failed must not hand back numbers"
    # every pair here has 4 key blocks (1 350 + 12 keys / 384): all of its dQ rows come from the poisoned chain
    assert torch.isnan(dq).all()
    assert torch.isfinite(bad[..., 768:].float()).all() and torch.equal(bad[..., 768:], good[..., 768:]), "dK / dV do not depend on the hand-off"
    with pytest.raises(ops.HandoffTimeout):
        optim.raise_if_handoff_failed()
    assert ops.fused_handoff_status() == 0, "raising clears the word"
    again = ops.attn_bwd(x, out, dout, lse, keys, fused=True, dq_mode=1)
    assert ops.fused_handoff_status() == 0 and torch.equal(again, good)


def test_the_optimizer_step_behind_a_timed_out_handoff_is_a_noop_and_the_next_one_raises():
    from vitxt_gqa_amd import ops, optim
    ops.reset_fused_status()
    torch.manual_seed(0)
    w = torch.nn.Parameter(torch.randn(300, 768, device=DEV))
    opt = optim.FusedClipAdam([w], lr=1e-2)
    w.grad = torch.randn_like(w)
    opt.step_clipped(0.25)                                   # a clean step moves the parameter
    torch.cuda.synchronize()
    w1 = w.detach().clone()
    m1 = opt.state[w]["exp_avg"].clone()
    x, dout, keys, out, lse = _case(B=1, L1=1100)
    ops.attn_bwd(x, out, dout, lse, keys, fused=True, dq_mode=0x101)
    w.grad = torch.full_like(w, float("nan"))               # what such a backward leaves in the gradients
    norm = opt.step_clipped(0.25)
    torch.cuda.synchronize()
    assert torch.isnan(norm), "the reported norm of a gated step is NaN"
    assert torch.equal(w.detach(), w1) and torch.equal(opt.state[w]["exp_avg"], m1), "the gated step must not touch parameters or moments"
    assert torch.isnan(w.grad).all()
    assert float(opt.state[w]["step"]) == 2.0               # (the host counted the gated step: it learns of the gate one step later)
    w.grad = torch.randn_like(w)
    with pytest.raises(ops.HandoffTimeout):
        opt.step_clipped(0.25)                               # the previous step's status word has reached the host
    assert ops.fused_handoff_status() == 0
    # the gated step applied nothing, so its count is taken back before the exception leaves - a trainer that catches the
    # error and carries on gets the bias correction of the updates that were really applied
    assert float(opt.state[w]["step"]) == 1.0
    opt.step_clipped(0.25)                                   # training could go on (from a checkpoint) once the cause is gone
    assert float(opt.state[w]["step"]) == 2.0
    torch.cuda.synchronize()
    assert not torch.equal(w.detach(), w1) and torch.isfinite(w).all()


def test_xcd_local_sums_are_checked_and_the_write_through_form_is_bit_identical():
    from vitxt_gqa_amd import ops, optim
    ops.reset_fused_status()
    x, dout, keys, out, lse = _case(B=3, L1=2100, seed=9)
    local = ops.attn_bwd(x, out, dout, lse, keys, fused=True, dq_mode=1, drop_p=0.1, drop_seed=77)
    torch.cuda.synchronize()
    assert ops.fused_handoff_status() == 0, "XCD groups on one XCD each: the placement check must be silent on this card"
    through = ops.attn_bwd(x, out, dout, lse, keys, fused=True, dq_mode=0x201, drop_p=0.1, drop_seed=77)
    assert ops.fused_handoff_status() == 0 and torch.equal(through, local), "same chain, same order: the scope of the stores changes no bit"
    # the write-through form does not look at the placement at all
    ops.attn_bwd(x, out, dout, lse, keys, fused=True, dq_mode=0x601, drop_p=0.1, drop_seed=77)
    assert ops.fused_handoff_status() == 0
    # the check, fed two XCD numbers per group
    flagged = ops.attn_bwd(x, out, dout, lse, keys, fused=True, dq_mode=0x401, drop_p=0.1, drop_seed=77)
    torch.cuda.synchronize()
    assert ops.fused_handoff_status() == 2, "a group seen on two XCDs sets bit 1 (and only that)"
    assert torch.equal(flagged, local)                       # (nothing was really misplaced here)
    w = torch.nn.Parameter(torch.randn(64, 768, device=DEV))
    opt = optim.FusedClipAdam([w], lr=1e-2)
    w0 = w.detach().clone()
    w.grad = torch.randn_like(w)
    norm = opt.step_clipped(0.25)
    torch.cuda.synchronize()
    assert torch.isnan(norm) and torch.equal(w.detach(), w0), "the step behind a flagged backward is gated off on the device"
    with pytest.raises(ops.HandoffPlacement):
        optim.raise_if_handoff_failed()
    assert ops.fused_handoff_status() == 0
    assert issubclass(ops.HandoffPlacement, ops.HandoffError) and issubclass(ops.HandoffTimeout, ops.HandoffError)


@pytest.mark.parametrize("drop_p", [0.0, 0.1])
def test_rows_behind_the_last_sample_are_never_staged(drop_p):
    from vitxt_gqa_amd import ops
    ops.reset_fused_status()
    B, L1, n_dec = 2, 1100 + 37, 12                           # L = 1149: the last query tile holds 61 rows, 3 are behind Lq
    L = L1 + n_dec
    assert L % 64 != 0
    g = torch.Generator().manual_seed(9)
    pad = 256                                                 # NaN rows right behind each buffer's last sample
    xb = torch.full((B * L + pad, 2304), float("nan"), dtype=torch.bfloat16, device=DEV)
    ob = torch.full((B * L + pad, 768), float("nan"), dtype=torch.bfloat16, device=DEV)
    db = torch.full((B * L + pad, 768), float("nan"), dtype=torch.bfloat16, device=DEV)
    xb[:B * L] = (torch.randn(B * L, 2304, generator=g) * 0.7).to(DEV).to(torch.bfloat16)
    db[:B * L] = torch.randn(B * L, 768, generator=g).to(DEV).to(torch.bfloat16)
    x, dout = xb[:B * L].view(B, L, 2304), db[:B * L].view(B, L, 768)
    valid = (torch.rand(B, L1, generator=g) < 0.8).to(DEV)
    valid[:, 0] = True
    keys = ops.compact_keys(valid, n_dec=n_dec, dec_row0=L1)
    kw = dict(drop_p=drop_p, drop_seed=77) if drop_p else {}
    o, lse = ops.attn_fwd(x, keys, **kw)
    ob[:B * L] = o.view(B * L, 768)
    out = ob[:B * L].view(B, L, 768)
    assert torch.isnan(xb[B * L:].float()).all() and torch.isnan(ob[B * L:].float()).all()
    got = ops.attn_bwd(x, out, dout, lse, keys, fused=True, dq_mode=1, **kw)
    torch.cuda.synchronize()
    assert ops.fused_handoff_status() == 0
    assert torch.isfinite(got.float()).all(), "a NaN behind the tensors reached the gradients: the rows behind Lq were read"
    two = ops.attn_bwd(x, out, dout, lse, keys, fused=False, **kw)
    scale = two.float().abs().max().item()
    assert (got.float() - two.float()).abs().max().item() < 3e-2 * max(1.0, scale)


@pytest.mark.parametrize("neighbour", ["gemm", "stream_copy", "both"])
def test_handoff_chain_under_a_cu_hungry_neighbour_stream(neighbour):
    """in data-parallel training an RCCL all-reduce of a gradient bucket runs on its own stream UNDER the fused attention
    backward (two 48 MB buckets are launched with ~146 ms of backward still to run).  The hand-off spin-waits between
    workgroups, and a collective kernel that takes CUs mid-launch changes which workgroups are resident while others spin - the one
    situation the single-stream tests never create.  Stand-in on one card: a second stream keeps the CUs busy with library GEMMs
    (every CU, in rounds) and / or 1 GB element-wise passes (HBM + L2 pressure on the XCD-local running sums) for the WHOLE duration of
    three fused backward launches at L = 10 132 with two unequal hand-off chains.  Asserted: the two streams really overlapped
    (events), the status word stays 0 (no bounded spin timed out, no XCD-group misplacement), and dQ / dK / dV are BIT-IDENTICAL to the
    solo launch - the ticket order guarantees a waiting block's predecessor has started whatever else occupies the card."""
    from vitxt_gqa_amd import ops
    ops.reset_fused_status()
    B, L1, n_dec = 4, 10120, 12
    L = L1 + n_dec
    g = torch.Generator().manual_seed(31)
    x = (torch.randn(B, L, 2304, generator=g) * 0.7).to(DEV).to(torch.bfloat16)
    dout = torch.randn(B, L, 768, generator=g).to(DEV).to(torch.bfloat16)
    keep = torch.tensor([0.7, 0.3, 0.9, 0.5]).view(B, 1)              # chains of 19, 8, 24 and 14 key blocks in one launch
    valid = (torch.rand(B, L1, generator=g) < keep).to(DEV)
    valid[:, 0] = True
    keys = ops.compact_keys(valid, n_dec=n_dec, dec_row0=L1)
    kw = dict(drop_p=0.1, drop_seed=4242)
    out, lse = ops.attn_fwd(x, keys, **kw)
    solo = ops.attn_bwd(x, out, dout, lse, keys, fused=True, dq_mode=1, **kw)
    torch.cuda.synchronize()
    assert ops.fused_handoff_status() == 0
    side = torch.cuda.Stream()
    a = torch.randn(8192, 8192, device=DEV, dtype=torch.bfloat16)
    b = torch.randn(8192, 8192, device=DEV, dtype=torch.bfloat16)
    big = torch.zeros(1 << 28, device=DEV, dtype=torch.float32)        # 1 GB
    base, s0, s1, m0, m1 = (torch.cuda.Event(enable_timing=True) for _ in range(5))
    base.record()
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        s0.record()
        for i in range(120):                                           # ~150 - 250 ms of neighbour work, enqueued before the backward
            if neighbour in ("gemm", "both"):
                torch.mm(a, b)
            if neighbour in ("stream_copy", "both"):
                big.add_(1.0)
        s1.record()
    m0.record()
    got = [ops.attn_bwd(x, out, dout, lse, keys, fused=True, dq_mode=1, **kw) for _ in range(3)]
    m1.record()
    torch.cuda.synchronize()
    t = {k: base.elapsed_time(e) for k, e in (("s0", s0), ("s1", s1), ("m0", m0), ("m1", m1))}
    overlap = min(t["s1"], t["m1"]) - max(t["s0"], t["m0"])
    print("neighbour %s: side stream %.1f..%.1f ms, fused backward x3 %.1f..%.1f ms, overlap %.1f ms" % (neighbour, t["s0"], t["s1"], t["m0"], t["m1"], overlap))
    assert overlap > 0.5 * (t["m1"] - t["m0"]), "the neighbour stream did not run under the backward launches: the rehearsal rehearsed nothing"
    assert ops.fused_handoff_status() == 0, "a hand-off wait timed out or a group was misplaced under the neighbour stream"
    for r in got:
        assert torch.equal(r, solo), "the hand-off's summation order (hence dQ) must not depend on what else runs on the card"
