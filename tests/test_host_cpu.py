"""CPU-side checks: the C-ABI library builds, loads and exports every symbol of include/t2s_hip.h; the
host-side mirrors (registry / BaseModel / SampleList / config / optimizer hook / schema) behave like the
reference's; argument validation fails loudly.  No kernel is launched here."""
import ctypes
import json
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from vitxt_gqa_amd import build, hipext
    build.build(verbose=False)
    return hipext.lib()


def test_library_exports_every_declared_symbol(lib):
    from vitxt_gqa_amd import hipext
    hdr = open(os.path.join(ROOT, "include", "t2s_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(t2s_[a-z0-9_]+)\s*\(", hdr)))
    assert declared == hipext.exported_symbols()
    raw = ctypes.CDLL(hipext.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), name
    assert lib.t2s_abi_version() == 6


def test_argument_validation_reports_errors(lib):
    rc = lib.t2s_gelu_fwd(None, None, 4, 0, None)
    assert rc != 0 and b"null pointer" in lib.t2s_last_error()
    rc = lib.t2s_attn_fwd(*([ctypes.c_void_p(16)] * 5 + [None, None] + [1, 12, 0, 4, 0, 0] + [768] * 6 + [0.125, 1, 0.0, 0, None]))
    assert rc != 0 and b"bad shape" in lib.t2s_last_error()
    rc = lib.t2s_attn_fwd(*([ctypes.c_void_p(16)] * 5 + [None, None] + [1, 12, 4, 4, 0, 0] + [770] * 6 + [0.125, 1, 0.0, 0, None]))
    assert rc != 0 and b"16 bytes" in lib.t2s_last_error()


def test_ops_refuse_cpu_tensors(lib):
    from vitxt_gqa_amd import ops
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.gelu_fwd(torch.zeros(4, 4))


def test_state_dict_schema_matches_reference():
    from vitxt_gqa_amd.testing import make_model
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "state_dict_schema.json")))
    m = make_model(20, 30, 1000, text_vocab=30522, dtype=torch.float32)
    sd = m.state_dict()
    assert list(sd.keys()) == ref["keys"]
    for k, v in sd.items():
        assert list(v.shape) == ref["shapes"][k], k
    # module. prefix tolerance of checkpoint.py:98-111 is a pure key rename
    m.load_state_dict({k: v for k, v in sd.items()})


def test_registry_and_boundary_surface():
    from vitxt_gqa_amd import SampleList, registry, training_config
    from vitxt_gqa_amd.schema import is_dead_param
    from vitxt_gqa_amd.testing import make_model
    m = make_model(6, 8, 64, text_vocab=100, dtype=torch.float32)
    assert registry.get_model_class("t2s") is type(m)
    assert registry.get_loss_class("pos_bce_loss") is not None and registry.get_loss_class("InfoNCE") is not None
    groups = m.get_optimizer_parameters(training_config())
    assert len(groups) == 2 and "lr" not in groups[0] and groups[1]["lr"] == 1e-4     # [rest], [mmt @1.0*lr]
    dead = [n for n, p in m.named_parameters() if not p.requires_grad]
    assert all(is_dead_param(n) for n in dead) and len(dead) == 58
    # group MEMBERSHIP is the reference's (t2s.py:356-376): every parameter, the never-trained ones included, the mmt module's
    # in the second group - so optimizer.state_dict() indices line up with reference checkpoints
    mmt = list(m.mmt.parameters())
    assert [id(p) for p in groups[1]["params"]] == [id(p) for p in mmt]
    ids = {id(p) for p in mmt}
    assert [id(p) for p in groups[0]["params"]] == [id(p) for p in m.parameters() if id(p) not in ids]
    assert sum(len(g["params"]) for g in groups) == len(list(m.parameters()))
    s = SampleList({"text": torch.zeros(3, 20, dtype=torch.long)})
    assert s.get_batch_size() == 3 and s.text.shape == (3, 20) and SampleList([("a", 1)]).a == 1


def test_lr_schedule_and_clip_semantics():
    from vitxt_gqa_amd.optim import lr_lambda_update, clip_gradients, build_optimizer
    from vitxt_gqa_amd import training_config
    cfg = training_config()
    assert abs(lr_lambda_update(0, cfg) - 0.2) < 1e-12 and abs(lr_lambda_update(1000, cfg) - 1.0) < 1e-12
    assert abs(lr_lambda_update(500, cfg) - 0.6) < 1e-12
    assert lr_lambda_update(10000, cfg) == pytest.approx(0.1) and lr_lambda_update(20001, cfg) == pytest.approx(0.01)
    lin = torch.nn.Linear(4, 4)
    lin.weight.grad = torch.ones(4, 4)
    lin.bias.grad = torch.zeros(4)
    n = clip_gradients(lin, cfg)
    assert float(n) == pytest.approx(4.0) and float(lin.weight.grad.norm()) == pytest.approx(0.25, rel=1e-4)


def test_bench_flop_model_matches_survey_table():
    """bench.py's FLOP model against the figures of SURVEY section 8 (fwd FLOPs per sample, attention-GEMM share)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for (F, P), total, attn in (((20, 30), 1.20e11, 1.43e10), ((64, 15), 2.08e11, 3.75e10), ((100, 100), 5.10e12, 3.47e12),
                                ((100, 1), 3.95e10, 1.79e9), ((300, 200), 1.33e14, 1.23e14)):
        t, a = bench.flops_per_sample_fwd(F, P, 5000)
        assert abs(t - total) / total < 0.01, (F, P, t)
        assert abs(a - attn) / attn < 0.01, (F, P, a)


def test_bench_gpus_n_spawns_its_own_ranks():
    """`python bench.py --gpus N` with no launcher starts N ranks through torch.distributed.run (and refuses when the box has
    fewer GPUs) - checked without a GPU through the dry-run switch."""
    import json
    import subprocess
    import sys
    bench = os.path.join(ROOT, "bench.py")
    env = dict(os.environ, T2S_BENCH_DRY_SPAWN="1", T2S_BENCH_ONE_GPU="1")
    env.pop("RANK", None)
    out = subprocess.run([sys.executable, bench, "--gpus", "4", "--steps", "2", "--warmup", "1"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    cmd = json.loads(out.stdout.strip().splitlines()[-1])["spawn"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "4", "--steps", "2", "--warmup", "1"]
    env.pop("T2S_BENCH_ONE_GPU")
    if __import__("torch").cuda.device_count() < 4:
        out = subprocess.run([sys.executable, bench, "--gpus", "4"], env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 2 and "only" in out.stderr


def test_tuned_gemm_file_is_well_formed_and_inert_without_a_gpu():
    """vitxt_gqa_amd/tuned/*.csv: torch TunableOp's format (validator lines, then op, shape, solution, time); without a GPU the
    loader does nothing."""
    import os
    import vitxt_gqa_amd.gemm_tuning as G
    assert os.path.exists(G._FILE)
    rows = [l.strip().split(",") for l in open(G._FILE) if l.strip()]
    val = [r for r in rows if r[0] == "Validator"]
    ops_ = [r for r in rows if r[0] != "Validator"]
    assert {r[1] for r in val} >= {"PT_VERSION", "HIPBLASLT_VERSION", "GCN_ARCH_NAME"}
    assert any(r[2].startswith("gfx950") for r in val if r[1] == "GCN_ARCH_NAME")
    assert len(ops_) > 20 and all(len(r) == 4 and float(r[3]) > 0 for r in ops_)
    import torch
    if not torch.cuda.is_available():
        assert G.enable_tuned_gemms() is False


def test_compact_key_list_is_clamped_to_its_buffer_when_the_bound_is_violated():
    """KeyList.compact (the pruned K / V projection of the pos / neg passes) trusts a caller-vouched structural bound.  If the bound
    is wrong all the same, the compact list must stay inside the [B, capK] buffers the kernels index: cnt + n_dec <= capK for every
    sample, every gathered row a real row of its own sample, the decoder keys still closing the list.  Pure tensor
    logic: runs on the CPU."""
    from vitxt_gqa_amd.ops import KeyList
    B, L1, n_dec = 3, 300, 12
    L = L1 + n_dec
    valid = torch.zeros(B, L1, dtype=torch.bool)
    valid[0, :40] = True            # inside the bound
    valid[1, ::2] = True            # 150 keys: violates a bound of 100
    valid[2, :] = True              # 300 keys: violates it grossly
    idx = torch.zeros(B, L, dtype=torch.int32)
    cnt = valid.sum(1).to(torch.int32)
    for b in range(B):
        rows = torch.nonzero(valid[b]).flatten().to(torch.int32)
        idx[b, :len(rows)] = rows
        idx[b, len(rows):len(rows) + n_dec] = torch.arange(L1, L, dtype=torch.int32)
    keys = KeyList(idx, cnt, n_dec, L1, cap_hint=100 + n_dec)
    keys.bound_is_structural = True
    keys_c, flat, capK = keys.compact(L)
    assert capK == 128 and keys_c.idx.shape == (B, capK) and flat.shape == (B * capK,)
    assert int((keys_c.cnt + n_dec).max()) <= capK and keys_c.cnt.tolist() == [40, 116, 116]
    flat = flat.view(B, capK)
    for b in range(B):
        n = int(keys_c.cnt[b])
        assert (flat[b] >= b * L).all() and (flat[b] < (b + 1) * L).all()                     # rows of the sample's own sequence
        assert flat[b, :n].tolist() == (idx[b, :n].long() + b * L).tolist()                   # the first n prefix keys, list order
        assert flat[b, n:n + n_dec].tolist() == list(range(b * L + L1, b * L + L))            # the decoder keys close the list
        assert (flat[b, n + n_dec:] == b * L).all()                                            # positions behind the list: row 0
    # an honest bound changes nothing
    k2 = KeyList(idx[:1], cnt[:1], n_dec, L1, cap_hint=40 + n_dec)
    k2c, f2, cap2 = k2.compact(L)
    assert cap2 == 64 and k2c.cnt.tolist() == [40] and f2[:52].tolist() == idx[0, :52].long().tolist()


def test_backward_form_policy():
    """ops._fused_policy (which attention-backward form a call takes when the caller does not say): with the ordered hand-off every
    long sequence takes the fused five-product kernel whatever its key bound (profiles/r04_light_launches.txt); with fp32 atomics only
    lists bounded at >= 2 048 keys do; short sequences (text_bert: 20 rows) and fp32 never; an explicit choice wins."""
    import types
    import torch
    from vitxt_gqa_amd import ops
    bf, f32 = torch.empty(1, dtype=torch.bfloat16), torch.empty(1, dtype=torch.float32)
    k = lambda hint: types.SimpleNamespace(cap_hint=hint)
    assert ops._fused_policy(None, bf, k(74), 10132, 1) and ops._fused_policy(None, bf, k(10132), 10132, 1)
    assert not ops._fused_policy(None, bf, k(74), 10132, 0) and ops._fused_policy(None, bf, k(2048), 10132, 0)
    assert not ops._fused_policy(None, bf, k(20), 20, 1) and not ops._fused_policy(None, bf, k(549), 1000, 1)
    assert not ops._fused_policy(None, f32, k(10132), 10132, 1) and not ops._fused_policy(True, f32, k(10132), 10132, 1)
    assert ops._fused_policy(True, bf, k(20), 20, 0) and not ops._fused_policy(False, bf, k(10132), 10132, 1)


def test_handoff_scope_switch_keeps_the_launch_policy():
    """T2S_FB_HANDOFF_SCOPE=agent sets dq_mode bit 9 (write-through running sums) on top of the hand-off: the backward-form policy
    and everything else that asks "is this the hand-off" looks at the low byte only."""
    import subprocess
    import sys
    import types
    import torch
    from vitxt_gqa_amd import ops
    bf = torch.empty(1, dtype=torch.bfloat16)
    k = lambda hint: types.SimpleNamespace(cap_hint=hint)
    assert ops._fused_policy(None, bf, k(74), 10132, 0x201) and not ops._fused_policy(None, bf, k(74), 10132, 0)
    env = dict(os.environ, T2S_FB_HANDOFF_SCOPE="agent")
    out = subprocess.run([sys.executable, "-c", "from vitxt_gqa_amd import ops; print(ops.ATTN_BWD_DQ_MODE)"], env=env, cwd=ROOT,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.split()[-1] == str(0x201), (out.stdout, out.stderr)
    env["T2S_ATTN_BWD_DQ"] = "atomic"          # the atomic form has no running sums: the switch is ignored
    out = subprocess.run([sys.executable, "-c", "from vitxt_gqa_amd import ops; print(ops.ATTN_BWD_DQ_MODE)"], env=env, cwd=ROOT,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.split()[-1] == "0", (out.stdout, out.stderr)


def test_status_words_decode_as_an_or_over_devices_and_survive_a_max_over_ranks():
    """(i) `fused_handoff_status(None)` ORs the per-device words (a sum turned two timeouts into "placement"); (ii) the MAX
    reduction of `GradBuckets.finish()` over ranks keeps BOTH bits when one rank timed out (1) and another saw a placement violation (2):
    the bits travel as 0 / 1 flags in words 2 and 3."""
    import torch
    from vitxt_gqa_amd import ops
    assert ops.decode_status_words([0, 5, 0, 0]) == 0
    assert ops.decode_status_words([1, 5, 1, 0]) == 1 and ops.decode_status_words([2, 5, 0, 1]) == 2
    rank_a, rank_b = torch.tensor([1, 11, 1, 0]), torch.tensor([2, 11, 0, 1])
    assert ops.decode_status_words(torch.maximum(rank_a, rank_b).tolist()) == 3          # MAX of the words alone would say 2
    saved = dict(ops._STICKY)
    try:
        ops._STICKY.clear()
        ops._STICKY[0] = torch.tensor([1, 3, 1, 0], dtype=torch.int32)
        ops._STICKY[1] = torch.tensor([1, 3, 1, 0], dtype=torch.int32)
        assert ops.fused_handoff_status() == 1                                            # two timeouts stay a timeout
        ops._STICKY[1] = torch.tensor([2, 3, 0, 1], dtype=torch.int32)
        assert ops.fused_handoff_status() == 3
    finally:
        ops._STICKY.clear()
        ops._STICKY.update(saved)
