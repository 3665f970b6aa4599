"""bench.py's one-line JSON contract, on a small configuration (BASELINE configs[0]-sized: B=2, 20 frames x 30 OCR): the keys the
driver reads, the roofline / cpu_baseline objects, null traffic for a configuration profiles/traffic.json does not hold."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--batch", "2", "--frames", "20", "--ocr", "30", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline"] + list(extra)
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line on stdout"
    return json.loads(lines[0])


def test_bench_line_contract_train_step():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    d = _run()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["unit"] == "samples/s" and d["data"] == "synthetic" and d["dtype"] == "bf16" and d["vs_baseline"] is None
    assert abs(d["value"] - 2 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]            # value = samples of the K steps / their time
    assert "workload" in d["config"] and "model" not in d["config"] and d["config"]["global_batch"] == 2
    assert "configs[" not in d["config"]["workload"]                                           # not one of BASELINE's named shapes
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0 < r["frac"] < 1
    assert r["traffic"] is None                                                               # measured for the B=64 100x100 configuration only
    assert "dropout_0" in d and d["dropout_0"]["ms_per_step"] > 0
    assert "cpu_baseline" not in d                                                            # --no-cpu-baseline


def test_bench_line_contract_forward_only():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    d = _run("--forward-only")
    assert "forward" in d["config"]["workload"] and "loss" not in d
    assert d["roofline"]["bound"] == "mfma" and "roofline_fwd" not in d


def test_bench_two_ranks_end_to_end_on_one_card():
    """`python bench.py --gpus 2` END TO END: the parent spawns torch.distributed.run, two fresh ranks share the one
    card of the test box (T2S_BENCH_ONE_GPU=1) over gloo (T2S_BENCH_BACKEND=gloo: RCCL needs one GPU per rank), shard the
    questions with the reference's sampler, run the bucketed gradient exchange from the backward hooks, the fused scalar reduce
    of the logged losses, the barrier and the MAX-over-ranks timing, and rank 0 prints the one JSON line."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, T2S_BENCH_BACKEND="gloo", T2S_BENCH_ONE_GPU="1")
    env.pop("RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "2", "--frames", "20", "--ocr", "30", "--steps", "2",
           "--warmup", "1", "--no-cpu-baseline", "--no-dropout0"]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, env=env)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line (rank 0 only)"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["backend"] == "gloo"
    assert d["config"]["global_batch"] == 4 and d["config"]["parallelism"] == "dp2" and d["scaling"] == "weak"
    assert d["collectives_per_step"] == d["n_buckets"] and d["n_buckets"] >= 1           # every bucket reduced once per step
    assert d["scalar_reduces_per_step"] == 1
    assert abs(d["value"] - 2 * 2 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]    # whole-job samples of the K steps / MAX time
    assert "cpu_baseline" not in d and d["loss"] == d["loss"]
    # the fields that make a badly scaling first multi-GPU run diagnosable
    m = d["multi_gpu"]
    assert m["ranks"] == 2 and all(len(v) == 2 for v in m["per_rank"].values())
    el = m["per_rank"]["elapsed_s"]
    assert m["elapsed_max_s"] == max(el) and m["elapsed_min_s"] == min(el) and el[m["rank_of_max"]] == max(el)
    assert max(el) <= d["ms_per_step"] * 2e-3 + 1e-6                                    # no rank's own time exceeds the MAX the line reports
    assert all(w >= 0 for w in m["per_rank"]["finish_wait_gpu_ms_per_step"]) and m["exposed_allreduce_wait_ms_per_step_max"] >= 0
    assert m["per_rank"]["handoff_status"] == [0.0, 0.0] and m["handoff_status_or"] == 0
    assert all(v >= 0 for v in m["per_rank"]["fused_launches"]) and all(v > 0 for v in m["per_rank"]["attn_bwd_ms_per_step"])
    assert len(m["buckets_mb"]) == d["n_buckets"] and "rccl_version" in m["comm"] and "channels" in m["comm"]
