#!/usr/bin/env python3
"""Headline benchmark: T2S-QA train-step samples/sec on synthetic 100-frame x 100-OCR x 20-q-token batches.

    python bench.py --gpus N --steps K --warmup W           (N > 1: launched by torch.distributed.run)

A step = forward (TextBert, embeds, QTV, grounding, 3 x MMT + heads) + pos_bce_loss + 1000*InfoNCE +
backward + global-norm clip 0.25 + Adam, batch 64 per GPU (BASELINE.json configs[2]; weak scaling:
per-GPU batch fixed, questions sharded across ranks, one RCCL gradient all-reduce per step).
Prints ONE JSON line on rank 0 (contract in the task statement) carrying `roofline` (the dominant kernel group of the
step: the attention backward, HIP-event timed inside the timed region; `roofline_fwd` = the attention forward kernel)
and, at N=1, `cpu_baseline` (the CPU oracle restatement of the reference timed on this box's host cores on a bounded
sample).  `python bench.py --gpus N` without a launcher starts the N ranks itself (spawn_ranks).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0       # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
T_Q, DEC = 20, 12
HID, FFN = 768, 3072


def flops_per_sample_fwd(F, P, V):
    """FLOP model of BASELINE.md section 4."""
    N = F * P
    L1, L2 = T_Q + F + N, T_Q + F + N + DEC
    layer = lambda L: 24 * L * HID * HID + 4 * L * L * HID
    attn = 4 * HID * (3 * T_Q ** 2 + 2 * L1 ** 2 + 9 * L2 ** 2)
    total = (3 * layer(T_Q) + 2 * layer(L1) + 9 * layer(L2) + 2 * HID * (1074 * F + 1008 * N)
             + 3 * (24 * HID * V + 24 * HID * HID + 2 * N * HID * HID + 24 * N * HID))
    return total, attn


class AttnTimer:
    """Wraps ops.attn_fwd / ops.attn_bwd with HIP events on the launch stream (torch's current stream is the stream the
    kernels are enqueued on); accumulates FLOPs and device time per launch."""

    def __init__(self):
        from vitxt_gqa_amd import ops
        self.ops, self.orig_f, self.orig_b = ops, ops.attn_fwd, ops.attn_bwd
        self.fwd, self.bwd, self.enabled = [], [], False

    def _timed(self, orig, store, prods):
        def f(qkv, *args, **kwargs):
            if not self.enabled:
                return orig(qkv, *args, **kwargs)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = orig(qkv, *args, **kwargs)
            e1.record()
            keys = args[0] if prods == 2 else args[3]
            B, L, _ = qkv.shape
            store.append((e0, e1, B, L, keys.cnt, keys.n_dec, self.ops.LAST_ATTN_BWD_PRODUCTS if prods == 5 else 2))
            return r
        return f

    def __enter__(self):
        self.ops.attn_fwd = self._timed(self.orig_f, self.fwd, 2)
        self.ops.attn_bwd = self._timed(self.orig_b, self.bwd, 5)
        return self

    def __exit__(self, *a):
        self.ops.attn_fwd, self.ops.attn_bwd = self.orig_f, self.orig_b

    @staticmethod
    def _summary(events, products, min_flops=1e9):
        """products: matrix products per (query, key) pair counted as algorithmic work (forward 2, backward 5)."""
        big = []
        for a, b, B, L, cnt, nd, executed in events:
            dense = 2.0 * products * B * 12 * L * L * 64
            if dense >= min_flops:
                vis = float((cnt.sum() + cnt.numel() * nd).item())
                alg = 2.0 * products * 12 * 64 * L * vis
                big.append((a.elapsed_time(b) * 1e-3, dense, alg, alg * executed / products, executed, vis / max(1, cnt.numel())))
        if not big:
            return None
        t, f, fx, fe = sum(x[0] for x in big), sum(x[1] for x in big), sum(x[2] for x in big), sum(x[3] for x in big)
        fused = [x for x in big if x[4] == 5]
        tf = sum(x[0] for x in fused)
        long_ = [x for x in big if x[5] >= 2048]          # launches whose lists average >= 2 048 keys (the ref pass and QTV)
        return dict(long_launches=len(long_), long_avg_ms=(1e3 * sum(x[0] for x in long_) / len(long_)) if long_ else None,
                    launches=len(big), avg_ms=1e3 * t / len(big), tflops_dense=f / t / 1e12, tflops=fx / t / 1e12, tflops_executed=fe / t / 1e12,
                    total_ms=1e3 * t, fused_launches=len(fused), fused_avg_ms=(1e3 * tf / len(fused)) if fused else None,
                    fused_tflops=(sum(x[2] for x in fused) / tf / 1e12) if fused else None)

    def summary_fwd(self):
        return self._summary(self.fwd, 2)

    def summary_bwd(self):
        return self._summary(self.bwd, 5)


class _Budget(Exception):
    pass


def cpu_baseline(V, seed=0, budget_s=330, full=(100, 100), guard_s=230):
    """Reference semantics (CPU oracle) timed on the host cores: ONE REAL full train step (forward, both losses, backward, clip,
    Adam; fp32; B=1) at the metric's own shape, 100 frames x 100 OCR tokens (L = 10 132), with the oracle's attention evaluated by
    torch's fused CPU attention on the same additive masks (oracle.ATTENTION_IMPL = "sdpa": BASELINE.md section 3 allows it; the
    eager form needs ~100 GB of [12, L, L] autograd state there; tests/test_oracle_golden.py pins the two forms to each other).
    Protocol: a warm-up + timed steps at 100 x 20 (L = 2 132) first - they warm the thread pool and price the full-size step by
    the FLOP model; the full-size point then runs 1 warm-up + up to 3 timed steps while an estimate of the next step fits the
    wall guard (``guard_s``), at least ONE timed step (without the warm-up when even two steps would not fit).  ``kind`` is
    "port, measured" when the full-size step was timed; if not even one step fits the guard (a box with few or slow cores) the
    largest measured point is EXTRAPOLATED by the FLOP model and ``kind`` says "port, extrapolated".  Runs BEFORE the GPU is
    touched, under a hard wall-clock budget."""
    import signal
    from oracle import t2s_oracle as O
    from vitxt_gqa_amd.init import make_state_dict
    from vitxt_gqa_amd.schema import state_dict_schema
    from vitxt_gqa_amd.synth import make_batch, make_noise
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 64))
    torch.set_num_threads(cores)
    Fs, Bs = full[0], 1
    f_full, _ = flops_per_sample_fwd(full[0], full[1], V)
    sd = make_state_dict(state_dict_schema(V), seed=seed)
    for k, v in sd.items():
        v.requires_grad_(not O.is_dead(k))
    result = {"value": None, "unit": "samples/s", "cores": cores, "kind": "port, extrapolated", "sample": "not measured (budget exceeded)",
              "attention": "torch SDPA (CPU) on the reference's additive masks"}
    points = []

    def on_alarm(signum, frame):
        raise _Budget()

    old = signal.signal(signal.SIGALRM, on_alarm)
    signal.alarm(int(budget_s))
    t_start = time.time()
    prev_impl = O.ATTENTION_IMPL
    O.ATTENTION_IMPL = "sdpa"
    try:
        for Ps in sorted({min(20, full[1]), full[1]}):
            is_full = Ps == full[1]
            L = T_Q + Fs + Fs * Ps + DEC
            f_s, _ = flops_per_sample_fwd(Fs, Ps, V)
            warm, max_steps = True, 3
            if points:
                est = points[-1]["s_per_step"] * f_s / points[-1]["flops"]              # FLOP-model estimate of one step here
                left = min(guard_s, budget_s * 0.95 - (time.time() - t_start))
                if est > left:
                    result["skipped_point"] = "B=1 x %d x %d (L=%d): estimated %.0f s/step against %.0f s left under the wall guard" % (Fs, Ps, L, est, left)
                    break
                warm = 2 * est <= left
                max_steps = max(1, min(3, int(left / est) - (1 if warm else 0)))
            batch = make_batch(Bs, Fs, Ps, V=V, seed=seed)
            e1, e2 = make_noise(Bs, Fs, Ps, seed)
            cfg = dict(frame_topk=5, ocr_topk=5, frame_num=Fs, ocr_frame_num=Ps)
            st = {}
            t_pt = time.time()
            if warm:
                O.train_step(sd, batch, cfg, st, 1, expo_frame=e1, expo_ocr=e2)      # untimed warm-up step at this shape
            times = []
            while len(times) < max_steps:
                t0 = time.time()
                O.train_step(sd, batch, cfg, st, len(times) + 2, expo_frame=e1, expo_ocr=e2)
                times.append(time.time() - t0)
                if is_full and (time.time() - t_pt) + times[-1] > guard_s:
                    break
            n = len(times)
            dt = sum(times) / n
            sd_t = (sum((x - dt) ** 2 for x in times) / max(1, n - 1)) ** 0.5
            points.append({"frames": Fs, "ocr_per_frame": Ps, "L": L, "s_per_step": dt, "s_per_step_std": sd_t, "steps": n, "warmup_steps": int(warm),
                           "flops": f_s, "samples_per_s": Bs / dt, "samples_per_s_std": Bs * sd_t / (dt * dt), "full": is_full})
    except _Budget:
        result["sample"] = "stopped by the %ds wall-clock budget" % budget_s
    finally:
        signal.alarm(0)
        signal.signal(signal.SIGALRM, old)
        O.ATTENTION_IMPL = prev_impl
    if points:
        big = points[-1]
        measured = bool(big["full"])
        what = ("oracle (plain-torch CPU restatement of the reference, attention through torch SDPA) full train step, fp32, %d threads, "
                "B=%d x %d frames x %d OCR/frame (L=%d): %d warm-up + %d timed step(s), %.2f +- %.2f s/step = %.4f samples/s MEASURED"
                % (cores, Bs, Fs, big["ocr_per_frame"], big["L"], big["warmup_steps"], big["steps"], big["s_per_step"], big["s_per_step_std"],
                   big["samples_per_s"]))
        result.update(kind="port, measured" if measured else "port, extrapolated",
                      value=big["samples_per_s"] * (1.0 if measured else big["flops"] / f_full),
                      measured_samples_per_s_at_sample_shape=big["samples_per_s"], measured_samples_per_s_std=big["samples_per_s_std"],
                      measured_points=[{k: p[k] for k in ("frames", "ocr_per_frame", "L", "s_per_step", "s_per_step_std", "steps", "warmup_steps")} for p in points],
                      sample=what + (" at the metric's own shape: value = that rate" if measured else
                                     "; value = that rate EXTRAPOLATED by the FLOP model (x%.4f) to the %d x %d workload, whose step did not fit "
                                     "the %d s wall guard on this host" % (big["flops"] / f_full, full[0], full[1], guard_s)))
        if len(points) >= 2:
            import math
            a, b = points[-2], points[-1]
            result.update(measured_scaling_exponent_in_L=math.log(b["s_per_step"] / a["s_per_step"]) / math.log(b["L"] / a["L"]),
                          flop_model_exponent_in_L=math.log(b["flops"] / a["flops"]) / math.log(b["L"] / a["L"]))
    return result


RANK_REPORT_FIELDS = ("elapsed_s", "finish_wait_gpu_ms_per_step", "finish_wait_host_ms_per_step", "handoff_status", "fused_launches",
                      "attn_bwd_ms_per_step", "attn_fwd_ms_per_step", "peak_mem_gb")


def _or_of(words):
    st = 0
    for v in words:
        if v is not None:          # (None = a rank that did not report the field)
            st |= int(v)
    return st


def gather_rank_report(local, device=None, group=None):
    """What makes a first multi-GPU run diagnosable from its one JSON line: every rank contributes one float64 vector
    (RANK_REPORT_FIELDS: its own wall time over the timed region, the time its compute stream / its host sat in
    ``GradBuckets.finish()`` per step = the EXPOSED part of the gradient all-reduce, its sticky hand-off status word and fused launch
    count, its attention kernel time per step), gathered with ONE ``all_gather``; rank 0 prints the per-rank lists plus min / max /
    rank-of-max of the step time.  Works on any backend (``dist.all_gather`` of equal-sized tensors); with no process group it reports
    the single rank."""
    import torch.distributed as dist
    vec = torch.tensor([float(local.get(k) if local.get(k) is not None else float("nan")) for k in RANK_REPORT_FIELDS], dtype=torch.float64,
                       device=device if device is not None else "cpu")
    if dist.is_available() and dist.is_initialized():
        world = dist.get_world_size(group)
        got = [torch.empty_like(vec) for _ in range(world)]
        dist.all_gather(got, vec, group=group)
    else:
        got = [vec]
    rows = [g.cpu().tolist() for g in got]
    per = {k: [(r[i] if r[i] == r[i] else None) for r in rows] for i, k in enumerate(RANK_REPORT_FIELDS)}      # NaN (not reported) -> null: strict JSON
    el = per["elapsed_s"]
    slow = max(range(len(el)), key=lambda r: el[r])
    return {"ranks": len(rows), "per_rank": per, "elapsed_min_s": min(el), "elapsed_max_s": max(el), "rank_of_max": slow,
            "spread_pct": 100.0 * (max(el) - min(el)) / max(el) if max(el) > 0 else 0.0,
            "exposed_allreduce_wait_ms_per_step_max": max([v for v in per["finish_wait_gpu_ms_per_step"] if v is not None] or [None]),
            "handoff_status_or": _or_of(per["handoff_status"]),
            "note": "per_rank lists are indexed by rank; elapsed_s = that rank's own wall time over the K timed steps (`ms_per_step` is the "
                    "MAX); finish_wait_gpu = HIP-event time the compute stream spent between the end of backward and the moment clip + "
                    "Adam could start (the exposed all-reduce, incl. the 16-byte status exchange); finish_wait_host = host time inside "
                    "GradBuckets.finish(); handoff_status = the rank's sticky status word of the fused attention backward (0 = clean)"}


RCCL_LOG = None          # rank 0: file RCCL's INFO log of the communicator set-up goes to (request_rccl_init_log), parsed by comm_environment


def request_rccl_init_log():
    """Rank 0, before the process group exists: have RCCL write its INFO log of the INIT subsystem to a private file (NCCL_DEBUG_FILE), so
    that the bench line can say how many channels the communicator really got and over which transport the ring runs - not only whether
    the environment pinned them.  An NCCL_DEBUG the user set is left alone (its output then goes where the user sent it)."""
    global RCCL_LOG
    # (a bare NCCL_DEBUG=VERSION / WARN - what this image exports - prints next to nothing and is raised to INFO into the private file;
    # a user's INFO / TRACE setting or an NCCL_DEBUG_FILE of their own is left alone)
    if os.environ.get("NCCL_DEBUG", "VERSION").upper() not in ("VERSION", "WARN") or "NCCL_DEBUG_FILE" in os.environ:
        return
    import tempfile
    RCCL_LOG = os.path.join(tempfile.gettempdir(), "t2s_bench_rccl_init.%d.log" % os.getpid())
    os.environ.update(NCCL_DEBUG="INFO", NCCL_DEBUG_SUBSYS="INIT,GRAPH", NCCL_DEBUG_FILE=RCCL_LOG)


def _parse_rccl_log(path):
    """-> {"coll_channels": n, "p2p_channels": n, "lines": [the channel / ring / transport summary lines]} or None."""
    import re
    try:
        text = open(path, errors="replace").read()
    except OSError:
        return None
    keep = [ln.split("NCCL INFO", 1)[-1].strip() for ln in text.splitlines()
            if re.search(r"coll channels|Channel 00/|Connected all rings|Connected all trees|nNodes|Ring 0+ :|00/0+ .*via ", ln)]
    out = {"lines": keep[:12]}
    m = re.search(r"(\d+) coll channels.*?(\d+) p2p channels", text)
    if m:
        out["coll_channels"], out["p2p_channels"] = int(m.group(1)), int(m.group(2))
    m = re.search(r"Channel \d+/(\d+)", text)
    if m:
        out["channels_listed"] = int(m.group(1))
    # (one "Channel 00/NN" / "Ring 0" line stands for its list)
    return out


def comm_environment():
    """The communication-side settings a scaling post-mortem asks for first: RCCL version, every NCCL_* / RCCL_* / HSA_* / HIP_* /
    GPU_* / ROCR_* variable of the process, whether NCCL_MIN_NCHANNELS / NCCL_MAX_NCHANNELS pinned the channel counts, and - parsed from
    RCCL's own INIT log of this run (request_rccl_init_log) - the number of collective / p2p channels the communicator got."""
    env = {k: v for k, v in sorted(os.environ.items()) if k.split("_")[0] in ("NCCL", "RCCL", "HSA", "HIP", "GPU", "ROCR")}
    ver = None
    try:
        ver = ".".join(str(x) for x in torch.cuda.nccl.version())
    except Exception:          # a build without the binding: not fatal for a report field
        pass
    log = _parse_rccl_log(RCCL_LOG) if RCCL_LOG else None
    return {"rccl_version": ver, "env": env,
            "channels": {"NCCL_MIN_NCHANNELS": os.environ.get("NCCL_MIN_NCHANNELS"), "NCCL_MAX_NCHANNELS": os.environ.get("NCCL_MAX_NCHANNELS"),
                         "pinned": "NCCL_MIN_NCHANNELS" in os.environ or "NCCL_MAX_NCHANNELS" in os.environ,
                         "from_rccl_init_log": log, "rccl_init_log_requested_by_bench": RCCL_LOG is not None}}


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: run `python -m torch.distributed.run --nproc-per-node N bench.py <same
    flags>` as a child process (one rank per GPU over RCCL; base_trainer.py:51-71 is the reference's counterpart)."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    one_gpu = os.environ.get("T2S_BENCH_ONE_GPU") == "1"
    if have < n and not one_gpu:
        print("bench.py: --gpus %d but only %d GPU(s) are visible.  (To rehearse the multi-rank control flow on one card: "
              "T2S_BENCH_BACKEND=gloo T2S_BENCH_ONE_GPU=1 python bench.py --gpus %d ...)" % (n, have, n), file=sys.stderr)
        return 2
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] spawning %d ranks: %s" % (n, " ".join(cmd)), file=sys.stderr, flush=True)
    if os.environ.get("T2S_BENCH_DRY_SPAWN") == "1":          # tests: show the command, start nothing
        print(json.dumps({"spawn": cmd}))
        return 0
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=64, help="per-GPU batch (BASELINE config: 64)")
    ap.add_argument("--frames", type=int, default=100)
    ap.add_argument("--ocr", type=int, default=100, help="OCR tokens per frame")
    ap.add_argument("--vocab", type=int, default=5000)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--forward-only", action="store_true", help="BASELINE configs[1]: forward-only throughput")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dropout0", action="store_true", help="skip the secondary measurement with all dropout probabilities 0")
    ap.add_argument("--host-inputs", action="store_true",
                    help="also time the step with every batch staged from host memory (pinned arena, one async H2D copy per "
                         "batch overlapped with the previous step): reported as pcie_inclusive, never as value")
    ap.add_argument("--compact-wire", action="store_true",
                    help="with --host-inputs: ship the OCR tokens as 64-byte slots and build the 604-d PHOC rows "
                         "(context_feature_1) on the GPU with t2s_phoc instead of transferring them")
    ap.add_argument("--seed", type=int, default=1234,
                    help="torch.manual_seed before the first step: the dropout seeds of every step follow from it, so `loss_step0` "
                         "(the loss of the first executed step) can be recomputed from the same seed (tests/test_fullsize_gpu.py)")
    ap.add_argument("--dropout", type=float, default=0.1,
                    help="every dropout probability of the model (hidden, attention-probability, embedding, obj/ocr input). "
                         "Default 0.1 = the reference's config default, which BASELINE.md prescribes for throughput runs; "
                         "0 is the parity configuration")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # launched without a launcher: start the N ranks ourselves.  This parent has not touched the GPU (device_count() does not
        # initialise it) and never will: the ranks are FRESH child processes of torch.distributed.run, rank 0 prints the JSON
        # line to the inherited stdout, and the parent exits with the launcher's status.
        sys.exit(spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != max(1, args.gpus):
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    cpu_res = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_res = cpu_baseline(args.vocab, full=(args.frames, args.ocr))          # host-only; runs before any GPU call
        print("[bench] cpu_baseline:", json.dumps(cpu_res), file=sys.stderr, flush=True)
    import torch.distributed as dist
    # rehearsal hooks (never set by the driver): T2S_BENCH_BACKEND=gloo and T2S_BENCH_ONE_GPU=1 run the multi-rank path with
    # every rank on cuda:0, so the N > 1 control flow (buckets, barriers, MAX over ranks) can be exercised on a 1-GPU box
    backend = os.environ.get("T2S_BENCH_BACKEND", "nccl")
    if os.environ.get("T2S_BENCH_ONE_GPU") == "1":
        local_rank = 0
    # T2S_BENCH_FORCE_DIST=1 (rehearsal, never set by the driver): take the process-group path with ONE rank too, so that the
    # RCCL communicator, the bucket all-reduces, the barrier and the MAX reduction really run on a 1-GPU box
    dist_on = world > 1 or os.environ.get("T2S_BENCH_FORCE_DIST") == "1"
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            if rank == 0:
                request_rccl_init_log()
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from vitxt_gqa_amd import training_config
    from vitxt_gqa_amd.ddp import DistributedSampler, GradBuckets, reduce_dict
    from vitxt_gqa_amd.optim import build_optimizer, clip_and_step, lr_lambda_update
    from vitxt_gqa_amd.synth import make_batch, make_noise
    from vitxt_gqa_amd.testing import make_model, to_device

    B, F, P, V = args.batch, args.frames, args.ocr, args.vocab
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    # memory guard: the step keeps ~330 KB of activations per token row with the GELU / LN1 outputs stored (211 GB at B=64,
    # L=10 132 - measured); where the card has less than that free (another tenant, a smaller part), switch to the recompute
    # form (-22 %: those outputs are rebuilt in backward, two more HBM passes per layer) instead of dying in the allocator.
    # Decided per rank from its own card before the first step: the form changes no collective and no result.
    import vitxt_gqa_amd.functional as _FN
    rows = B * (T_Q + F + F * P + 3 * DEC)
    need = rows * 330e3 + 6e9
    free_b, total_b = torch.cuda.mem_get_info(dev)
    recompute = (not args.forward_only) and need > 0.97 * free_b
    if recompute:
        _FN.RECOMPUTE_ACTIVATIONS = True
        print("[bench] rank %d: %.0f GB free < %.0f GB needed: recomputing GELU / LN1 outputs in backward" % (rank, free_b / 1e9, need / 1e9),
              file=sys.stderr, flush=True)
    model = make_model(F, P, V, seed=0, dtype=dtype, dropout=args.dropout).to(dev)        # identical weights on every rank (name-seeded)
    model.train(True)          # --forward-only = the teacher-forced training forward under no_grad (not the 12-step greedy decode)
    cfg = training_config()
    opt = build_optimizer(model, cfg)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda it: lr_lambda_update(it, cfg))
    buckets = GradBuckets(model.parameters(), single_rank_collectives=dist_on)
    # each rank takes its shard of the global batch of world * B questions (weak scaling: B per GPU) the way the reference's
    # loader does: DistributedSampler, epoch-seeded permutation + contiguous chunk (samplers.py:42-60, seeded with the epoch by
    # base_trainer.py:221).  The synthetic "dataset" is B-question blocks; block g is generated from seed 100 + g.
    sampler = DistributedSampler(world, num_replicas=world, rank=rank, shuffle=True)
    sampler.set_epoch(1)
    block = int(sampler.indices()[0])
    batch = to_device(make_batch(B, F, P, V=V, seed=100 + block), dev)
    batch.grounding_noise = tuple(t.to(dev) for t in make_noise(B, F, P, seed=100 + block))
    scalar_reduces = [0]
    first_loss = []
    finish_log = None          # a list inside the timed region: (event before, event after, host seconds) of every GradBuckets.finish()
    torch.manual_seed(args.seed)        # (model and batch above are name- / block-seeded; this governs the dropout seeds of the steps)

    def step(batch=batch):
        nonlocal finish_log
        if args.forward_only:
            with torch.no_grad():
                return model.forward(batch)
        out = model(batch)
        if dist_on:
            # the logging exchange of the reference's loop (_update_meter, base_trainer.py:293-301: reduce_dict of the losses and of
            # the metrics every iteration) as ONE stacked reduce to rank 0
            reduce_dict({**out["losses"], **{"metric/" + k: v for k, v in out.get("metrics", {}).items()}})
            scalar_reduces[0] += 1 if world > 1 else 0          # reduce_dict issues no collective in a one-rank group
        loss = sum(l.mean() for l in out["losses"].values())
        if not first_loss:
            first_loss.append(loss.detach())          # kept on the device; read after the timed region
        buckets.reset()
        loss.backward()
        if dist_on and finish_log is not None:
            # the exposed part of the gradient exchange: what the compute stream (events) and the host (wall clock) spend in finish()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            h0 = time.perf_counter()
            buckets.finish()
            h1 = time.perf_counter()
            e1.record()
            finish_log.append((e0, e1, h1 - h0))
        else:
            buckets.finish()
        clip_and_step(model, opt, cfg)          # global-norm clip 0.25 + Adam (one fused multi-tensor pass)
        sched.step()
        return loss

    def sync():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    with AttnTimer() as timer:
        for _ in range(args.warmup):
            step()
        sync()
        timer.enabled = True
        sr0 = scalar_reduces[0]
        finish_log = []
        t0 = time.perf_counter()
        for _ in range(args.steps):
            last = step()
        # this rank's OWN time for its K steps (its device drained), taken before the barrier that makes every rank wait for the
        # slowest: the per-rank figures of `multi_gpu` - the contract's `elapsed` (barrier on both sides, MAX over ranks) follows
        torch.cuda.synchronize()
        own_elapsed = time.perf_counter() - t0
        sync()
        elapsed = time.perf_counter() - t0
        timer.enabled = False
        timed_finish, finish_log = finish_log, None
        scalar_reduces_per_step = (scalar_reduces[0] - sr0) / max(1, args.steps)
    collectives_per_step = buckets.launched / max(1, args.steps + args.warmup)        # gradient all-reduces launched per step
    if dist_on:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    pcie = None
    if args.host_inputs:
        from vitxt_gqa_amd.staging import ArenaLayout, BatchStager, phoc_expander
        host = make_batch(B, F, P, V=V, seed=100 + rank)
        if args.compact_wire:
            from vitxt_gqa_amd.synth import make_token_slots
            phoc_spec = {"context_feature_1": (tuple(host.pop("context_feature_1").shape), torch.float32)}
            host["ocr_token_slots"] = make_token_slots(B, F * P, seed=100 + rank)
            stager = BatchStager(ArenaLayout.from_batch(host), device=dev, depth=2, device_only=phoc_spec, post_upload=[phoc_expander()])
        else:
            stager = BatchStager(ArenaLayout.from_batch(host), device=dev, depth=2)
        n = args.warmup + args.steps
        it = stager.prefetch(host for _ in range(n))
        t0 = None
        for i, d in enumerate(it):
            if i == args.warmup:
                sync()
                t0 = time.perf_counter()
            d["dataset_name"], d["dataset_type"] = "vtextgqa", "train"
            step(d)
        sync()
        el = time.perf_counter() - t0
        if dist_on:
            t = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        pcie = {"value": world * B * args.steps / el, "unit": "samples/s", "ms_per_step": 1e3 * el / args.steps,
                "host_bytes_per_step": stager.layout.nbytes, "compact_wire": bool(args.compact_wire),
                "note": "same step, every batch copied from a pageable host batch into a pinned arena by a loader thread and "
                        "uploaded with one async H2D copy overlapped with the previous step (vitxt_gqa_amd/staging.py)"}

    nodrop = None
    if args.dropout > 0 and not args.forward_only and not args.no_dropout0:
        # secondary figure: the same step in the parity configuration (all dropout probabilities 0)
        model.set_dropout(0.0)
        step()
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        sync()
        el = time.perf_counter() - t0
        if dist_on:
            t = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        nodrop = {"value": world * B * args.steps / el, "unit": "samples/s", "ms_per_step": 1e3 * el / args.steps,
                  "note": "same step with every dropout probability 0 (the parity configuration)"}
        model.set_dropout(args.dropout)

    f_total, f_attn = flops_per_sample_fwd(F, P, V)
    mult = 1.0 if args.forward_only else 3.0
    sps = world * B * args.steps / elapsed
    att_f, att_b = timer.summary_fwd(), timer.summary_bwd()
    shape = "batch %d/GPU, %d frames x %d OCR/frame x 20 q-tokens, 12 decode steps, V=%d" % (B, F, P, V)
    if (B, F, P, V) == (64, 100, 100, 5000):
        shape += " (BASELINE.json configs[%d]%s)" % (1 if args.forward_only else 2, "; configs[3] per GPU" if world > 1 else "")
    elif (F, P) == (300, 200):
        shape += " (BASELINE.json configs[4]: long-sequence stress)"
    else:
        shape += " (not a BASELINE.json configuration)"
    res = {
        "metric": ("forward-only" if args.forward_only else "train-step") + " samples/sec (T2S, %d-frame x %d-OCR synthetic)" % (F, P),
        "value": sps, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "T2S-QA %s, %s" % ("forward" if args.forward_only else "full train step (fwd+bwd+clip+Adam, pos-BCE + 1000*InfoNCE)", shape),
                   "global_batch": world * B, "seq_len": T_Q + F + F * P + DEC, "parallelism": "dp%d" % world,
                   "dropout": args.dropout,
                   "precision": "bf16 MFMA operands, fp32 accumulate / residual stream / master weights" if args.dtype == "bf16" else "fp32"},
        "ranks_seen": dist.get_world_size() if dist_on else 1,
        "backend": (dist.get_backend() + (" (RCCL)" if dist.get_backend() == "nccl" else "")) if dist_on else None,
        "collectives_per_step": collectives_per_step, "n_buckets": len(buckets.buckets),
        "scalar_reduces_per_step": scalar_reduces_per_step,
        "sampler": "DistributedSampler(shuffle=True, epoch 1): rank %d of %d took question block %d" % (rank, world, block),
        "model_flops_per_sample": mult * f_total,
        # priced at the reference's DENSE-mask FLOPs although the pos / neg passes see <= 537 of the 10 132
        # keys: a dense-equivalent rate per GPU for comparison with the FLOP model, NOT a utilisation (roofline.* is)
        "model_tflops_dense_mask_equivalent": sps * mult * f_total / 1e12 / world,
        "attention_gemm_fraction_of_flops": f_attn / f_total,
        "peak_mem_gb": torch.cuda.max_memory_allocated(dev) / 2 ** 30, "recompute_activations": bool(recompute),
    }
    if not args.forward_only:
        from vitxt_gqa_amd import ops as _ops
        # how the fused attention backward sums dQ across key blocks, and the OR of the status words of EVERY call of the run (0: no
        # bounded spin of the hand-off ever timed out; read here, behind the timed region's synchronisation)
        from vitxt_gqa_amd import functional as _FN
        # which GEMMs of the BERT block ran on the own MFMA kernels (csrc/gemm_bf16.hip) instead of the library (T2S_OWN_GEMM)
        res["own_gemm"] = sorted(_FN.OWN_GEMM)
        _ho = (_ops.ATTN_BWD_DQ_MODE & 0xff) == 1
        res["attn_bwd_dq"] = {"mode": "ordered hand-off (bit-reproducible)" if _ho else "fp32 atomics",
                              "running_sums": ("write-through (T2S_FB_HANDOFF_SCOPE=agent)" if _ops.ATTN_BWD_DQ_MODE & 0x200 else
                                               "XCD-local: kept in the L2 of the XCD that runs the pair's key blocks, placement checked in the kernel") if _ho else None,
                              "status": _ops.fused_handoff_status()}
        res["loss"] = float(last.detach())
        # the loss of the FIRST executed step (warm-up or timed): a function of --seed, the name-seeded weights and the block-seeded
        # batch alone - tests/test_fullsize_gpu.py recomputes it at B=64
        res["loss_step0"] = float(first_loss[0]) if first_loss else None
        res["seed"] = args.seed
    # HBM traffic per launch from the PMC counters is a property of ONE configuration: profiles/traffic.json is keyed by
    # (B, F, P, dropout) and anything else reports null
    traffic, traffic_round = {}, None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        tfile = json.load(open(tpath))
        traffic, traffic_round = tfile.get("B%d_F%d_P%d_drop%g" % (B, F, P, args.dropout), {}), tfile.get("_round")
    # MFMA-busy of the attention kernels (SQ_VALU_MFMA_BUSY_CYCLES per SIMD-cycle, tools/pmc_attn.sh on the kernels alone) - a
    # property of the kernel build, read from profiles/mfma_busy.json when the dropout setting matches
    busy, busy_round = {}, None
    bpath = os.path.join(ROOT, "profiles", "mfma_busy.json")
    if os.path.exists(bpath):
        bfile = json.load(open(bpath))
        busy, busy_round = bfile.get("drop%g" % args.dropout, {}), bfile.get("_round")
    # provenance of the two figures that are NOT measured in this run (they need rocprofv3 --pmc passes): file, configuration
    # and round they were taken at, carried in the line itself
    tkey = "B%d_F%d_P%d_drop%g" % (B, F, P, args.dropout)
    traffic_source = ("profiles/traffic.json[%s]: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this bench command (%s), NOT measured "
                      "in this run" % (tkey, traffic_round or "the build the file was committed with")) if traffic else None
    busy_source = ("profiles/mfma_busy.json[drop%g]: SQ_VALU_MFMA_BUSY_CYCLES of the kernel ALONE in tools/attn_probe.py at B=8, L=10132, "
                   "70 %% of the keys visible (%s), NOT measured in this run" % (args.dropout, busy_round or "the build the file was committed with")) if busy else None
    L_seq = T_Q + F + F * P + (3 * DEC if not args.forward_only else DEC)
    alg_fwd = 4.0 * B * L_seq * HID * 2            # Q, K, V read + O written once, bf16 (dense upper bound: every key visible)
    alg_bwd = 8.0 * B * L_seq * HID * 2            # Q, K, V, O, dO read + dQ, dK, dV written once
    fwd_block = bwd_block = None
    if att_f:
        fwd_block = {"kernel": "attn_fwd_bf16_kernel (all launches with >= 1 GFLOP in the timed region)",
                     "bound": "mfma", "achieved": att_f["tflops"], "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                     "frac": att_f["tflops"] / PEAK_BF16_TFLOPS, "traffic": traffic.get("attn_fwd"),
                     "mfma_busy": (busy.get("attn_fwd_bf16_kernel") or {}).get("mfma_busy"),
                     "traffic_source": traffic_source, "mfma_busy_source": busy_source,
                     "algorithmic_bytes_per_launch": alg_fwd,
                     "traffic_over_algorithmic": (traffic.get("attn_fwd") / alg_fwd) if traffic.get("attn_fwd") else None,
                     "launches": att_f["launches"], "avg_launch_ms": att_f["avg_ms"], "ms_per_step": att_f["total_ms"] / args.steps,
                     "achieved_dense_mask_equivalent": att_f["tflops_dense"],
                     "note": "achieved = algorithmic attention-GEMM FLOPs over the VISIBLE keys (2 products: 4*12*64*L*sum_b(visible keys) "
                             "per launch; masked keys are skipped by key compaction, exact in fp32) / HIP-event time; "
                             "achieved_dense_mask_equivalent prices the same launches at the reference's dense-mask FLOPs "
                             "4*B*12*L^2*64 (SURVEY 8d) and is not a utilisation"}
    if att_b:
        bwd_block = {"kernel": "attention backward launch group (attn_delta + attn_dkdv_bf16_kernel + attn_dq_bf16_kernel, or attn_delta_prep + "
                               "attn_bwd_fused_bf16_kernel + attn_dq_cast for the launches that take the fused 5-product form; all launches "
                               "with >= 1 GFLOP in the timed region)",
                     "bound": "mfma", "achieved": att_b["tflops"], "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                     "frac": att_b["tflops"] / PEAK_BF16_TFLOPS, "traffic": traffic.get("attn_bwd"),
                     "mfma_busy": (busy.get("attn_bwd_fused_bf16_kernel") or {}).get("mfma_busy"),
                     "traffic_source": traffic_source, "mfma_busy_source": busy_source,
                     "algorithmic_bytes_per_launch": alg_bwd,
                     "traffic_over_algorithmic": (traffic.get("attn_bwd") / alg_bwd) if traffic.get("attn_bwd") else None,
                     "launches": att_b["launches"], "fused_5_product_launches": att_b["fused_launches"], "avg_launch_ms": att_b["avg_ms"],
                     "fused_avg_launch_ms": att_b["fused_avg_ms"], "fused_achieved": att_b["fused_tflops"],
                     "long_list_launches": att_b["long_launches"], "long_list_avg_launch_ms": att_b["long_avg_ms"],
                     "ms_per_step": att_b["total_ms"] / args.steps, "achieved_executed": att_b["tflops_executed"],
                     "note": "achieved = ALGORITHMIC backward FLOPs over the visible keys (5 products: 10*12*64*L*sum_b(visible keys) per "
                             "launch) / HIP-event time around ops.attn_bwd; achieved_executed counts the products the kernels really "
                             "run (7 per (query, key) pair in the two-kernel form, which recomputes S and dP; 5 in the fused form); "
                             "fused_avg_launch_ms / fused_achieved: the launches that took the fused form alone (their group in a rocprofv3 "
                             "kernel trace: attn_delta_prep + attn_bwd_fused<.,3,.>; with the dQ hand-off every launch of a long sequence, "
                             "the short lists of the pos / neg passes included); long_list_avg_launch_ms: the launches whose lists average "
                             ">= 2 048 keys (the ref pass and QTV: the 45 ms launches); "
                             "mfma_busy: SQ_VALU_MFMA_BUSY_CYCLES per SIMD-cycle of the fused kernel alone (profiles/mfma_busy.json); "
                             "traffic_over_algorithmic: PMC bytes per launch / (Q, K, V, O, dO read + dQ, dK, dV written once) - the "
                             "excess is the hand-off's running fp32 dQ sums: they stay in the XCD's L2 between key blocks (write-back "
                             "stores) but are still written back several times per launch (DESIGN section 5, profiles/r05_handoff_scope.txt)"}
    # `roofline` = the dominant kernel group of the step (the attention backward in a train step, the forward otherwise)
    if bwd_block is not None:
        res["roofline"], res["roofline_fwd"] = bwd_block, fwd_block
    elif fwd_block is not None:
        res["roofline"] = fwd_block
    if dist_on:
        from vitxt_gqa_amd import ops as _ops
        k = max(1, args.steps)
        local = {"elapsed_s": own_elapsed,
                 "finish_wait_gpu_ms_per_step": sum(a.elapsed_time(b) for a, b, _ in timed_finish) / k,
                 "finish_wait_host_ms_per_step": 1e3 * sum(h for _, _, h in timed_finish) / k,
                 "handoff_status": _ops.fused_handoff_status() if not args.forward_only else 0,
                 "fused_launches": _ops.fused_launches_seen(dev) if not args.forward_only else 0,
                 "attn_bwd_ms_per_step": (att_b["total_ms"] / k) if att_b else None,
                 "attn_fwd_ms_per_step": (att_f["total_ms"] / k) if att_f else None,
                 "peak_mem_gb": torch.cuda.max_memory_allocated(dev) / 2 ** 30}
        rep = gather_rank_report(local, device=dev if backend == "nccl" else None)
        rep["comm"] = comm_environment()
        rep["buckets_mb"] = [round(f.numel() * 4 / 2 ** 20, 1) for f, _ in buckets.buckets]
        res["multi_gpu"] = rep
    if cpu_res is not None:
        res["cpu_baseline"] = cpu_res
    if pcie is not None:
        res["pcie_inclusive"] = pcie
    if nodrop is not None:
        res["dropout_0"] = nodrop
    if rank == 0:
        print(json.dumps(res), flush=True)
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
