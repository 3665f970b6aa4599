"""The NT GEMM's work map: N-tiles per group (T2S_GEMM_NT_GROUP, read by t2s_gemm_nt on every call) on the step's NT shapes, interleaved
rounds in one process, random operands.  b = tiles_n is the round-5 map (all N-tiles of an M-block side by side).
Output: profiles/r05_gemm_nt_map.txt.   With PROBE_PMC=1: one call per form only (for the rocprofv3 --pmc FETCH_SIZE pass)."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vitxt_gqa_amd import gemm as G  # noqa: E402

dev = "cuda"
M = int(os.environ.get("PROBE_M", "649984"))
PMC = os.environ.get("PROBE_PMC", "0") == "1"
ROUNDS, REPS = (1, 1) if PMC else (4, 5)


def rnd(*shape):
    return (torch.rand(*shape, device=dev) * 2 - 1).to(torch.bfloat16)


def run(title, flops, call, widths):
    res = {b: [] for b in widths}
    for b in widths:
        os.environ["T2S_GEMM_NT_GROUP"] = str(b)
        call()
    torch.cuda.synchronize()
    for _ in range(ROUNDS):
        for b in widths:
            os.environ["T2S_GEMM_NT_GROUP"] = str(b)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(REPS):
                call()
            e1.record()
            torch.cuda.synchronize()
            res[b].append(e0.elapsed_time(e1) / REPS)
    print(title)
    for b in widths:
        med = statistics.median(res[b])
        print("   N-tiles per group %2d   median %7.3f ms  min %7.3f ms   %6.0f TFLOP/s" % (b, med, min(res[b]), flops / med / 1e9))
    sys.stdout.flush()


N, K = 3072, 768
dy, w, u = rnd(M, K), (rnd(N, K).float() * 0.05).to(torch.bfloat16), (torch.randn(M, N, device=dev) * 1.5).to(torch.bfloat16)
bias = rnd(N)
W12 = [12, 6, 4, 3, 2, 1]
run("FFN-in forward + GELU (dual epilogue)  M=%d N=%d K=%d" % (M, N, K), 2.0 * M * N * K, lambda: G.gemm_nt_gelu_dual(dy, w, bias), W12)
run("dgrad FFN-out + GELU' epilogue  M=%d N=%d K=%d" % (M, N, K), 2.0 * M * N * K, lambda: G.gemm_nt_gelu_grad(dy, w, u), W12)
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
run("plain NT + bias  M=%d N=%d K=%d" % (M, N, K), 2.0 * M * N * K, lambda: G.gemm_nt(dy, w, bias, out=out), W12)
del dy, w, u, out
N, K = 768, 3072
a, w = rnd(M, K), (rnd(N, K).float() * 0.05).to(torch.bfloat16)
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
bias = rnd(N)
run("plain NT + bias  M=%d N=%d K=%d" % (M, N, K), 2.0 * M * N * K, lambda: G.gemm_nt(a, w, bias, out=out), [3, 2, 1])
