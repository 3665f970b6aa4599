"""Round 5: the own 8-phase MFMA GEMM family against the library on the train step's shapes (interleaved rounds in one process,
random operands, median / min; cdna_hip_programming.md rules 24, 25).  Output: profiles/r05_gemm_probe.txt."""
import sys
import os
import statistics

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vitxt_gqa_amd import gemm as G, ops  # noqa: E402

dev = "cuda"
M = int(os.environ.get("PROBE_M", "649984"))
ROUNDS, REPS = 4, 5


def timeit(fns):
    res = {k: [] for k in fns}
    for k, f in fns.items():
        f()
    torch.cuda.synchronize()
    for _ in range(ROUNDS):
        for k, f in fns.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(REPS):
                f()
            e1.record()
            torch.cuda.synchronize()
            res[k].append(e0.elapsed_time(e1) / REPS)
    return {k: (statistics.median(v), min(v)) for k, v in res.items()}


def rnd(*shape):
    return (torch.rand(*shape, device=dev) * 2 - 1).to(torch.bfloat16)


def report(title, flops, r):
    print(title)
    for k, (med, mn) in r.items():
        print("   %-44s median %7.3f ms  min %7.3f ms   %6.0f TFLOP/s" % (k, med, mn, flops / med / 1e9))
    sys.stdout.flush()


for (N, K) in [(768, 3072), (3072, 768), (2304, 768), (768, 768), (768, 2304)]:
    a, w = rnd(M, K), (rnd(N, K).float() * 0.05).to(torch.bfloat16)
    bias = rnd(N)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    r = timeit({"own gemm_nt + bias": lambda: G.gemm_nt(a, w, bias, out=out),
                "library addmm": lambda: torch.addmm(bias, a, w.t(), out=out)})
    report("NT  M=%d N=%d K=%d" % (M, N, K), 2.0 * M * N * K, r)
    del a, w, out

# FFN-out dgrad + GELU backward: own fused vs library GEMM + gelu_bwd
N, K = 3072, 768
dy, w, u = rnd(M, K), (rnd(N, K).float() * 0.05).to(torch.bfloat16), (torch.randn(M, N, device=dev) * 1.5).to(torch.bfloat16)
r = timeit({"own gemm_nt + gelu' epilogue (du, db)": lambda: G.gemm_nt_gelu_grad(dy, w, u),
            "library mm + gelu_bwd": lambda: ops.gelu_bwd(dy @ w.t(), u)})
report("dgrad FFN-out + GELU backward  M=%d N=%d K=%d" % (M, N, K), 2.0 * M * N * K, r)
bias = rnd(N)
r = timeit({"own gemm_nt dual epilogue (u, gelu(u))": lambda: G.gemm_nt_gelu_dual(dy, w, bias),
            "library addmm + gelu_fwd": lambda: ops.gelu_fwd(torch.addmm(bias, dy, w.t()))})
report("FFN-in forward + GELU  M=%d N=%d K=%d" % (M, N, K), 2.0 * M * N * K, r)
del dy, w, u

# weight gradients
B = 64
for (n_out, n_in) in [(768, 3072), (3072, 768), (2304, 768), (768, 768)]:
    dy, x = rnd(M, n_out), rnd(M, n_in)
    G_ = 16 if n_out * n_in == 3072 * 768 else B

    def lib():
        part = torch.bmm(dy.view(G_, M // G_, -1).transpose(1, 2), x.view(G_, M // G_, -1))
        return part.sum(0, dtype=torch.float32)
    r = timeit({"own gemm_wgrad (split-K slabs + ordered sum)": lambda: G.gemm_wgrad(dy, x),
                "library bmm over row groups + fp32 sum": lib})
    report("wgrad rows=%d n_out=%d n_in=%d (splits %d)" % (M, n_out, n_in, G.X.lib().t2s_gemm_wgrad_splits(M, n_out, n_in)), 2.0 * M * n_out * n_in, r)
    del dy, x
