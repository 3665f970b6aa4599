"""Library GEMM rates, NN vs NT operand layout, for the dgrad shapes of the BERT layers (M = B*L rows, random bf16 operands, interleaved
rounds in one process): dy[M, N] @ W[N, K]  (W as stored by nn.Linear: "NN")  vs  dy @ Wt.t() with Wt = W.t().contiguous() [K, N] ("NT"),
plain and as the in-place accumulate addmm_ the backward uses; plus the forward addmm with W vs a pre-transposed copy."""
import sys, torch
M = int(sys.argv[1]) if len(sys.argv) > 1 else 64 * 10156
dev = "cuda"
def timeit(fns, n=6, rounds=3):
    res = {k: [] for k in fns}
    for k, f in fns.items():
        f()
    torch.cuda.synchronize()
    for _ in range(rounds):
        for k, f in fns.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                f()
            e1.record()
            torch.cuda.synchronize()
            res[k].append(e0.elapsed_time(e1) / n)
    return {k: min(v) for k, v in res.items()}
for N, K in ((3072, 768), (768, 3072), (2304, 768), (768, 768)):      # dy is [M, N], W is [N, K] (nn.Linear(K -> N))
    dy = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    wt = w.t().contiguous()
    b = torch.randn(N, device=dev, dtype=torch.bfloat16)
    acc = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * M * K * N
    r = timeit({"dgrad NN  dy @ W": lambda: dy @ w, "dgrad NT  dy @ Wt.t()": lambda: dy @ wt.t(),
                "dgrad NN  acc.addmm_(dy, W)": lambda: acc.addmm_(dy, w), "dgrad NT  acc.addmm_(dy, Wt.t())": lambda: acc.addmm_(dy, wt.t()),
                "fwd   NT  addmm(b, x, W.t())": lambda: torch.addmm(b, x, w.t()), "fwd   NN  addmm(b, x, Wt)": lambda: torch.addmm(b, x, wt)})
    print("Linear(%d -> %d), M = %d" % (K, N, M))
    for k, v in r.items():
        print("   %-36s %7.3f ms %7.1f TFLOP/s" % (k, v, fl / v / 1e9))
