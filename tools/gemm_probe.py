"""Library GEMM rates for the shapes of the T2S BERT layers (M = B*L rows): forward x@W^T, dgrad dy@W, wgrad dy^T@x."""
import sys, time, torch
M = int(sys.argv[1]) if len(sys.argv) > 1 else 64 * 10132
dev = "cuda"
def bench(fn, flops, name, n=5):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / n
    print("%-46s %8.3f ms %8.1f TF/s" % (name, dt * 1e3, flops / dt / 1e12))
for K, N in ((768, 2304), (768, 768), (768, 3072), (3072, 768)):
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    b = torch.randn(N, device=dev, dtype=torch.bfloat16)
    dy = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
    wt = w.t().contiguous()
    fl = 2.0 * M * K * N
    bench(lambda: torch.nn.functional.linear(x, w, b), fl, "fwd  linear(x[M,%d], W[%d,%d])+b" % (K, N, K))
    bench(lambda: torch.addmm(b, x, wt), fl, "fwd  addmm(b, x, Wt[%d,%d])" % (K, N))
    bench(lambda: dy @ w, fl, "dgrad dy[M,%d] @ W[%d,%d]" % (N, N, K))
    bench(lambda: dy.t() @ x, fl, "wgrad dy^T[%d,M] @ x[M,%d]" % (N, K))
    bench(lambda: (x.t() @ dy), fl, "wgrad x^T[%d,M] @ dy[M,%d]" % (K, N))
    bench(lambda: dy.sum(0), 0.0, "bias grad dy.sum(0)")
