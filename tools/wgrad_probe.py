"""wgrad variants: dW[N,K] = dy[M,N]^T x[M,K], M = B*L.  One GEMM (split-K left to the library) vs batched per sample + sum."""
import time, torch
B, L = 64, 10132
M = B * L
dev = "cuda"
def bench(fn, flops, name, n=5):
    r = fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / n
    print("%-52s %8.3f ms %8.1f TF/s" % (name, dt * 1e3, flops / dt / 1e12))
    return r
for K, N in ((768, 2304), (768, 768), (768, 3072), (3072, 768)):
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    dy = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * M * K * N
    r0 = bench(lambda: dy.t() @ x, fl, "N=%d K=%d one GEMM dy^T @ x" % (N, K))
    xb, dyb = x.view(B, L, K), dy.view(B, L, N)
    r1 = bench(lambda: torch.bmm(dyb.transpose(1, 2), xb).sum(0), fl, "   bmm per sample (64) + sum")
    for G in (8, 16, 32):
        xg, dyg = x.view(G, M // G, K), dy.view(G, M // G, N)
        bench(lambda: torch.bmm(dyg.transpose(1, 2), xg).float().sum(0), fl, "   bmm %d groups + fp32 sum" % G)
    xg, dyg = x.view(16, M // 16, K), dy.view(16, M // 16, N)
    bench(lambda: torch.bmm(xg.transpose(1, 2), dyg).float().sum(0), fl, "   bmm 16 groups x^T@dy + fp32 sum")
    print("   max diff bmm vs one:", (r0.float() - r1.float()).abs().max().item(), "scale", r0.float().abs().max().item())
