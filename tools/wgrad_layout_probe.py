"""Batched weight-gradient GEMM (functional._wgrad): dW[out, in] = sum over row groups of dy_g^T x_g, as a function of the number of
groups G the B * L rows are cut into (G = B = 64: one partial product per sample, the shipped form; G = 1: one GEMM, the library's
split-K); random bf16 operands, interleaved rounds in one process."""
import torch
B, L = 64, 10156
dev = "cuda"
def timeit(fns, n=5, rounds=3):
    res = {k: [] for k in fns}
    for f in fns.values():
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
for N, K in ((3072, 768), (768, 3072), (2304, 768), (768, 768)):      # dy [B*L, N], x [B*L, K], dW [N, K]
    dy = torch.randn(B * L, N, device=dev, dtype=torch.bfloat16)
    x = torch.randn(B * L, K, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * B * L * K * N
    fns = {}
    for G in (256, 128, 64, 32, 16, 8, 4):
        if (B * L) % G:
            continue
        fns["G = %3d" % G] = (lambda G=G: torch.bmm(dy.view(G, B * L // G, N).transpose(1, 2), x.view(G, B * L // G, K)).sum(0, dtype=torch.float32))
    fns["G =   1 (dy^T @ x)"] = lambda: (dy.t() @ x).float()
    # round 4 (the Cijk_Ailk_Bjlk MT256x256x32 family IS this batched wgrad of the two FFN weights): the transposed product
    # dW^T[in, out] = x_g^T dy_g - the same operands with the roles of the library's A / B swapped (another tile selection); the caller
    # would take the [in, out] fp32 sum transposed (2.4 M elements: free)
    for G in (64, 16):
        fns["G = %3d, dW^T = x^T dy" % G] = (lambda G=G: torch.bmm(x.view(G, B * L // G, K).transpose(1, 2), dy.view(G, B * L // G, N)).sum(0, dtype=torch.float32))
    r = timeit(fns)
    print("dW[%d, %d], rows = %d" % (N, K, B * L))
    for k, v in r.items():
        print("   %-26s %7.3f ms %7.1f TFLOP/s" % (k, v, fl / v / 1e9))
