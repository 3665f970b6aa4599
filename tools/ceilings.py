#!/usr/bin/env python3
"""Measured ceilings of the box ("record a measured GEMM / stream-copy ceiling next to the vendor peaks"):
library bf16 GEMM rate on a square problem and on the step's own shapes, and the HBM rate of a copy and of a read-only pass.
usage: python tools/ceilings.py   -> one JSON line"""
import json
import time

import torch

dev = "cuda:0"


def timed(fn, n=10):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n


res = {"gemm_tflops": {}, "hbm_tbps": {}}
for name, (M, N, K) in {"square 8192^3": (8192, 8192, 8192), "QKV fwd  M=648448 N=2304 K=768": (648448, 2304, 768),
                        "FFN-in fwd  M=648448 N=3072 K=768": (648448, 3072, 768), "FFN-out fwd  M=648448 N=768 K=3072": (648448, 768, 3072)}.items():
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    dt = timed(lambda: a @ w.t(), 5)
    res["gemm_tflops"][name] = 2.0 * M * N * K / dt / 1e12
    del a, w
x = torch.empty(1 << 30, device=dev, dtype=torch.float32)          # 4 GiB
y = torch.empty_like(x)
x.normal_()
dt = timed(lambda: y.copy_(x))
res["hbm_tbps"]["copy 4 GiB (read + write)"] = 2.0 * x.numel() * 4 / dt / 1e12
dt = timed(lambda: x.sum())
res["hbm_tbps"]["read-only pass over 4 GiB (sum)"] = x.numel() * 4 / dt / 1e12
dt = timed(lambda: y.zero_())
res["hbm_tbps"]["fill 4 GiB (write only)"] = x.numel() * 4 / dt / 1e12
res["vendor_peaks"] = {"bf16_dense_tflops": 2500.0, "hbm_tbps": 8.0}
print(json.dumps(res))
