"""Runs the own GEMM kernels (and the library on the same shapes) a few times: the program rocprofv3 --pmc profiles in tools/pmc_gemm.sh."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vitxt_gqa_amd import gemm as G  # noqa: E402

M = 649984
dev = "cuda"


def rnd(*s):
    return (torch.rand(*s, device=dev) * 2 - 1).to(torch.bfloat16)


a, w, b = rnd(M, 3072), (rnd(768, 3072).float() * 0.05).to(torch.bfloat16), rnd(768)
out = torch.empty(M, 768, dtype=torch.bfloat16, device=dev)
for _ in range(3):
    G.gemm_nt(a, w, b, out=out)
    torch.addmm(b, a, w.t(), out=out)
del a, out
a, w, b = rnd(M, 768), (rnd(3072, 768).float() * 0.05).to(torch.bfloat16), rnd(3072)
out = torch.empty(M, 3072, dtype=torch.bfloat16, device=dev)
for _ in range(3):
    G.gemm_nt(a, w, b, out=out)
del out
dy, x = rnd(M, 768), a
for _ in range(3):
    G.gemm_wgrad(dy, x)
torch.cuda.synchronize()
