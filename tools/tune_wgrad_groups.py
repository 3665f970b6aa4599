#!/usr/bin/env python3
"""TunableOp over the batched weight-gradient GEMMs that round 4 regrouped (16 row groups for the two FFN weights: shapes the recorded
file of round 3 does not hold): loads vitxt_gqa_amd/tuned/gemm_gfx950_b64_100x100.csv, lets torch time the library's solutions for the
new shapes only, prints default vs selected times and writes the merged file to gpurun_out/ (copy it over the recorded file by hand).
usage (GPU box): python tools/tune_wgrad_groups.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PYTORCH_TUNABLEOP_ENABLED"] = "1"          # (keeps vitxt_gqa_amd.gemm_tuning from touching TunableOp itself)
import torch  # noqa: E402
import torch.cuda.tunable as tunable  # noqa: E402

src = os.path.join(ROOT, "vitxt_gqa_amd", "tuned", "gemm_gfx950_b64_100x100.csv")
dst = os.path.join(ROOT, "gpurun_out", "gemm_gfx950_b64_100x100_with_g16.csv")
os.makedirs(os.path.dirname(dst), exist_ok=True)
dev = torch.device("cuda", 0)
G = 16
shapes = [(rows, o, i) for rows in (64 * 10156, 64 * 10120) for (o, i) in ((3072, 768), (768, 3072))]


def run(dy, x, n=8):
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(n):
        a.record()
        torch.bmm(dy.view(G, dy.size(0) // G, -1).transpose(1, 2), x.view(G, x.size(0) // G, -1))
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]


tens = {}
for rows, o, i in shapes:
    tens[(rows, o, i)] = (torch.randn(rows, o, device=dev).to(torch.bfloat16), torch.randn(rows, i, device=dev).to(torch.bfloat16))
tunable.enable(False)
base = {k: run(*v) for k, v in tens.items()}
tunable.enable(True)
tunable.tuning_enable(True)
tunable.set_max_tuning_duration(60)
tunable.set_max_tuning_iterations(50)
if hasattr(tunable, "write_file_on_exit"):
    tunable.write_file_on_exit(False)
ok = tunable.read_file(src)
print("recorded file loaded:", ok)
t0 = time.time()
for k, v in tens.items():
    run(*v, n=1)                                     # first call of a new shape: TunableOp times the candidates
print("tuning took %.0f s" % (time.time() - t0))
tunable.tuning_enable(False)
for k, v in tens.items():
    t = run(*v)
    fl = 2.0 * k[0] * k[1] * k[2]
    print("rows %d dW[%d, %d] G=%d: default %.3f ms (%.0f TFLOP/s) -> selected %.3f ms (%.0f TFLOP/s)" % (k[0], k[1], k[2], G, base[k], fl / base[k] / 1e9, t, fl / t / 1e9))
tunable.write_file(dst)
new = [l for l in open(dst) if "_B_16_" in l]
print("new entries:")
for l in new:
    print("  " + l.strip())
