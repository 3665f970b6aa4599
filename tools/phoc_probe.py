"""t2s_phoc rate against its HBM-write roofline, next to the reference's C extension on one host core.
usage: python tools/phoc_probe.py [n_tokens] [width]"""
import sys, time, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vitxt_gqa_amd import ops
from vitxt_gqa_amd.synth import make_token_slots
n = int(sys.argv[1]) if len(sys.argv) > 1 else 640000
width = int(sys.argv[2]) if len(sys.argv) > 2 else 64
slots = make_token_slots(1, n, width=width, seed=1)[0]
dev = slots.cuda()
out = torch.empty(n, 604, device="cuda")
ops.phoc(dev, out=out); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): ops.phoc(dev, out=out)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
byts = n * (604 * 4 + width)
print("t2s_phoc: %d tokens x %d B -> %.3f ms, %.1f M tokens/s, %.2f TB/s of 8 TB/s HBM (algorithmic %d B/token)" % (n, width, ms, n / ms / 1e3, byts / ms / 1e9, 604 * 4 + width))
from oracle import phoc_oracle as po
ref = po.reference_build_phoc_raw()
words = [bytes(r[r != 0]).decode() for r in slots[:20000].numpy()]
t = time.perf_counter()
if ref is not None:
    for w in words: np.array(ref(w), dtype=np.float32)
    kind = "reference extension (cphoc.c from oracle/_ref) + np.array, as build_phoc.py does"
else:
    for w in words: po.build_phoc_raw(w)
    kind = "oracle port"
dt = time.perf_counter() - t
print("cpu baseline, 1 core, %s: %.3f M tokens/s" % (kind, len(words) / dt / 1e6))
