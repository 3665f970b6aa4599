import os, sys, glob
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from vitxt_gqa_amd.gemm_tuning import enable_tuned_gemms
os.chdir("/tmp")
before = set(glob.glob("/tmp/*.csv"))
print("enabled:", enable_tuned_gemms())
import torch.cuda.tunable as t
print("is_enabled", t.is_enabled(), "tuning", t.tuning_is_enabled(), "n results", len(t.get_results()), "filename", t.get_filename())
a = torch.randn(647680, 768, device="cuda", dtype=torch.bfloat16); w = torch.randn(2304, 768, device="cuda", dtype=torch.bfloat16); b = torch.randn(2304, device="cuda", dtype=torch.bfloat16)
for _ in range(3): y = torch.addmm(b, a, w.t())
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): y = torch.addmm(b, a, w.t())
e1.record(); torch.cuda.synchronize()
print("addmm ms", e0.elapsed_time(e1) / 10)
import atexit
atexit.register(lambda: print("new csv files in cwd:", set(glob.glob("/tmp/*.csv")) - before))
