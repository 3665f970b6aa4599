"""Library-GEMM algorithm selection for the benchmark shapes.

The encoder's linear layers are plain library GEMMs (hipBLASLt through torch).  For each (transpose, M, N, K, leading dims) the
library's default heuristic is not always its fastest solution; PyTorch's TunableOp can time the candidates once and replay the
winners.  ``tuned/gemm_gfx950_b64_100x100.csv`` holds the winners for the GEMM shapes of the headline workload (B=64, 100 frames x
100 OCR tokens, bf16), recorded on an MI355X with this image's libraries by ``tools/tune_gemms.sh``; 22 of its 61 shapes stay on
the default.  Loading it is read-only: no tuning happens at run time, shapes that are not in the file take the library default, and
a file recorded with other library versions is ignored by torch's validators.  Worth 0.7 % of the benchmark step (same-box A/B
731.9 -> 726.9 ms).  ``T2S_TUNED_GEMMS=0`` disables it."""
import os

import torch

_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned", "gemm_gfx950_b64_100x100.csv")
_state = {"done": False, "on": False}


def enable_tuned_gemms(path=_FILE):
    """Idempotent.  Returns True when the recorded selections are active in this process."""
    if _state["done"]:
        return _state["on"]
    _state["done"] = True
    if os.environ.get("T2S_TUNED_GEMMS", "1") == "0" or not torch.cuda.is_available() or not os.path.exists(path):
        return False
    if os.environ.get("PYTORCH_TUNABLEOP_ENABLED") is not None:
        return False                      # the caller drives TunableOp itself (tools/tune_gemms.sh records a new file that way)
    try:
        import torch.cuda.tunable as tunable
        tunable.enable(True)
        tunable.tuning_enable(False)      # replay only
        if hasattr(tunable, "write_file_on_exit"):
            tunable.write_file_on_exit(False)
        _state["on"] = bool(tunable.read_file(path))
        if not _state["on"]:
            tunable.enable(False)
    except Exception:                     # an older torch without the module: the library defaults stay
        _state["on"] = False
    return _state["on"]
