"""Builds libt2s_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m vitxt_gqa_amd.build [--force]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libt2s_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -fno-slp-vectorize: under plain -O3 the SLP vectoriser packs adjacent scalar f32 multiplies / adds of the attention kernels
# into v_pk_mul_f32 / v_pk_fma_f32, which issue slower beside MFMAs than the scalar forms (MI355X_MICROARCH.md, per-instruction
# constants); measured on the attention backward: 6.10 -> 5.87 ms without dropout, 7.95 -> 7.76 ms with (B=8, same box)
FLAGS = ["--offload-arch=gfx950", "-O3", "-fno-slp-vectorize", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function"]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = sources() + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]
    deps.append(os.path.join(os.path.dirname(HERE), "include", "t2s_hip.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not _stale():
        return OUT
    cmd = [HIPCC] + FLAGS + ["-o", OUT] + sources()
    if verbose:
        print("[vitxt_gqa_amd.build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
