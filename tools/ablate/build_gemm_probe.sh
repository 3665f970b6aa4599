#!/bin/bash
# Builds the own-GEMM prototype (tools/ablate/gemm_bf16.hip: 256 x 256 x 64 tile, LDS-DMA staging, bias + exact-GELU epilogue through a
# bf16-indexed table) into tools/ablate/_build/libgemm_probe.so for tools/gemm_epilogue_probe.py.  Not part of libt2s_hip.so.
set -e
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/tools/ablate/_build
mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -fPIC -shared -Wno-unused-function \
  -o $out/libgemm_probe.so $root/tools/ablate/gemm_bf16.hip $root/vitxt_gqa_amd/csrc/abi.cpp
ls -la $out/libgemm_probe.so
