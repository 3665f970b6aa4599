#!/usr/bin/env python3
"""Generates the stamped DIAGNOSTIC source of the 256-key INTERLEAVED fused attention backward (tools/ablate/attn_bwd_fused_bf16_ilv256.hip):
cycle stamps (s_memtime) around the segments of the interleaved sweep's tile loop, summed per workgroup by wave 0 and written to a 16 KB
tail added to the workspace.  The variant source carries no diagnostics; this script patches a copy at build time.
    python tools/ablate/make_fb_diag.py OUT.hip
Segments: 0 slot 0 (G1(b0) + the 16 dQ MFMAs of the previous tile) | 1 slots 1..3 + the dQ hand-off (fma, stores) |
2 slots 4..7 | 3 K^T preload + vmcnt(0) + flag wait | 4 stage write + sum loads + barrier + flag store | 5 preload behind the barrier."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# the source to patch: "ilv256" = the 256-key interleaved variant, "ilv384" = the 384-key kernel with the dQ product inside slots 0 and 1 of
# the next tile's phase A (tools/ablate/variants/attn_bwd_fused_bf16_ilv384.hip); both measured against the shipped serial kernel in round 5
WHICH = sys.argv[1] if len(sys.argv) > 2 else "ilv256"
OUT = sys.argv[-1]
assert len(sys.argv) >= 2 and OUT.endswith(".hip"), "usage: make_fb_diag.py [ilv256|ilv384] OUT.hip"
# "product" = the shipped kernel (since the interleaved 384-key form took over: the ilv384 schedule with XCD-local running sums)
# a path to a .hip file = that source, stamped like the product (round 6: timing-only ablations of the shipped sweep, tools/ablate/make_fb_ablation.py)
src = open(WHICH if WHICH.endswith(".hip") else
           os.path.join(ROOT, "tools", "ablate", "attn_bwd_fused_bf16_ilv256.hip") if WHICH == "ilv256" else
           os.path.join(ROOT, "vitxt_gqa_amd", "csrc", "attn_bwd_fused_bf16.hip") if WHICH == "product" else
           os.path.join(ROOT, "tools", "ablate", "variants", "attn_bwd_fused_bf16_ilv384.hip")).read()


def rep(a, b):
    global src
    assert src.count(a) == 1, (src.count(a), a[:80])
    src = src.replace(a, b)


rep('#include "attn_common.h"', '#include "%s"' % os.path.join(ROOT, "vitxt_gqa_amd", "csrc", "attn_common.h"))
rep("  int handoff;\n", "  int handoff;\n  unsigned long long* dbg;      // DIAGNOSTIC: 256 x 8 words behind the workspace\n")
rep("  return ctrl + rowc + sums;\n", "  return ctrl + rowc + sums + 16384;\n")
rep("  w.handoff = handoff;\n", "  w.handoff = handoff;\n  w.dbg = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(workspace) + attn_bwd_fused_workspace_bytes(p.B, p.H, p.Lq) - 16384);\n"
    "  if (hipMemsetAsync(w.dbg, 0, 16384, st) != hipSuccess) return 3;\n")
TICK = '{ FB_FENCE(); asm volatile("s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(st_t1) :: "memory"); st_sum[K] += st_t1 - st_t0; st_t0 = st_t1; FB_FENCE(); }\n'


def tick(k):
    return TICK.replace("K", str(k))
rep("    for (int qt = 0; qt < nqt; ++qt) {\n      const int buf = qt & 1;\n",
    "    unsigned long long st_sum[6] = {0, 0, 0, 0, 0, 0}, st_t0 = 0, st_t1 = 0;\n"
    '    asm volatile("s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(st_t0) :: "memory");\n'
    "    const unsigned long long st_c0 = st_t0, st_r0 = __builtin_amdgcn_s_memrealtime();\n"
    "    for (int qt = 0; qt < nqt; ++qt) {\n      const int buf = qt & 1;\n")
if WHICH == "ilv256":
    rep("        FB_LD_KF(1); FB_FENCE();\n        FB_SLOT_G1E(1, 0);", "        " + tick(0) + "        FB_LD_KF(1); FB_FENCE();\n        FB_SLOT_G1E(1, 0);")
    rep("          if (qt > 0) { FB_DQ_FINALIZE(qt - 1); }\n", "          if (qt > 0) { FB_DQ_FINALIZE(qt - 1); }\n          " + tick(1))
else:       # ilv384 (384 keys, dQ product in slots 0 and 1): segment 0 = slot 0, 1 = slot 1 + hand-off, 2 = slots 2..11
    rep("          FB_LD_KF(1); FB_FENCE();\n          FB_THR(0); FB_LD_RK(0);", "          " + tick(0) + "          FB_LD_KF(1); FB_FENCE();\n          FB_THR(0); FB_LD_RK(0);")
    rep("          else if (qt > 0) { FB_DQ_FINALIZE(qt - 1); }\n", "          else if (qt > 0) { FB_DQ_FINALIZE(qt - 1); }\n          " + tick(1))
rep("      // Stage the next tile here, at the end of phase A: its DMA pieces", "      " + tick(2) + "      // Stage the next tile here, at the end of phase A: its DMA pieces")
rep("      FB_STAGE_WRITE(buf ^ 1);\n", "      " + tick(3) + "      FB_STAGE_WRITE(buf ^ 1);\n")
a4 = "      if constexpr (ILV) {\n        // the dQ product of this tile runs in slot 0 of the next one" if WHICH == "ilv256" else "      if constexpr (ILV) {\n        // the dQ product of this tile runs in slots 0 and 1 of the next one"
rep(a4, "      " + tick(4) + a4)
rep("        FB_LD_KF(0);\n        FB_FENCE();\n      } else {", "        FB_LD_KF(0);\n        FB_FENCE();\n        " + tick(5) + "      } else {")
rep("    };\n    if constexpr (MODE == 3) {",
    "    if (ILV && lane == 0 && wave == 0) {\n"
    "      unsigned long long* dbg = w.dbg + (int64_t)(blockIdx.x & 255) * 8;\n"
    "      for (int k = 0; k < 6; ++k) dbg[k] = st_sum[k];\n"
    "      dbg[6] = (unsigned long long)nqt;\n"
    "      dbg[7] = ((st_t0 - st_c0) * 1000ull) / (__builtin_amdgcn_s_memrealtime() - st_r0 + 1);\n"
    "    }\n"
    "    };\n    if constexpr (MODE == 3) {")
src = "// GENERATED by tools/ablate/make_fb_diag.py from tools/ablate/attn_bwd_fused_bf16_ilv256.hip: DIAGNOSTIC build (cycle stamps), never shipped.\n" + src
open(OUT, "w").write(src)
