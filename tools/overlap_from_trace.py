#!/usr/bin/env python3
"""Do the gradient all-reduce collectives overlap the backward?  Read from a rocprofv3 --kernel-trace CSV of

    T2S_BENCH_FORCE_DIST=1 rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-dropout0

(the RCCL path with the one rank a 1-GPU box allows: communicator, the bucket all-reduces launched from the backward hooks).
Per train step (a step ends with adam_step_kernel): the collective kernels (names containing "nccl" / "rccl"), the last backward
kernel (the last kernel before grad_sqnorm_kernel that is not a collective), and how many collectives START before that kernel
ENDS.  Exit code 1 unless at least ``min_overlapped`` (default 5) of every step's collectives do; exit code 3 when the trace
holds no collective kernel at all (a one-rank in-place all-reduce may be elided by the library: then the timeline proves nothing
and the summary says so).  usage: overlap_from_trace.py <rocprof output dir> [min_overlapped]"""
import csv
import glob
import json
import sys


def main():
    root = sys.argv[1]
    need = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    files = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)
    if not files:
        print(json.dumps({"error": "no kernel_trace.csv under " + root}))
        return 2
    rows = sorted(csv.DictReader(open(files[0])), key=lambda r: int(r["Start_Timestamp"]))
    steps, cur = [], []
    for r in rows:
        cur.append(r)
        if "adam_step_kernel" in r["Kernel_Name"]:
            steps.append(cur)
            cur = []
    out = {"trace": files[0], "steps": [], "min_overlapped_required": need}
    ok, any_coll = True, False
    for si, st in enumerate(steps):
        is_coll = lambda n: ("nccl" in n.lower() or "rccl" in n.lower())
        coll = [r for r in st if is_coll(r["Kernel_Name"])]
        any_coll |= bool(coll)
        try:
            i_norm = next(i for i, r in enumerate(st) if "grad_sqnorm_kernel" in r["Kernel_Name"])
        except StopIteration:
            continue
        bwd = [r for r in st[:i_norm] if not is_coll(r["Kernel_Name"])]
        if not bwd:
            continue
        last_end = max(int(r["End_Timestamp"]) for r in bwd)
        t0 = int(st[0]["Start_Timestamp"])
        over = [r for r in coll if int(r["Start_Timestamp"]) < last_end]
        # how much compute ran while each collective was in flight: a collective that starts before the last backward kernel ends
        rec = {"step": si, "collective_kernels": len(coll), "start_before_last_backward_kernel_ends": len(over),
               "last_backward_kernel_end_ms": (last_end - t0) / 1e6,
               "collective_start_ms": [round((int(r["Start_Timestamp"]) - t0) / 1e6, 3) for r in coll],
               "collective_duration_ms": [round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, 3) for r in coll],
               "collective_names": sorted({r["Kernel_Name"][:60] for r in coll})}
        out["steps"].append(rec)
        if coll and len(over) < min(need, len(coll)):
            ok = False
    out["verdict"] = ("no collective kernel in the trace: a one-rank all-reduce was elided by the library, the timeline proves nothing"
                      if not any_coll else ("overlap: yes" if ok else "overlap: NO"))
    print(json.dumps(out, indent=1))
    return 3 if not any_coll else (0 if ok else 1)


if __name__ == "__main__":
    sys.exit(main())
