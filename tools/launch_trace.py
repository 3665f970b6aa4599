#!/usr/bin/env python3
"""Per-launch durations of the attention kernels from a rocprofv3 --kernel-trace CSV, in launch order (one train step):
shows how the step's attention time splits between the heavy launches (QTV layers and the ref MMT pass, ~70 % of the
keys visible) and the light ones (pos / neg passes, TextBert).
usage: launch_trace.py <rocprof output dir> [skip_first_n_steps_fraction]"""
import csv
import glob
import sys

root = sys.argv[1]
f = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
half = len(rows) // 2                 # --steps 1 --warmup 1: the second half of the trace is the timed step
rows = rows[half:]
tot = {}
for r in rows:
    n = r["Kernel_Name"]
    if "attn_" not in n:
        continue
    short = n.split("::")[-1].split("(")[0][:40]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    tot.setdefault(short, []).append(d)
for k, v in tot.items():
    print("%-42s n=%3d total %8.2f ms  launches: %s" % (k, len(v), sum(v), " ".join("%.2f" % x for x in v)))
