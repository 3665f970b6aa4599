#!/usr/bin/env python3
"""Per-CALL view of a rocprofv3 --kernel-trace CSV: for the heaviest kernel names, the individual launch durations grouped by
(grid, workgroup) size - the aggregated stats hide which of a library kernel's many call sites carry its time.
usage: python tools/ktrace_calls.py <rocprof output dir> [steps] [top names] [name filter]"""
import collections
import csv
import glob
import sys

root = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 8
flt = sys.argv[4] if len(sys.argv) > 4 else ""
f = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)[0]
by = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    name = r.get("Kernel_Name") or r.get("Name")
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    shape = "grid %sx%sx%s wg %s lds %s" % (r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["Workgroup_Size_X"], r["LDS_Block_Size"])
    by[name].append((shape, d))
tot = {n: sum(d for _, d in v) for n, v in by.items()}
names = [n for n in sorted(tot, key=tot.get, reverse=True) if flt in n][:top]
for n in names:
    print("%s\n   total %.2f ms/step over %d calls" % (n[:150], tot[n] / steps, len(by[n])))
    g = collections.defaultdict(list)
    for shape, d in by[n]:
        g[shape].append(d)
    for shape, ds in sorted(g.items(), key=lambda kv: -sum(kv[1])):
        ds.sort()
        print("     %-46s n=%4d  %8.2f ms/step  median %7.3f ms  min %7.3f  max %7.3f" % (shape, len(ds), sum(ds) / steps, ds[len(ds) // 2], ds[0], ds[-1]))
