#!/usr/bin/env python3
"""Print the top kernels of a rocprofv3 --kernel-trace --stats CSV (per-step ms given the number of steps)."""
import csv
import glob
import sys

root, steps = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
f = glob.glob(root + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total %.1f ms  (%.1f ms/step over %g steps)" % (tot / 1e6, tot / 1e6 / steps, steps))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 25]:
    print("%-86s n=%5s %8.2f ms/step avg %8.3f ms %5.1f%%" % (r["Name"][:86], r["Calls"], float(r["TotalDurationNs"]) / 1e6 / steps,
                                                             float(r["AverageNs"]) / 1e6, float(r["Percentage"])))
