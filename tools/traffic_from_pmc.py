#!/usr/bin/env python3
"""HBM bytes per launch of the attention kernels from the two rocprofv3 PMC passes of tools/profile_round.sh.

usage: traffic_from_pmc.py <out dir with fetch/ write/ bench.json>   -> JSON on stdout (the entry of profiles/traffic.json)

bytes of a dispatch = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024 (FETCH_SIZE / WRITE_SIZE are in KB; gfx950 reports half of a wide
coalesced read stream: MI355X_MICROARCH.md, HBM / rocprofv3 section).  The PMC passes run `bench.py --steps 1 --warmup 1`: two
identical steps, so a family's bytes per step are its total over the run / 2.  "Per launch" divides by the number of launches
bench.py prices (those with >= 1 GFLOP: `roofline*.launches / steps` of the bench line); the launches below that bound (the
20-token text encoder) move kilobytes and are left in the numerator."""
import collections
import csv
import glob
import json
import os
import sys

out = sys.argv[1]
STEPS_PROFILED = 2


def total(counter_dir, counter):
    acc = collections.defaultdict(float)
    n = collections.defaultdict(int)
    for f in glob.glob(out + "/" + counter_dir + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]] += float(r["Counter_Value"])
                n[r["Kernel_Name"]] += 1
    return acc, n


fetch, nf = total("fetch", "FETCH_SIZE")
write, _ = total("write", "WRITE_SIZE")
bench = json.loads([l for l in open(out + "/bench.json") if l.startswith("{")][-1])
fam = {"attn_fwd": ("attn_fwd_",),
       "attn_bwd": ("attn_delta", "attn_dkdv", "attn_dq", "attn_bwd_fused")}
res = {"_source": "tools/traffic_from_pmc.py over tools/profile_round.sh's PMC passes: sum over the family's dispatches of "
                  "2*FETCH_SIZE*1024 + WRITE_SIZE*1024, per step, / launches per step priced by bench.py"}
for key, pats in fam.items():
    names = [k for k in set(fetch) | set(write) if any(p in k for p in pats)]
    b = sum(2 * fetch.get(k, 0.0) * 1024 + write.get(k, 0.0) * 1024 for k in names) / STEPS_PROFILED
    blk = bench["roofline"] if key == "attn_bwd" else bench.get("roofline_fwd", bench["roofline"])
    per_step = blk["launches"] / bench["steps"]
    res[key] = b / per_step
    res["_" + key + "_detail"] = {"bytes_per_step": b, "launches_per_step": per_step,
                                  "kernels": {k[:70]: {"dispatches_per_step": nf.get(k, 0) / STEPS_PROFILED,
                                                       "fetch_kb_per_step": fetch.get(k, 0.0) / STEPS_PROFILED,
                                                       "write_kb_per_step": write.get(k, 0.0) / STEPS_PROFILED} for k in sorted(names)}}
# "_round": which build / call the passes ran on - bench.py quotes it beside the figure it reads from this file
print(json.dumps({"_round": "profile tag " + os.path.basename(os.path.normpath(out)),
                  "B%d_F%d_P%d_drop%g" % (64, 100, 100, bench["config"]["dropout"]): res}, indent=1))
