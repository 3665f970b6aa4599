#!/usr/bin/env python3
"""MFMA-busy fraction of the attention kernels from the SQ passes of tools/pmc_attn.sh:
    busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)
(the counter sums the cycles in which a SIMD's matrix pipe is busy over the 1024 SIMDs of the chip; GRBM_GUI_ACTIVE sums the
kernel's cycles over the 8 XCDs: MI355X_MICROARCH.md, DVFS give-back).  usage: mfma_busy_from_pmc.py <pmc dir> [label] -> JSON"""
import collections
import csv
import glob
import json
import sys

root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in acc.items():
    if "attn" not in k or "SQ_VALU_MFMA_BUSY_CYCLES" not in cs or "GRBM_GUI_ACTIVE" not in cs:
        continue
    mean = lambda v: sum(v) / len(v)
    busy, act = mean(cs["SQ_VALU_MFMA_BUSY_CYCLES"]), mean(cs["GRBM_GUI_ACTIVE"])
    if busy <= 0 or act <= 0:
        continue
    e = {"mfma_busy": busy / (act / 8 * 1024)}
    if "SQ_INSTS_VALU" in cs and "SQ_INSTS_MFMA" in cs and mean(cs["SQ_INSTS_MFMA"]) > 0:
        e["valu_per_mfma"] = mean(cs["SQ_INSTS_VALU"]) / mean(cs["SQ_INSTS_MFMA"])
    if "SQ_LDS_BANK_CONFLICT" in cs and "SQ_LDS_IDX_ACTIVE" in cs and mean(cs["SQ_LDS_IDX_ACTIVE"]) > 0:
        e["lds_bank_conflict_frac"] = mean(cs["SQ_LDS_BANK_CONFLICT"]) / mean(cs["SQ_LDS_IDX_ACTIVE"])
    name = k.replace("void ", "").replace("(anonymous namespace)::", "").strip()
    base = name.split("<")[0].split("(")[0]
    # one entry per kernel FAMILY; of a family's instantiations keep the one with the most MFMA-busy cycles (the steady-state form)
    if base not in out or busy > out[base]["_busy_cycles"]:
        e["_busy_cycles"] = busy
        e["instantiation"] = name[:90]
        out[base] = e
for e in out.values():
    e.pop("_busy_cycles", None)
print(json.dumps({sys.argv[2] if len(sys.argv) > 2 else "attn_probe": out}, indent=1))
