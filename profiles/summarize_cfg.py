#!/usr/bin/env python3
"""gpurun_out/<tag>/ (profiles/collect_cfg.sh) -> profiles/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats),
profiles/<tag>_pmc.json (per kernel: mean of every counter over its launches, HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) KB) and
profiles/<tag>_stages.txt (bench-style stage times per picture, HIP events)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1]
here = os.path.dirname(os.path.abspath(__file__))
src = os.path.join(here, "..", "gpurun_out", tag)
def newest(pattern):
    """gpurun merges every collection of a tag into the same directory: only the latest file of a pass counts."""
    found = glob.glob(pattern, recursive=True)
    return max(found, key=os.path.getmtime) if found else None


shutil.copy(newest(os.path.join(src, "trace", "**", "*kernel_stats.csv")), os.path.join(here, tag + "_kernel_stats.csv"))
shutil.copy(os.path.join(src, "stages.txt"), os.path.join(here, tag + "_stages.txt"))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in filter(None, (newest(os.path.join(d, "**", "*counter_collection.csv")) for d in sorted(glob.glob(os.path.join(src, "pmc*"))) if os.path.isdir(d))):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        k = k[5:] if k.startswith("void ") else k
        if k.startswith("k_"):
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
pmc = {}
for k, ctrs in sorted(agg.items()):
    pmc[k] = {c: round(sum(v) / len(v)) for c, v in sorted(ctrs.items())}
    pmc[k]["launches"] = max(len(v) for v in ctrs.values())
    if "FETCH_SIZE" in pmc[k] and "WRITE_SIZE" in pmc[k]:
        pmc[k]["hbm_bytes_2xFETCH_plus_WRITE"] = int((2 * pmc[k]["FETCH_SIZE"] + pmc[k]["WRITE_SIZE"]) * 1024)
note = "per-launch means over ALL launches of `python3 profiles/cfg_run.py` (1024 pictures per launch; two passes over the stream's pictures), separate rocprofv3 --pmc passes; FETCH_SIZE / WRITE_SIZE in KB as reported"
json.dump({"note": note, **pmc}, open(os.path.join(here, tag + "_pmc.json"), "w"), indent=1)
print(open(os.path.join(here, tag + "_kernel_stats.csv")).read())
print(json.dumps({k: v.get("hbm_bytes_2xFETCH_plus_WRITE") for k, v in pmc.items()}))
