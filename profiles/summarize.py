#!/usr/bin/env python3
"""Turn gpurun_out/<tag>/ (written by profiles/collect.sh on the GPU box) into the committed summaries:
  profiles/<tag>_bench.json          bench.py's JSON line
  profiles/<tag>_kernel_stats.csv    rocprofv3 --kernel-trace --stats summary (per-kernel calls / total / average ns)
  profiles/<tag>_pmc.json            per-kernel, per-launch averages of every collected counter
  profiles/traffic_latest.json       HBM bytes per launch per bench kernel name (read by bench.py -> roofline.traffic)
HBM traffic = 2 x FETCH_SIZE + WRITE_SIZE (KB as rocprofv3 reports them; FETCH_SIZE counts 128-byte requests as 64 bytes
on gfx950 - MI355X_MICROARCH.md, section HBM)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
here = os.path.dirname(os.path.abspath(__file__))
src = os.path.join(here, "..", "gpurun_out", tag)

shutil.copy(os.path.join(src, "bench.json"), os.path.join(here, tag + "_bench.json"))
def newest(pattern):
    """gpurun merges every collection of a tag into the same directory: only the latest file of a pass counts."""
    found = glob.glob(pattern, recursive=True)
    return max(found, key=os.path.getmtime) if found else None


stats = newest(os.path.join(src, "trace", "**", "*kernel_stats.csv"))
shutil.copy(stats, os.path.join(here, tag + "_kernel_stats.csv"))

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in filter(None, (newest(os.path.join(d, "**", "*counter_collection.csv")) for d in sorted(glob.glob(os.path.join(src, "pmc*")))  if os.path.isdir(d))):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        k = k[5:] if k.startswith("void ") else k              # (template instances: "void k_deblock_bs<false>")
        if k.startswith("k_"):
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
# steady state only: the first launch of k_intra / k_deblock* is the IDR picture (all intra), the k_mc_* kernels have none for it
pmc = {}
for k, ctrs in agg.items():
    pmc[k] = {}
    for c, v in ctrs.items():
        vals = v[1:] if not k.startswith("k_mc") and k != "k_intra_sparse" and len(v) > 1 else v
        pmc[k][c] = round(sum(vals) / len(vals))
    pmc[k]["launches_averaged"] = len(vals)
note = ("per-launch averages over the P-picture launches of `python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline` "
        "(bench.py's default batch of 1080p pictures per launch: see traffic_latest.json), separate rocprofv3 --pmc passes; FETCH_SIZE / WRITE_SIZE in KB as reported")
for k in ("k_mc", "k_mc_sort"):
    if k in pmc:
        pmc[k]["hbm_bytes_2xFETCH_plus_WRITE"] = int((2 * pmc[k].get("FETCH_SIZE", 0) + pmc[k].get("WRITE_SIZE", 0)) * 1024)
json.dump({"note": note, **pmc}, open(os.path.join(here, tag + "_pmc.json"), "w"), indent=1)

def hbm(k):
    return int((2 * pmc[k].get("FETCH_SIZE", 0) + pmc[k].get("WRITE_SIZE", 0)) * 1024)
def hbm_any(prefix):
    return sum(hbm(k) for k in pmc if k == prefix or k.startswith(prefix + "<") or k.startswith(prefix + "_"))
# (the bench's P launches run k_intra_sparse; k_intra itself only sees the all-intra IDR launch of the warm-up)
# (round 4: in P batches the edge-info pass runs inside the k_intra_sparse launch - k_deblock_bs then only appears for the IDR launch
# of the warm-up and is no part of the deblocking stage's launches)
bs_own_launch = any(k.startswith("k_deblock_bs") and pmc[k].get("launches_averaged", 0) + 1 >= pmc.get("k_deblock", {}).get("launches_averaged", 0) for k in pmc)
traffic = {"inter": hbm_any("k_mc"), "intra": hbm("k_intra_sparse" if "k_intra_sparse" in pmc else "k_intra"), "deblock": hbm_any("k_deblock") if bs_own_launch else hbm("k_deblock"),
           "unit": "bytes per launch", "source": tag + "_pmc.json", "formula": "(2*FETCH_SIZE + WRITE_SIZE) KB"}
sys.path.insert(0, os.path.join(here, ".."))
import bench                                               # noqa: E402 - the fingerprint of the kernels these counters belong to
traffic["kernels_sha256"] = bench.kernel_fingerprint()      # (run summarize.py on the tree the collection ran on)
traffic["pictures_per_launch"] = int(json.load(open(os.path.join(here, tag + "_bench.json")))["config"]["pictures_per_step"])   # bench.py's default batch, which collect.sh profiles
if "--no-latest" in sys.argv:                              # (a profile of another batch size: its own file, bench.py's roofline keeps the default batch's)
    json.dump(traffic, open(os.path.join(here, tag + "_traffic.json"), "w"), indent=1)
else:
    json.dump(traffic, open(os.path.join(here, "traffic_latest.json"), "w"), indent=1)
print(json.dumps(traffic))
print(open(os.path.join(here, tag + "_kernel_stats.csv")).read())
