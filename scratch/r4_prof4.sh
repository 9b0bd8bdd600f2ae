#!/bin/bash
out=gpurun_out/r4_prof4; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 scratch/r4_cfg4.py > $out/trace.log 2>&1 || { echo "trace failed"; tail -5 $out/trace.log; exit 1; }
python3 - $out <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"].split("(")[0]
    n = n[5:] if n.startswith("void ") else n
    if n.startswith("k_"): print("%-20s calls %4s  avg %10.1f us  total %8.2f ms" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
