#!/bin/bash
# round 5: kernel times of config 4 per library variant (rocprofv3 --kernel-trace --stats of profiles/cfg_run.py)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in ${@:-base cur}; do
  out=gpurun_out/r5_cfg4trace_$v; rm -rf $out; mkdir -p $out
  L=""; [ $v != cur ] && L=$GRAFT_REPO_ROOT/scratch/lib_$v.so
  P264AMD_LIB=$L timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 profiles/cfg_run.py config4 1024 > $out.log 2>&1 || { echo "trace failed $v"; tail -3 $out.log; continue; }
  echo "== $v"
  python3 - $out <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"].split("(")[0]; n = n[5:] if n.startswith("void ") else n
    if n.startswith("k_"): print("  %-20s calls %3s avg %9.1f us total %8.2f ms" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
  rm -rf $out
done 2>&1 | tee gpurun_out/r5_cfg4trace.log
