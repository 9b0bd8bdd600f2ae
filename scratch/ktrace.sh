#!/bin/bash
# per-kernel durations of the last bench step; usage: ktrace.sh tag   (environment passes through)
out=gpurun_out/kt_$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $out/trace.log 2>&1 || { tail -5 $out/trace.log; exit 1; }
python3 - $out/trace $1 <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith("k_") and not r["Kernel_Name"].startswith("k_tile")]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-8:]
print(sys.argv[2], " ".join("%s %.0f" % (r["Kernel_Name"].split("(")[0][2:], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in last))
PY
