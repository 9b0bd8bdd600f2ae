#!/bin/bash
# does P264AMD_CONCURRENT=1 really overlap the MC kernels?  kernel trace -> per-step timeline
out=gpurun_out/conc
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export P264AMD_CONCURRENT=$1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/trace$1 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $out/trace$1.log 2>&1 || { tail -5 $out/trace$1.log; exit 1; }
python3 - $out/trace$1 <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith("k_")]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = [r for r in rows][-9:]
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    print("%-18s start %9.1f us  end %9.1f us  dur %8.1f" % (r["Kernel_Name"].split("(")[0], (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
grep -o '"ms_per_step": [0-9.]*' $out/trace$1.log
