#!/bin/bash
# round 5: request sizes towards memory and L2 hit rates per kernel (two counters per pass; a refused group costs its timeout)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r5_pmc2; mkdir -p $out
i=0
for ctr in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_EA0_RDREQ_128B_sum" "GRBM_GUI_ACTIVE" "TA_TA_BUSY_sum TA_BUFFER_READ_WAVEFRONTS_sum"; do
  i=$((i+1))
  echo "[r5] pmc2 pass $i: $ctr" >> $out/progress.log
  timeout -k 10 ${PASS_TIMEOUT:-110} rocprofv3 --pmc $ctr --kernel-include-regex "^(void )?k_" --output-format csv -d $out/p$i -- python3 bench.py --steps 2 --warmup 1 --streams ${STREAMS:-2048} --no-cpu-baseline --no-extras > $out/p$i.log 2>&1 || { echo "pass $i failed: $ctr"; grep -m1 "Missing\|rror\|nvalid\|exceeds" $out/p$i.log; }
done
python3 - $out <<'PY' | tee $out/pmc2.txt
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if k.startswith("void "): k = k[5:]
        if k.startswith("k_"): agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    print(k)
    for c in sorted(agg[k]):
        v = agg[k][c]; v = v[1:] if len(v) > 2 else v
        print("   %-36s %16.0f  (%d launches)" % (c, sum(v) / len(v), len(v)))
PY
rm -rf $out/p*/
echo "[r5] pmc2 done" | tee -a $out/progress.log
