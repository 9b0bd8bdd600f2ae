#!/bin/bash
# Quick look while iterating on a kernel (run through gpurun from the repo root):
#   gpurun --timeout 600 -- 'bash profiles/quick.sh tag [pmc]'
# kernel trace of a short bench run -> gpurun_out/<tag>/stats.txt; with "pmc": instruction mix, wave cycles and HBM traffic
# per kernel (separate --pmc passes, never combined with a trace) -> gpurun_out/<tag>/pmc.txt
tag=${1:-quick}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
BENCH="bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $BENCH > $out/trace.log 2>&1 || { echo "trace failed"; tail -5 $out/trace.log; exit 1; }
python3 - $out <<'PY' | tee $out/stats.txt
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"].split("(")[0]
    n = n[5:] if n.startswith("void ") else n
    if n.startswith("k_"): print("%-16s calls %4s  avg %10.1f us  total %8.2f ms  %5s %%" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
tail -c 600 $out/trace.log | grep -o '"kernels".*' | head -c 400; echo
[ "$2" = "pmc" ] || exit 0
i=0
for ctr in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES"; do
  i=$((i+1))
  echo "[quick] pmc pass $i: $ctr" >> $out/progress.log
  timeout -k 10 300 rocprofv3 --pmc $ctr --output-format csv -d $out/pmc$i -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $out/pmc$i.log 2>&1 || { echo "pmc pass $i failed: $ctr"; grep -m2 "Missing\|rror" $out/pmc$i.log; }
done
python3 - $out <<'PY' | tee $out/pmc.txt
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if k.startswith("k_"): agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    print(k)
    for c in sorted(agg[k]):
        v = agg[k][c]; v = v[1:] if len(v) > 1 else v          # the first launch is the IDR picture
        print("   %-28s %16.0f" % (c, sum(v) / len(v)))
PY
find $out -name "*agent_info.csv" -delete
