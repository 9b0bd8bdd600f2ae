#!/bin/bash
# round 5: LDS bank-conflict cycles of k_mc by variant (one counter pass each) and the stage time.  usage: r5_ldsconf.sh "<variant ...>"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r5_ldsconf; mkdir -p $out
for v in $1; do
  L=$GRAFT_REPO_ROOT/scratch/lib_$v.so
  P264AMD_TIMING_BUILD_OK=1 P264AMD_BENCH_NO_GOLDEN=1 P264AMD_LIB=$L timeout -k 10 200 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS --kernel-include-regex "^k_mc" --output-format csv -d $out/$v -- python3 bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline > $out/$v.log 2>&1 || { echo "pmc failed $v"; tail -3 $out/$v.log; continue; }
  python3 - $out/$v $v <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("k_mc("): agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(sys.argv[2], {k: round(sum(v) / len(v) / 1e6, 1) for k, v in agg.items()}, "M per launch")
PY
done 2>&1 | tee $out/log_$(date +%H%M%S).txt
[ -z "$PMC_ONLY" ] && SKIP_TESTS=1 NOGOLD=1 bash scratch/r5_ab.sh "$1" 2048
