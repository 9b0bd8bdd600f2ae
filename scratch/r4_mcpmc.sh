#!/bin/bash
# round 4: vector instructions of k_mc with and without its arithmetic (timing-experiment builds)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4_mcpmc
for v in mbase mcopynores; do
  P264AMD_BENCH_NO_GOLDEN=1 P264AMD_LIB=$GRAFT_REPO_ROOT/scratch/lib_$v.so timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU --output-format csv -d gpurun_out/r4_mcpmc/$v -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r4_mcpmc/$v.log 2>&1
  python3 - gpurun_out/r4_mcpmc/$v <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if k.startswith("k_mc"): agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg): print(sys.argv[1].split("/")[-1], k, {c: round(sum(v) / len(v) / 1e6, 1) for c, v in agg[k].items()})
PY
done 2>&1 | tee gpurun_out/r4_mcpmc/summary.txt
find gpurun_out/r4_mcpmc -name "*agent_info.csv" -delete
