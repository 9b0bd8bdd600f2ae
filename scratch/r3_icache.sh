#!/bin/bash
out=gpurun_out/r3_icache; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --list-avail 2>/dev/null | grep -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQC_INST[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*\|SQ_WAIT_INST[A-Z_]*\|SQ_INSTS_BRANCH\|SQ_VALU_MFMA_BUSY_CYCLES\|SQ_ACTIVE_INST_VALU\|SQ_THREAD_CYCLES_VALU\|SQ_IFETCH_LEVEL" | sort -u > $out/avail.txt
cat $out/avail.txt | tr '\n' ' '
timeout -k 10 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH --output-format csv -d $out/pmc1 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $out/pmc1.log 2>&1 || { grep -m3 "Missing\|rror" $out/pmc1.log; }
python3 - $out <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if k.startswith("k_"): agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    print(k, {c: round(sum(v[1:]) / max(len(v[1:]), 1)) for c, v in agg[k].items()})
PY
