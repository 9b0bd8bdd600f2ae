#!/bin/bash
# memory-path counters per kernel (separate --pmc passes, no trace domains); usage: pmc_mem.sh tag
tag=${1:-pm}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for ctr in "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS SQ_BUSY_CYCLES" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_TOTAL_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_GATE_EN1_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum" \
           "TCC_BUSY_sum TCC_TAG_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_REQ_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum"; do
  i=$((i+1))
  echo "[pmc_mem] pass $i: $ctr" >> $out/progress.log
  timeout -k 10 300 rocprofv3 --pmc $ctr --output-format csv -d $out/pmc$i -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $out/pmc$i.log 2>&1 || { echo "pass $i failed: $ctr"; grep -m2 "Missing\|rror\|nvalid" $out/pmc$i.log; }
done
python3 - $out <<'PY' | tee $out/pmc.txt
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if k.startswith("k_") and not k.startswith("k_tile"): agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for k in agg for c in agg[k]})
ks = sorted(agg)
print("%-40s" % "" + "".join("%16s" % k[2:] for k in ks))
for c in names:
    row = []
    for k in ks:
        v = agg[k].get(c, [0]); v = v[1:] if len(v) > 1 else v
        row.append(sum(v) / len(v))
    print("%-40s" % c + "".join("%16.0f" % x for x in row))
PY
find $out -name "*agent_info.csv" -delete
