#!/bin/bash
# round 5: memory-path counters of k_mc (full kernel and its memory skeleton, scratch/lib_sk.so = -DEXPM_LUMA_COPY=1 -DEXPM_RESID=0)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r5_pmc; mkdir -p $out
i=0
for lib in ${LIBS:-full sk}; do
for ctr in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_BUFFER_TOTAL_CYCLES_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVES" \
           "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  echo "[r5] pmc $lib pass $i: $ctr" >> $out/progress.log
  L=""; [ $lib != full ] && L=$GRAFT_REPO_ROOT/scratch/lib_$lib.so
  P264AMD_TIMING_BUILD_OK=1 P264AMD_BENCH_NO_GOLDEN=1 P264AMD_LIB=$L timeout -k 10 240 rocprofv3 --pmc $ctr --kernel-include-regex "${KRE:-k_mc}" --output-format csv -d $out/pmc_${lib}_$i -- python3 bench.py --steps 2 --warmup 1 --streams ${STREAMS:-2048} --no-cpu-baseline --no-extras > $out/pmc_${lib}_$i.log 2>&1 || { echo "pass $i failed: $ctr"; grep -m2 "Missing\|rror\|nvalid" $out/pmc_${lib}_$i.log; }
done
done
python3 - $out ${LIBS:-full sk} <<'PY' | tee $out/pmc.txt
import csv, glob, sys, collections
for lib in sys.argv[2:]:
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(sys.argv[1] + "/pmc_%s_*/**/*counter_collection.csv" % lib, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if k.startswith("void "): k = k[5:]
            if k.startswith("k_"): agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(agg):
        print(lib, k)
        for c in sorted(agg[k]):
            v = agg[k][c]
            print("   %-44s %16.0f  (%d launches)" % (c, sum(v) / len(v), len(v)))
PY
find $out -name "*agent_info.csv" -delete
rm -rf $out/pmc_*_*/
echo "[r5] pmc done" | tee -a $out/progress.log
