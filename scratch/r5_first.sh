#!/bin/bash
# round 5, first GPU session: the GPU suite (with the bench's own batch under a parity test), the headline at 2048 and at 256
# pictures per launch, the kernel trace at 256, and the memory-path counters of k_mc (full kernel and its memory skeleton)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r5_first; mkdir -p $out
echo "[r5] gpu tests" | tee $out/progress.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1 || { echo "gpu tests failed"; tail -30 $out/tests.log; exit 1; }
tail -2 $out/tests.log
echo "[r5] bench 2048" | tee -a $out/progress.log
python bench.py --no-cpu-baseline --no-extras > $out/b2048.json 2> $out/b2048.err || { echo "bench failed"; tail -5 $out/b2048.err; exit 1; }
python - $out/b2048.json <<'PY'
import sys, json
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k = d["kernels"]
print("2048:", round(d["value"]), d["ms_per_step"], {n: k[n]["avg_ms"] for n in k}, d["launch"])
PY
echo "[r5] bench 256" | tee -a $out/progress.log
python bench.py --only-batch-256 > $out/b256.json 2> $out/b256.err || { echo "b256 failed"; tail -5 $out/b256.err; exit 1; }
python - $out/b256.json <<'PY'
import sys, json
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])["extras"]["batch_256"]
print("256:", d["value"], d["ms_per_step"], {n: (s["avg_ms"], s["frac_of_hbm_peak"]) for n, s in d["stages"].items()}, d["launch"], d["last_picture_matches_reference"])
PY
echo "[r5] trace 256" | tee -a $out/progress.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace256 -- python3 bench.py --only-batch-256 --steps 10 > $out/trace256.log 2>&1 || { echo "trace failed"; tail -5 $out/trace256.log; }
python3 - $out <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/trace256/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"].split("(")[0]; n = n[5:] if n.startswith("void ") else n
        if n.startswith("k_"): print("%-16s calls %4s  avg %10.1f us" % (n, r["Calls"], float(r["AverageNs"]) / 1e3))
PY
[ -n "$R5_NO_PMC" ] && { echo "[r5] done (no pmc)"; exit 0; }
i=0
for lib in full sk; do
for ctr in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_BUFFER_TOTAL_CYCLES_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES"; do
  i=$((i+1))
  echo "[r5] pmc $lib pass $i: $ctr" >> $out/progress.log
  L=""; [ $lib = sk ] && L=$GRAFT_REPO_ROOT/scratch/lib_sk.so
  P264AMD_TIMING_BUILD_OK=1 P264AMD_BENCH_NO_GOLDEN=1 P264AMD_LIB=$L timeout -k 10 240 rocprofv3 --pmc $ctr --kernel-include-regex "k_mc" --output-format csv -d $out/pmc_${lib}_$i -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $out/pmc_${lib}_$i.log 2>&1 || { echo "pass $i failed: $ctr"; grep -m2 "Missing\|rror\|nvalid" $out/pmc_${lib}_$i.log; }
done
done
python3 - $out <<'PY' | tee $out/pmc.txt
import csv, glob, sys, collections
for lib in ("full", "sk"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(sys.argv[1] + "/pmc_%s_*/**/*counter_collection.csv" % lib, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if k.startswith("k_mc"): agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(agg):
        print(lib, k)
        for c in sorted(agg[k]):
            v = agg[k][c]
            print("   %-44s %16.0f" % (c, sum(v) / len(v)))
PY
find $out -name "*agent_info.csv" -delete
echo "[r5] done" | tee -a $out/progress.log
