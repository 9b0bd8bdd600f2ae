#!/bin/bash
# Collect the round's evidence on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1100 -- 'bash profiles/collect.sh r01_final'
# 1. bench.py (default arguments) -> <tag>_bench.json
# 2. rocprofv3 --kernel-trace --stats of the same command -> <tag>_kernel_stats.csv
# 3. separate --pmc passes (never combined with traces): HBM traffic and instruction mix -> <tag>_pmc_*.csv
# Everything lands in gpurun_out/<tag>/; profiles/summarize.py turns it into the committed summaries.
tag=${1:-r01}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# STREAMS=<n>: the same passes at n pictures per launch (e.g. SURVEY 8d's batch: STREAMS=256 bash profiles/collect.sh r05_b256)
S=${STREAMS:+--streams $STREAMS}
BENCH="bench.py --steps 3 --warmup 1 --no-extras $S"
BENCH_FULL="bench.py $S"
echo "[collect] bench" | tee $out/progress.log
python3 bench.py $S ${STREAMS:+--no-extras} > $out/bench.json 2> $out/bench.err || { echo "bench failed"; tail -5 $out/bench.err; exit 1; }
echo "[collect] kernel trace" | tee -a $out/progress.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $BENCH_FULL --no-cpu-baseline --no-extras > $out/trace.log 2>&1 || { echo "trace failed"; tail -5 $out/trace.log; exit 1; }
i=0
for ctr in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  echo "[collect] pmc pass $i: $ctr" | tee -a $out/progress.log
  timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-include-regex "^(void )?k_" --output-format csv -d $out/pmc$i -- python3 $BENCH --no-cpu-baseline > $out/pmc$i.log 2>&1 || { echo "pmc pass $i failed"; grep -m2 "Missing\|error" $out/pmc$i.log; exit 1; }
done
find $out -name "*agent_info.csv" -delete
echo "[collect] done" | tee -a $out/progress.log
