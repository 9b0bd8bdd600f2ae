#!/bin/bash
# Kernel trace + HBM / instruction counters of a non-metric workload (run through gpurun from the repo root):
#   gpurun --timeout 900 -- 'bash profiles/collect_cfg.sh config4 r04_cfg4'
# -> gpurun_out/<tag>/ ; profiles/summarize_cfg.py <tag> turns it into profiles/<tag>_kernel_stats.csv and <tag>_pmc.json.
# (--pmc passes are never combined with trace domains; the program comes directly after `--`; counters only for our own
# kernels - with the thousands of copy kernels of the set-up counted too a pass takes minutes.)
cfg=${1:-config4}; tag=${2:-r04_cfg4}; npic=${3:-13}     # (counter passes of an all-intra stream: 3 pictures are plenty and keep the pass short)
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 profiles/cfg_run.py $cfg > $out/trace.log 2>&1 || { echo "trace failed"; tail -5 $out/trace.log; exit 1; }
grep picture $out/trace.log > $out/stages.txt
i=0
for ctr in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  echo "[collect_cfg] pmc pass $i: $ctr" | tee -a $out/progress.log
  timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-include-regex "^(void )?k_" --output-format csv -d $out/pmc$i -- python3 profiles/cfg_run.py $cfg 1024 $npic > $out/pmc$i.log 2>&1 || { echo "pmc pass $i failed"; grep -m2 "Missing\|rror" $out/pmc$i.log; exit 1; }
done
find $out -name "*agent_info.csv" -delete
echo "[collect_cfg] done" | tee -a $out/progress.log
