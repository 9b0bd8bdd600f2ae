#!/bin/bash
# usage: scratch/kstats.sh <outdir> [bench args]  -> per-kernel average durations
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=$1; shift
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > $out.log 2>&1 || { echo "rocprofv3 failed"; tail -5 $out.log; exit 1; }
f=$(find $out -name "*kernel_stats.csv" | head -1)
cat $f | cut -d, -f1-6 | head -12
