#!/bin/bash
# usage: scratch/pmc.sh <outdir> "<counters>" [bench args]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=$1; shift; ctr=$1; shift
timeout -k 10 150 rocprofv3 --pmc $ctr --output-format csv -d $out -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $out.log 2>&1 || { echo "rocprofv3 failed"; grep -m3 "Missing\|error" $out.log; exit 1; }
python scratch/pmc_summary.py $out
