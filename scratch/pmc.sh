#!/bin/bash
# usage: scratch/pmc.sh <outdir> "<counters>" [bench args]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=$1; shift; ctr=$1; shift
rocprofv3 --pmc $ctr --output-format csv -d $out -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $out.log 2>&1
python scratch/pmc_summary.py $out
