#!/bin/bash
# usage: scratch/pmc_k.sh "<counters>" <kernel-name-prefix> [bench args]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ctr=$1; k=$2; shift; shift
out=gpurun_out/pmck
rm -rf $out
timeout -k 10 150 rocprofv3 --pmc $ctr --output-format csv -d $out -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $out.log 2>&1 || { echo "rocprofv3 failed"; exit 1; }
python scratch/pmc_summary.py $out | grep "$k"
