#!/bin/bash
# usage: scratch/pmc_var.sh <variant|-> "<counters>"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
v=$1; ctr=$2
[ "$v" != "-" ] && export P264AMD_LIB=$GRAFT_REPO_ROOT/scratch/lib_$v.so
out=gpurun_out/pmcv_$v
timeout -k 10 150 rocprofv3 --pmc $ctr --output-format csv -d $out -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out.log 2>&1 || { echo "rocprofv3 failed"; exit 1; }
echo "== $v"; python scratch/pmc_summary.py $out | grep "k_inter\|k_deblock_bs"
