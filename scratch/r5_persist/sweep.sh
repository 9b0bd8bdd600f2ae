#!/bin/bash
# round 5: k_mc with workgroups that stay (mc_roles' loop over the pieces of work): the MC stage by workgroups in the launch
# (P264AMD_MC_RESIDENT; >= pieces = one piece per workgroup as before) and by pieces per picture.  usage: r5_persist.sh "<resident ...>" "<wgs per picture ...>"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r5_persist; mkdir -p $out
for st in ${3:-2048}; do
for wg in ${2:-48}; do
for res in $1; do
  echo -n "streams $st wgs/pic $wg resident $res: "
  P264AMD_MC_RESIDENT=$res P264AMD_MC_WGS_PER_PIC=$wg python bench.py --steps 10 --warmup 2 --streams $st --no-cpu-baseline --no-extras 2>$out/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; print(round(d['value']), 'inter', k['inter']['avg_ms'], 'golden', d['golden_check'].get('checked'))" || tail -3 $out/err.log
done
done
done 2>&1 | tee $out/log_$(date +%H%M%S).txt
