#!/bin/bash
# round 5: one parser thread, two builds of the host sources side by side on the box's host cores (no GPU used).
# here:  scratch/r5_parse_ab.sh build <name>     -> scratch/t/drv_<name> from the working tree (-O3 -march=x86-64-v3, as build.py)
# box:   gpurun -- 'bash scratch/r5_parse_ab.sh run base cur'
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
if [ "$1" = build ]; then
  n=$2; shift 2                                      # (further arguments: extra compiler flags)
  W=/tmp/pab_$n; mkdir -p $W scratch/t
  set -- "$n" "$n" "$@"
  for f in parser vlc cabac dropin pipeline fanout input_layout; do gcc -O3 -march=x86-64-v3 -falign-functions=64 "${@:3}" -std=gnu11 -Iinclude -Ip264decoder_amd/csrc/host -c p264decoder_amd/csrc/host/$f.c -o $W/$f.o || exit 1; done
  gcc -O2 -Iinclude -c tests/tools/hip_stub.c -o $W/stub.o && gcc -O2 -Iinclude scratch/pg/drv.c $W/*.o -o scratch/t/drv_$2 -lpthread -ldl 2>/dev/null && echo built scratch/t/drv_$2
  exit
fi
shift
S3=$(python3 -c "
from tests import synth_cases
print(synth_cases.generate('--mbw 120 --mbh 68 --frames 24 --gop 0 --seed 1000 --coded 12 --maxlevel 12 --crop-bottom 4'))")
S4=$(python3 -c "
from tests import synth_cases
print(synth_cases.generate(synth_cases.ORACLE_CASES['main_1080p_cabac_ipb']))")
mkdir -p gpurun_out/r5_parse_ab
for rep in 1 2; do for n in "$@"; do
  echo -n "$n: CAVLC "; for i in 1 2 3 4 5; do scratch/t/drv_$n $S3 20 | tr '\n' ' '; done; echo -n " CABAC "; for i in 1 2 3; do scratch/t/drv_$n $S4 6 | tr '\n' ' '; done; echo
done; done 2>&1 | tee gpurun_out/r5_parse_ab/log_$(date +%H%M%S).txt
