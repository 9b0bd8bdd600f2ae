#!/bin/bash
# single-threaded parse rate by compiler flags, on the GPU box's host
R=$GRAFT_REPO_ROOT; cd $R
S=$(python3 -c "
from tests import synth_cases
print(synth_cases.generate('--mbw 120 --mbh 68 --frames 24 --gop 0 --seed 1000 --coded 12 --maxlevel 12 --crop-bottom 4'))")
S2=$(python3 -c "
from tests import synth_cases
print(synth_cases.generate(synth_cases.ORACLE_CASES['main_1080p_cabac_ipb']))")
n=0
for flags in "-O2" "-O3" "-O2 -march=x86-64-v3" "-O3 -march=x86-64-v3" "-O2 -fno-semantic-interposition -fvisibility=hidden" "-O3 -fno-semantic-interposition" "-O2 -flto"; do
  n=$((n+1))
  for f in parser vlc cabac dropin pipeline fanout input_layout; do gcc $flags -std=gnu11 -I$R/include -I$R/p264decoder_amd/csrc/host -c $R/p264decoder_amd/csrc/host/$f.c -o /tmp/$f.o || exit 1; done
  gcc -O2 -I$R/include -c $R/tests/tools/hip_stub.c -o /tmp/stub.o; gcc $flags -I$R/include $R/scratch/pg/drv.c /tmp/parser.o /tmp/vlc.o /tmp/cabac.o /tmp/dropin.o /tmp/pipeline.o /tmp/fanout.o /tmp/input_layout.o /tmp/stub.o -o /tmp/drv_$n -lpthread -ldl 2>/dev/null
  echo -n "$flags: cavlc "; /tmp/drv_$n $S 15 | tr '\n' ' '; echo -n " cabac "; /tmp/drv_$n $S2 5
done
