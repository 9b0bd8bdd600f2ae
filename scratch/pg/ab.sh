#!/bin/bash
# A/B of the single-threaded parse rate on the GPU box's host: HEAD's host sources against the working tree's
R=$GRAFT_REPO_ROOT; cd $R
S=$(python3 -c "
from tests import synth_cases
print(synth_cases.generate('--mbw 120 --mbh 68 --frames 24 --gop 0 --seed 1000 --coded 12 --maxlevel 12 --crop-bottom 4'))")
S2=$(python3 -c "
from tests import synth_cases
print(synth_cases.generate(synth_cases.ORACLE_CASES['main_1080p_cabac_ipb']))")
build() { # dir of host sources, output
  for f in parser vlc cabac dropin pipeline fanout input_layout; do gcc -O3 -std=gnu11 -fPIC -I$R/include -I$1 -c $1/$f.c -o /tmp/$f.o || exit 1; done
  gcc -O2 -I$R/include -c $R/tests/tools/hip_stub.c -o /tmp/stub.o; gcc -O3 -I$R/include $R/scratch/pg/drv.c /tmp/parser.o /tmp/vlc.o /tmp/cabac.o /tmp/dropin.o /tmp/pipeline.o /tmp/fanout.o /tmp/input_layout.o /tmp/stub.o -o $2 -lpthread -ldl 2>/dev/null
}
build $R/scratch/pg/old /tmp/drv_old; build $R/p264decoder_amd/csrc/host /tmp/drv_new
for i in 1 2 3; do echo -n "old: "; /tmp/drv_old $S 15; echo -n "new: "; /tmp/drv_new $S 15; done
echo -n "cabac old: "; /tmp/drv_old $S2 5; echo -n "cabac new: "; /tmp/drv_new $S2 5
