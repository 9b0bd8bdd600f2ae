#!/bin/bash
# gprof of the single-threaded parse on the GPU box's host cores (no GPU used): scratch/pg/run.sh
R=$GRAFT_REPO_ROOT; cd $R/scratch/pg
S=$(cd $R && python3 -c "
from tests import synth_cases
print(synth_cases.generate('--mbw 120 --mbh 68 --frames 24 --gop 0 --seed 1000 --coded 12 --maxlevel 12 --crop-bottom 4'))")
for f in parser vlc cabac dropin pipeline fanout input_layout; do gcc -O2 -pg -g -fno-inline-functions-called-once -std=gnu11 -I$R/include -I$R/p264decoder_amd/csrc/host -c $R/p264decoder_amd/csrc/host/$f.c -o /tmp/$f.o; done
gcc -O2 -pg -I$R/include -c $R/tests/tools/hip_stub.c -o /tmp/stub.o && gcc -O2 -pg -I$R/include drv.c /tmp/parser.o /tmp/vlc.o /tmp/cabac.o /tmp/dropin.o /tmp/pipeline.o /tmp/fanout.o /tmp/input_layout.o /tmp/stub.o -o /tmp/drv_pg -lpthread -ldl
cd /tmp && ./drv_pg $S 30 && gprof ./drv_pg gmon.out 2>/dev/null | head -30
for f in parser vlc cabac dropin pipeline fanout input_layout; do gcc -O2 -std=gnu11 -I$R/include -I$R/p264decoder_amd/csrc/host -c $R/p264decoder_amd/csrc/host/$f.c -o /tmp/$f.o; done
gcc -O2 -I$R/include $R/scratch/pg/drv.c /tmp/parser.o /tmp/vlc.o /tmp/cabac.o /tmp/dropin.o /tmp/pipeline.o /tmp/fanout.o /tmp/input_layout.o /tmp/stub.o -o /tmp/drv_o2 -lpthread -ldl && ./drv_o2 $S 20
