#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
S2=$(python3 -c "
from tests import synth_cases
print(synth_cases.generate(synth_cases.ORACLE_CASES['main_1080p_cabac_ipb']))")
for f in parser vlc cabac dropin pipeline fanout input_layout; do gcc -O2 -pg -g -fno-inline-functions-called-once -std=gnu11 -I$R/include -I$R/p264decoder_amd/csrc/host -c $R/p264decoder_amd/csrc/host/$f.c -o /tmp/$f.o; done
gcc -O2 -pg -I$R/include -c $R/tests/tools/hip_stub.c -o /tmp/stub.o && gcc -O2 -pg -I$R/include scratch/pg/drv.c /tmp/parser.o /tmp/vlc.o /tmp/cabac.o /tmp/dropin.o /tmp/pipeline.o /tmp/fanout.o /tmp/input_layout.o /tmp/stub.o -o /tmp/drv_pg -lpthread -ldl 2>/dev/null
cd /tmp && ./drv_pg $S2 12 && gprof -b ./drv_pg gmon.out 2>/dev/null | head -28; ls -la $S2
