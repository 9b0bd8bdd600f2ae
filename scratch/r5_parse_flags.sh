#!/bin/bash
# round 5: single-thread CAVLC / CABAC parse rate on the GPU box's host cores by compiler flags and with a profile-guided build
# (no GPU used).  usage: gpurun -- 'bash scratch/r5_parse_flags.sh'
R=$GRAFT_REPO_ROOT; cd $R
out=gpurun_out/r5_parse_flags; mkdir -p $out; W=/tmp/pf; mkdir -p $W
S3=$(python3 -c "
from tests import synth_cases
print(synth_cases.generate('--mbw 120 --mbh 68 --frames 24 --gop 0 --seed 1000 --coded 12 --maxlevel 12 --crop-bottom 4'))")
S4=$(python3 -c "
from tests import synth_cases
print(synth_cases.generate(synth_cases.ORACLE_CASES['main_1080p_cabac_ipb']))")
H=$R/p264decoder_amd/csrc/host
gcc -O2 -I$R/include -c $R/tests/tools/hip_stub.c -o $W/stub.o
build() {   # name, flags...
  n=$1; shift
  for f in parser vlc cabac dropin pipeline fanout input_layout; do gcc "$@" -std=gnu11 -I$R/include -I$H -c $H/$f.c -o $W/${n}_$f.o 2>>$out/warn_$n.log || return 1; done
  gcc -O2 -I$R/include $R/scratch/pg/drv.c $W/${n}_parser.o $W/${n}_vlc.o $W/${n}_cabac.o $W/${n}_dropin.o $W/${n}_pipeline.o $W/${n}_fanout.o $W/${n}_input_layout.o $W/stub.o -o $W/drv_$n -lpthread -ldl $LINK 2>>$out/warn_$n.log
}
rate() { echo -n "$1: CAVLC "; for i in 1 2 3 4 5; do $W/drv_$1 $S3 20 | tr '\n' ' '; done; echo -n " CABAC "; for i in 1 2 3; do $W/drv_$1 $S4 6 | tr '\n' ' '; done; echo; }
{
lscpu | grep "Model name"
build o3 -O3 && rate o3
build o3v3 -O3 -march=x86-64-v3 && rate o3v3
build o3zen -O3 -march=znver3 && rate o3zen
build o2v3 -O2 -march=x86-64-v3 && rate o2v3
# profile-guided: instrumented build, one pass over both streams, rebuilt with the profile (same object names: the .gcda files are found by them)
LINK=-lgcov build pgo -O3 -fprofile-generate -fprofile-update=single && (cd $W && ./drv_pgo $S3 2 >/dev/null && ./drv_pgo $S4 1 >/dev/null)
ls $W/*.gcda | wc -l
build pgo -O3 -fprofile-use -fprofile-correction && rate pgo
build pgov3 -O3 -march=x86-64-v3 -fprofile-use -fprofile-correction 2>/dev/null; for f in parser vlc cabac dropin pipeline fanout input_layout; do cp $W/pgo_$f.gcda $W/pgov3_$f.gcda; done; build pgov3 -O3 -march=x86-64-v3 -fprofile-use -fprofile-correction && rate pgov3
rate o3
} 2>&1 | tee $out/log.txt
grep -c "missing-profile\|not found" $out/warn_pgo.log
