#!/bin/bash
# usage: scratch/variant.sh <name> <-D flags...>   -> scratch/lib_<name>.so
name=$1; shift
cd /root/repo
# (any EXPM_* / EXPD_* switch makes it a timing build: wrong pictures, runs only with P264AMD_TIMING_BUILD_OK=1)
tb=""; case "$*" in *EXPM_*|*EXPD_*) tb="-DP264AMD_TIMING_BUILD";; esac
objs=""
# (round 6: k_deblock is a translation unit of its own with the max-ILP scheduling strategy, as build.py compiles it; a tree from
#  before that has no k_deblock.hip)
for tu in p264hip k_deblock; do
  [ -f p264decoder_amd/csrc/hip/$tu.hip ] || continue
  extra=""; [ $tu = k_deblock ] && extra="${KDB_FLAGS--mllvm -amdgpu-sched-strategy=max-ilp}"     # KDB_FLAGS: other options for that unit alone
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Ip264decoder_amd/csrc/hip $tb $extra "$@" -c p264decoder_amd/csrc/hip/$tu.hip -o /tmp/var_${name}_$tu.o || exit 1
  objs="$objs /tmp/var_${name}_$tu.o"
done
hipcc --offload-arch=gfx950 -shared -o scratch/lib_$name.so $objs p264decoder_amd/build/*.c.o p264decoder_amd/build/fan_rccl.hip.o -lpthread -ldl && echo built $name
