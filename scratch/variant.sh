#!/bin/bash
# usage: scratch/variant.sh <name> <-D flags...>   -> scratch/lib_<name>.so
name=$1; shift
cd /root/repo
# (any EXPM_* / EXPD_* switch makes it a timing build: wrong pictures, runs only with P264AMD_TIMING_BUILD_OK=1)
tb=""; case "$*" in *EXPM_*|*EXPD_*) tb="-DP264AMD_TIMING_BUILD";; esac
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Ip264decoder_amd/csrc/hip $tb "$@" -c p264decoder_amd/csrc/hip/p264hip.hip -o /tmp/var_$name.o && \
hipcc --offload-arch=gfx950 -shared -o scratch/lib_$name.so /tmp/var_$name.o p264decoder_amd/build/*.c.o p264decoder_amd/build/fan_rccl.hip.o -lpthread -ldl && echo built $name
