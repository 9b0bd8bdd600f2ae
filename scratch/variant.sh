#!/bin/bash
# usage: scratch/variant.sh <name> <-D flags...>   -> scratch/lib_<name>.so
name=$1; shift
cd /root/repo
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Ip264decoder_amd/csrc/hip "$@" -c p264decoder_amd/csrc/hip/p264hip.hip -o /tmp/var_$name.o && \
hipcc --offload-arch=gfx950 -shared -o scratch/lib_$name.so /tmp/var_$name.o p264decoder_amd/build/*.c.o p264decoder_amd/build/fan_rccl.hip.o -lpthread -ldl && echo built $name
