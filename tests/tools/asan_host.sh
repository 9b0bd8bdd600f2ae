#!/bin/bash
# AddressSanitizer + UBSan over the host code (CAVLC parser, drop-in API, pipeline) - CPU only, the HIP layer is stubbed out
# (GPU sanitizers are not available on the pool).  Parses every golden stream, 60 randomly corrupted / truncated streams
# and runs the threaded parse-only pipeline.  Usage: tests/tools/asan_host.sh
set -e
here=$(cd "$(dirname "$0")" && pwd); root=$(cd "$here/../.." && pwd); out=${TMPDIR:-/tmp}/p264amd_asan_$$
mkdir -p $out
for f in parser vlc cabac dropin pipeline fanout input_layout compact cpu_check; do
  gcc -O1 -g -std=gnu11 -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer -I$root/include -I$root/p264decoder_amd/csrc/host \
      -c $root/p264decoder_amd/csrc/host/$f.c -o $out/$f.o
done
gcc -O1 -g -fPIC -fsanitize=address,undefined -I$root/include -c $here/hip_stub.c -o $out/stub.o
gcc -shared -fsanitize=address,undefined -o $out/libp264amd_asan.so $out/parser.o $out/vlc.o $out/cabac.o $out/dropin.o $out/pipeline.o $out/fanout.o $out/input_layout.o $out/compact.o $out/cpu_check.o $out/stub.o -lpthread
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python3 $here/asan_host.py $out/libp264amd_asan.so 2>&1 | grep -v "^p264amd:" | tail -20
rm -rf $out
