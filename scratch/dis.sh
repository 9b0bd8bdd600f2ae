#!/bin/bash
# usage: scratch/dis.sh <lib.so> <outdir>   -> <outdir>/all.s (gfx950 disassembly of every kernel), <outdir>/<kernel>.s per kernel
lib=$1; out=$2
mkdir -p $out
d=$(mktemp -d)
cp $lib $d/l.so
(cd $d && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading l.so > /dev/null)
# (one code object per translation unit since round 6: all of them)
for o in $d/l.so.*gfx950; do /opt/rocm/lib/llvm/bin/llvm-objdump -d --no-show-raw-insn $o | sed 's/ *\/\/ [0-9A-F]*:.*$//'; done > $out/all.s
python3 - $out <<'PY'
import re, sys
out = sys.argv[1]
cur = None
for line in open(out + "/all.s"):
    m = re.match(r"^[0-9a-f]+ <(_Z\d+)?(k_[a-z_0-9]+?)(PK6PicDev|Ph|IL).*>:", line)
    if m:
        name = m.group(2) + ("_b1" if "ILb1" in line else "_b0" if "ILb0" in line else "")
        cur = open(out + "/" + name + ".s", "w")
    if cur:
        cur.write(line)
PY
rm -rf $d
ls $out
