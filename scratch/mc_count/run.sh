#!/bin/bash
# static instruction counts of k_mc's roles per key: one build per (role, key) with the key forced (EXPM_FORCE_KEY), dead classes
# compiled out; prints vector / scalar / LDS / memory instructions of the whole kernel (prologue ~ 40 vector instructions)
cd /root/repo
names=(copy H V diag C CH CV gen)
count() { # role key label
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Ip264decoder_amd/csrc/hip -DP264AMD_TIMING_BUILD -DROLE=$1 -DEXPM_FORCE_KEY=$2 ${EXTRA} -S --cuda-device-only -o /tmp/mc_role_$1_$2.s scratch/mc_count/role.hip 2>/dev/null || { echo "build failed $1 $2"; return; }
  python3 - /tmp/mc_role_$1_$2.s "$3" <<'PY'
import re, sys, collections
c = collections.Counter()
inside = False
for l in open(sys.argv[1]):
    if re.match(r"^_Z6k_role.*:", l): inside = True; continue
    if inside and l.strip().startswith(".section"): inside = False
    if not inside: continue
    m = re.match(r"\s+([a-z_0-9]+)\s", l)
    if not m: continue
    op = m.group(1)
    if op.startswith("v_"): c["valu"] += 1; c["cheap"] += op.split("_e")[0] in ("v_add_u32","v_sub_u32","v_subrev_u32","v_and_b32","v_or_b32","v_xor_b32","v_lshrrev_b32","v_ashrrev_i32","v_mov_b32","v_cndmask_b32","v_add_u16","v_not_b32")
    elif op.startswith("s_waitcnt"): c["wait"] += 1
    elif op.startswith("s_"): c["salu"] += 1
    elif op.startswith("ds_"): c["lds"] += 1
    elif op.startswith(("buffer_load","global_load")): c["vld"] += 1
    elif op.startswith(("buffer_store","global_store")): c["vst"] += 1
m = re.search(r"vgpr_count:\s+(\d+)", open(sys.argv[1]).read())
print("%-28s valu %4d (cheap %3d)  salu %4d  lds %3d  loads %3d  stores %2d  waits %2d" % (sys.argv[2], c["valu"], c["cheap"], c["salu"], c["lds"], c["vld"], c["vst"], c["wait"]))
PY
}
for pc in 0 1 2 3 4 5 6; do for fl in 0 16; do count 0 $((pc+fl)) "luma MB ${names[$pc]} resid=$((fl/16))"; done; done
count 0 $((3+8)) "luma MB diag clamp"
for pc in 0 1 2 3 4 5 6; do count 1 $pc "luma quad ${names[$pc]}"; done
count 1 $((3+16)) "luma quad diag resid"
count 1 $((7+8)) "luma quad gen(sub8x8)"
for k in 0 2 1; do count 2 $k "chroma MB key $k"; done
for k in 0 2 1; do count 3 $k "chroma quad key $k"; done
