"""Registers, scratch and LDS of every gfx950 kernel in libp264amd.so, read from the code object's notes.

    python -m p264decoder_amd.tools.kernel_resources [lib.so]

What the binary holds, not what the source hopes for: DESIGN.md's occupancy figures and
tests/test_kernel_resources.py ("no kernel of the hot path spills") both come from here.
Needs llvm-objdump / llvm-readelf of the ROCm toolchain (present wherever hipcc is).
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM_BIN = "/opt/rocm/lib/llvm/bin"
FIELDS = ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
          "group_segment_fixed_size", "max_flat_workgroup_size")


def _tool(name):
    for cand in (os.path.join(LLVM_BIN, name), shutil.which(name)):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("%s not found (ROCm LLVM tools)" % name)


def short_name(mangled):
    """_Z9k_deblockPK6PicDev... -> k_deblock; template instances keep their argument: k_deblock_bs<true>."""
    m = re.match(r"_Z\d+(k_[a-z0-9_]+?)(ILb([01])E)?(Ev)?P", mangled)
    if not m:
        return mangled
    return m.group(1) + ("" if m.group(3) is None else "<%s>" % ("true" if m.group(3) == "1" else "false"))


def kernel_resources(lib_path):
    """{kernel name: {field: int}} for the gfx950 code objects bundled in `lib_path`."""
    tmp = tempfile.mkdtemp(prefix="p264res_")
    try:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(lib_path, local)
        subprocess.run([_tool("llvm-objdump"), "--offloading", "lib.so"], cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        objs = [f for f in os.listdir(tmp) if "gfx950" in f]
        if not objs:
            raise RuntimeError("no gfx950 code object in %s" % lib_path)
        # (one code object per translation unit: p264hip.hip, k_deblock.hip - built with its own options since round 6 -, fan_rccl.hip)
        notes = "\n".join(subprocess.run([_tool("llvm-readelf"), "--notes", o], cwd=tmp, check=True, stdout=subprocess.PIPE, text=True).stdout for o in sorted(objs))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    out = {}
    cur = None
    # the metadata lists every kernel as a YAML map whose keys come in alphabetical order: `.name` sits in the middle, so the
    # fields of one kernel are gathered until the next `- .agpr_count` / `- .args` entry starts
    for line in notes.splitlines():
        if re.match(r"\s*-\s+\.(agpr_count|args):", line):
            cur = {}
        m = re.match(r"\s*(?:-\s+)?\.([a-z_]+):\s+(\S+)\s*$", line)
        if cur is None or not m:
            continue
        key, val = m.group(1), m.group(2)
        if key == "name":
            out[short_name(val)] = cur
        elif key in FIELDS:
            cur[key] = int(val)
    return {k: v for k, v in out.items() if k.startswith("k_")}


def main():
    from p264decoder_amd import _native
    lib = sys.argv[1] if len(sys.argv) > 1 else _native.LIB_PATH
    res = kernel_resources(lib)
    print("%-22s %5s %5s %6s %6s %8s %8s" % ("kernel", "vgpr", "sgpr", "vspill", "sspill", "scratch", "lds"))
    for k in sorted(res):
        r = res[k]
        print("%-22s %5d %5d %6d %6d %8d %8d" % (k, r.get("vgpr_count", -1), r.get("sgpr_count", -1), r.get("vgpr_spill_count", -1), r.get("sgpr_spill_count", -1),
                                                  r.get("private_segment_fixed_size", -1), r.get("group_segment_fixed_size", -1)))


if __name__ == "__main__":
    main()
