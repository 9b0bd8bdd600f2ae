"""In-tree build of libp264amd.so: host C (parser, drop-in API) with gcc, HIP kernels and the
C-ABI layer with hipcc for gfx950, linked by hipcc.  ``python -m p264decoder_amd.build``.

hipcc cross-compiles without a GPU, so this also is the "does it build" check
(__graft_entry__.build()).  The .so is git-ignored but travels to the GPU box.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
INC = os.path.join(ROOT, "include")
HOST_DIR = os.path.join(HERE, "csrc", "host")
HIP_DIR = os.path.join(HERE, "csrc", "hip")
OBJ_DIR = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libp264amd.so")
TOOLS_DIR = os.path.join(HERE, "tools")

HOST_SRCS = ["parser.c", "vlc.c", "cabac.c", "dropin.c", "pipeline.c", "fanout.c", "input_layout.c", "compact.c", "cpu_check.c"]
HOST_BASELINE = {"cpu_check.c"}      # built WITHOUT HOST_ARCH: its constructor looks at the CPU before any x86-64-v3 code runs
HIP_SRCS = ["p264hip.hip", "k_deblock.hip", "fan_rccl.hip"]
# options of single translation units (k_deblock.hip: the backend's max-ILP scheduling strategy - kernel_deblock.h says why)
HIP_EXTRA = {"k_deblock.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]}
HIP_ARCH = "gfx950"
# host code for AVX2 / BMI2 machines (every EPYC; the GPU boxes are Zen 5): the CABAC parse gains 4.5 % per thread, CAVLC
# nothing (scratch/r5_parse_flags.sh; -march=znver3, -O2 and a profile-guided build all lose on CAVLC).  p264parse_open refuses
# to run on an older CPU - and so does every other public entry of the host objects (cpu_check.c: a library constructor, the one
# object built for plain x86-64).  Needs gcc >= 11 (-march=x86-64-v3).
HOST_ARCH = ["-march=x86-64-v3", "-falign-functions=64"]     # (functions on cache-line boundaries: CAVLC + 1.5 % per thread, scratch/r5_parse_ab.sh)


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the gfx950 kernels cannot be built")


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("build step failed: %s\n%s" % (" ".join(cmd), r.stdout))
    return r.stdout


def _headers():
    hs = []
    for d in (INC, HOST_DIR, HIP_DIR):
        hs += [os.path.join(d, f) for f in os.listdir(d) if f.endswith(".h")]
    return hs


def build(force=False, verbose=False):
    os.makedirs(OBJ_DIR, exist_ok=True)
    hdrs = _headers()
    objs = []
    for s in HOST_SRCS:
        src = os.path.join(HOST_DIR, s)
        obj = os.path.join(OBJ_DIR, s + ".o")
        if force or _newer(obj, [src] + hdrs):
            out = _run(["gcc", "-O3"] + ([] if s in HOST_BASELINE else HOST_ARCH) + ["-std=gnu11", "-fPIC", "-Wall", "-Wextra", "-I" + INC, "-I" + HOST_DIR, "-c", src, "-o", obj])
            if verbose and out:
                print(out)
        objs.append(obj)
    hipcc = _hipcc()
    for s in HIP_SRCS:
        src = os.path.join(HIP_DIR, s)
        obj = os.path.join(OBJ_DIR, s + ".o")
        if force or _newer(obj, [src] + hdrs):
            out = _run([hipcc, "--offload-arch=" + HIP_ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall"] + HIP_EXTRA.get(s, []) +
                       ["-I" + INC, "-I" + HIP_DIR, "-c", src, "-o", obj])
            if verbose and out:
                print(out)
        objs.append(obj)
    if force or _newer(LIB, objs):
        _run([hipcc, "--offload-arch=" + HIP_ARCH, "-shared", "-o", LIB] + objs + ["-lpthread", "-ldl"])
    build_tools(force=force)
    return LIB


def build_tools(force=False):
    """Host-only helper programs (synthetic stream writer)."""
    built = []
    src = os.path.join(TOOLS_DIR, "synth264.c")
    exe = os.path.join(TOOLS_DIR, "synth264")
    if os.path.exists(src) and (force or _newer(exe, [src] + _headers())):
        _run(["gcc", "-O2", "-std=gnu11", "-Wall", "-Wextra", "-I" + INC, "-I" + HOST_DIR, src, "-o", exe])
    if os.path.exists(exe):
        built.append(exe)
    # command-line decoder on the drop-in API (same interface as the reference's `p264decoder -d`)
    src = os.path.join(TOOLS_DIR, "p264decoder_cli.c")
    exe = os.path.join(TOOLS_DIR, "p264decoder_amd")
    if os.path.exists(src) and os.path.exists(LIB) and (force or _newer(exe, [src, LIB] + _headers())):
        _run(["gcc", "-O2", "-std=gnu11", "-Wall", "-Wextra", "-I" + INC, src, "-o", exe,
              "-L" + HERE, "-lp264amd", "-Wl,-rpath,$ORIGIN/.."])
    if os.path.exists(exe):
        built.append(exe)
    return built


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
