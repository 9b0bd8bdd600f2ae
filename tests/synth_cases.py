"""Synthetic workloads (BASELINE.json configs 2-3 and small edge cases), written by our own stream
writer p264decoder_amd/tools/synth264 inside the subset the reference decodes correctly.

Streams are NOT stored: they are a pure function of the arguments below.  What is committed
(tests/golden/synth_<name>.sha256) is the SHA-256 of the stream itself (generator determinism)
followed by the per-picture SHA-256 of the REAL reference decoder's output for that stream,
produced by tests/golden/make_golden.py from oracle/_ref.
"""
import hashlib
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
TOOL = os.path.join(ROOT, "p264decoder_amd", "tools", "synth264")
CACHE = os.path.join("/tmp", "p264amd_streams_%d" % os.getuid())

# name -> (synth264 arguments, pictures checked by the CPU suite (None = all))
CASES = {
    # BASELINE config 2: 1280x720, Baseline CAVLC, I slices only (intra + IDCT path)
    "cfg2_720p_intra": ("--mbw 80 --mbh 45 --frames 30 --intra-only --seed 2 --coded 25 --maxlevel 32", 6),
    # BASELINE config 3: 1920x1080 (coded 1088, crop flag set and ignored like the reference does), I+P, GOP 30.
    # --maxlevel 12 keeps the centre half-pel samples inside the reference's clip LUT (core/clip1.h);
    # beyond it the reference reads past its table (undefined), see DESIGN.md "A-Q13".
    "cfg3_1080p_ip": ("--mbw 120 --mbh 68 --frames 60 --gop 30 --seed 3 --coded 12 --maxlevel 12 --crop-bottom 4", 4),
    # config 3 with the level range SURVEY 8d gives it (+-32): seed 305 is one whose 30 pictures contain no centre half-pel
    # sample outside the reference's clip table (A-Q13; found by search, the oracle counts such samples and the CPU test
    # asserts there are none) - so the reference is defined on it and pins it
    "cfg3_1080p_ip_l32": ("--mbw 120 --mbh 68 --frames 30 --gop 30 --seed 305 --coded 12 --maxlevel 32 --crop-bottom 4", 3),
    # the throughput stream of the north star: 1 IDR + P pictures only
    "cfg3_1080p_allp": ("--mbw 120 --mbh 68 --frames 30 --gop 0 --seed 33 --coded 12 --maxlevel 12 --crop-bottom 4", 3),
    # the same stream at the length SURVEY 8d specifies for it: 1 IDR + 299 P pictures (the first 30 are the stream above);
    # bench.py takes its golden pictures from here when --warmup + --steps go past 29
    "cfg3_1080p_allp_300": ("--mbw 120 --mbh 68 --frames 300 --gop 0 --seed 33 --coded 12 --maxlevel 12 --crop-bottom 4", 2),
    # small cases for the edges of the arithmetic
    "cif_ip": ("--mbw 22 --mbh 18 --frames 24 --gop 8 --seed 7 --coded 20 --maxlevel 12", None),
    "tiny_1x1": ("--mbw 1 --mbh 1 --frames 10 --gop 5 --seed 11 --coded 40 --maxlevel 8", None),
    "row_1xN": ("--mbw 9 --mbh 1 --frames 8 --gop 4 --seed 12 --coded 30 --maxlevel 8", None),
    "col_Nx1": ("--mbw 1 --mbh 7 --frames 8 --gop 4 --seed 13 --coded 30 --maxlevel 8", None),
    "wide_70": ("--mbw 70 --mbh 3 --frames 6 --gop 3 --seed 14 --coded 20 --maxlevel 8", None),
    "qp0": ("--mbw 6 --mbh 5 --frames 6 --gop 3 --seed 20 --qp 0 --coded 30 --maxlevel 32", None),
    "qp12": ("--mbw 6 --mbh 5 --frames 6 --gop 3 --seed 21 --qp 12 --coded 30 --maxlevel 32", None),
    "qp38": ("--mbw 6 --mbh 5 --frames 6 --gop 3 --seed 22 --qp 38 --coded 30 --maxlevel 4", None),
    "qp51": ("--mbw 6 --mbh 5 --frames 6 --gop 3 --seed 23 --qp 51 --coded 30 --maxlevel 1", None),
    "cqo_neg": ("--mbw 6 --mbh 5 --frames 6 --gop 3 --seed 24 --qp 30 --cqo -7 --coded 30 --maxlevel 8", None),
    "cqo_pos": ("--mbw 6 --mbh 5 --frames 6 --gop 3 --seed 25 --qp 40 --cqo 9 --coded 30 --maxlevel 3", None),
    "nodeblock": ("--mbw 8 --mbh 6 --frames 8 --gop 4 --seed 26 --nodeblock --coded 25 --maxlevel 12", None),
    "dense": ("--mbw 8 --mbh 6 --frames 8 --gop 4 --seed 27 --coded 90 --maxlevel 6", None),
    # per-macroblock QP (mb_qp_delta; the reference adds it to the slice QP and carries the last QP over skipped /
    # residual-free macroblocks and across pictures, SURVEY A-Q2) and non-zero deblocking offsets (used unshifted, A-Q3)
    "qpdelta": ("--mbw 11 --mbh 9 --frames 12 --gop 6 --seed 51 --qp 28 --qp-delta 6 --coded 25 --maxlevel 6", None),
    "qpdelta_wide": ("--mbw 8 --mbh 6 --frames 10 --gop 5 --seed 52 --qp 26 --qp-delta 25 --coded 30 --maxlevel 2", None),
    "dboffs_pos": ("--mbw 11 --mbh 9 --frames 10 --gop 5 --seed 53 --qp 30 --deblock-offsets 6 5 --coded 20 --maxlevel 8", None),
    "dboffs_neg": ("--mbw 11 --mbh 9 --frames 10 --gop 5 --seed 54 --qp 34 --deblock-offsets -6 -4 --coded 20 --maxlevel 6", None),
    "qpd_dbo": ("--mbw 12 --mbh 7 --frames 10 --gop 5 --seed 55 --qp 30 --qp-delta 8 --deblock-offsets 3 -2 --cqo 4 --coded 25 --maxlevel 5", None),
    "qpd_1080p": ("--mbw 120 --mbh 68 --frames 4 --gop 0 --seed 56 --qp 27 --qp-delta 5 --deblock-offsets 2 1 --coded 12 --maxlevel 8 --crop-bottom 4", 2),
    # larger than 8192 macroblocks: the work-list sort classifies twice instead of keeping its macroblocks in registers, nine
    # bands of work lists, 17 deblocking bands of eight rows for a single picture
    "uhd_2160p_allp": ("--mbw 240 --mbh 135 --frames 3 --gop 0 --seed 61 --qp 29 --qp-delta 3 --coded 12 --maxlevel 10", 2),
    "mv_far": ("--mbw 10 --mbh 8 --frames 10 --gop 10 --seed 28 --mvmax 64 --coded 5 --maxlevel 6", None),
}
# Streams the reference cannot decode (B pictures): pinned to the CPU ORACLE's output instead (tests/golden/oracle_<name>.sha256,
# made by tests/golden/make_oracle_golden.py) - parity with the reference is unpinned for them
ORACLE_CASES = {
    # what BASELINE config 4 is with the CAVLC entropy coder: 1920x1088 Main profile, I + P + B (two B pictures between
    # reference pictures), spatial direct prediction, implicit weights, deblocking
    "main_1080p_ipb": "--mbw 120 --mbh 68 --frames 13 --seed 91 --refs 2 --bframes 2 --implicit --d8inf --coded 12 --maxlevel 12 --crop-bottom 4",
    # BASELINE config 4 itself: the same pictures with the CABAC entropy coder (1920x1080 Main profile, CABAC + B pictures + deblocking)
    "main_1080p_cabac_ipb": "--mbw 120 --mbh 68 --frames 13 --seed 91 --refs 2 --bframes 2 --implicit --d8inf --coded 12 --maxlevel 12 --crop-bottom 4 --cabac",
}


def oracle_golden(name):
    """(stream sha256, [per-picture sha256 of the oracle's output ...])"""
    lines = open(os.path.join(GOLDEN, "oracle_%s.sha256" % name)).read().split()
    return lines[0], lines[1:]


BIG = ("cfg2_720p_intra", "cfg3_1080p_ip", "cfg3_1080p_ip_l32", "cfg3_1080p_allp", "cfg3_1080p_allp_300", "qpd_1080p", "uhd_2160p_allp")


def ensure_tool():
    if not os.path.exists(TOOL):
        from p264decoder_amd import build
        build.build_tools()
    return TOOL


def generate(name, extra_args=None):
    """Write (or reuse) the stream of a case; returns its path."""
    ensure_tool()
    os.makedirs(CACHE, exist_ok=True)
    args = CASES[name][0] if name in CASES else name
    if extra_args:
        args = args + " " + extra_args
    tag = hashlib.sha1(args.encode()).hexdigest()[:16]
    path = os.path.join(CACHE, "%s.264" % tag)
    if not os.path.exists(path):
        tmp = path + ".tmp%d" % os.getpid()
        subprocess.run([TOOL, tmp] + args.split(), check=True)
        os.replace(tmp, path)
    return path


def stream_bytes(name):
    return open(generate(name), "rb").read()


def golden(name):
    """(stream sha256, [per-picture sha256 ...]) from tests/golden/synth_<name>.sha256"""
    lines = open(os.path.join(GOLDEN, "synth_%s.sha256" % name)).read().split()
    return lines[0], lines[1:]


def regenerate_golden(ref_hashes, write_hashes):
    """Called by tests/golden/make_golden.py (needs the real reference, build container only)."""
    for name in CASES:
        path = generate(name)
        digest = hashlib.sha256(open(path, "rb").read()).hexdigest()
        rows = ref_hashes(path)
        with open(os.path.join(GOLDEN, "synth_%s.sha256" % name), "w") as f:
            f.write(digest + "\n")
            for r in rows:
                f.write(r[1] + "\n")
        print("%-18s %3d pictures  stream %s" % (name, len(rows), digest[:16]))
