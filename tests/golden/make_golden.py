#!/usr/bin/env python3
"""Regenerates the golden fixtures under tests/golden/ by RUNNING THE REAL REFERENCE here
(oracle/_ref, built from /root/reference by `make -C oracle ref`).  Only the outputs - hashes and
small packed vectors - are committed; no reference source is.

  f26_frames.sha256        per-frame SHA-256 of the reference's MB-aligned I420 output for
                           bin/f26.264 (the reference's only test asset, kept as tests/golden/f26.264)
  synth_*.sha256           the same for the synthetic streams written by p264decoder_amd/tools/synth264
                           (streams are regenerated deterministically from their seed, not stored)
  kat_*.npz                kernel-level known-answer vectors through the reference's function tables
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
DRIVER = os.path.join(ROOT, "oracle", "_ref", "p264ref_driver")


def ref_hashes(stream_path):
    out = subprocess.run([DRIVER, "hash", stream_path], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True, text=True).stdout
    return [l.split() for l in out.strip().splitlines()]


def write_hashes(name, rows):
    with open(os.path.join(HERE, name), "w") as f:
        for idx, digest, w, h in rows:
            f.write("%s %s %s %s\n" % (idx, digest, w, h))


def main():
    if not os.path.exists(DRIVER):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True)
    write_hashes("f26_frames.sha256", ref_hashes(os.path.join(HERE, "f26.264")))
    sys.path.insert(0, ROOT)
    try:
        from tests import synth_cases
    except Exception as e:  # synthetic cases are added later in the build
        print("no synthetic cases yet:", e)
        return
    synth_cases.regenerate_golden(ref_hashes, write_hashes)


if __name__ == "__main__":
    main()
