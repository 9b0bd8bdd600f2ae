#!/usr/bin/env python3
"""Per-picture SHA-256 of the CPU ORACLE (not of the reference: it cannot decode these streams) for the Main-profile B-picture
workload below - what BASELINE config 4 looks like with the CAVLC entropy coder: 1920x1088, I + P + B pictures.  The hashes let
a box without the oracle's slow decode (bench.py's extras) check the HIP output; tests/test_bslices.py compares HIP and oracle
directly on small streams, tests/test_gpu_main_profile.py on this one.  Parity with the reference is UNPINNED for B pictures
(SURVEY 8c): these hashes pin the product to the oracle only."""
import hashlib
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from p264decoder_amd import Parser                      # noqa: E402
from tests import oracle_bind, synth_cases             # noqa: E402
from tests.conftest import frame_sha256                # noqa: E402

for name, args in synth_cases.ORACLE_CASES.items():
    data = open(synth_cases.generate(args), "rb").read()
    parser = Parser(quiet=True)
    pics = parser.parse_stream(data)
    ora = oracle_bind.load()
    store = oracle_bind.FrameStore(pics[0].mb_w, pics[0].mb_h, parser.slots)
    with open(os.path.join(HERE, "oracle_%s.sha256" % name), "w") as f:
        f.write(hashlib.sha256(data).hexdigest() + "\n")
        for p in pics:
            f.write(frame_sha256(*oracle_bind.reconstruct(ora, store, p)) + "\n")
    print(name, len(pics), "pictures")
