#!/usr/bin/env python3
"""round 6: a race soak of the row-wavefront kernels at the bench's batch shapes (one column of lag per row is new this round): the golden
1080p all-P stream on S private clones, 30 pictures, EVERY stream's picture hashed against the reference decoder's hash at
pictures 9, 19 and 29 (an error in any picture of a stream stays in its reference chain).  usage: r6_soak_batch.py [S ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from p264decoder_amd import HipReconstructor, Parser, _native   # noqa: E402
from tests import synth_cases                                  # noqa: E402
from tests.conftest import frame_sha256                        # noqa: E402

lib = _native.load()
_, hashes = synth_cases.golden("cfg3_1080p_allp")
parser = Parser(quiet=True, lib=lib)
pics = parser.parse_stream(synth_cases.stream_bytes("cfg3_1080p_allp"), limit=30)
T = len(pics)
bad = 0
for S in [int(a) for a in sys.argv[1:]] or [2048, 256, 515, 1280]:
    t0 = time.time()
    hip = HipReconstructor(pics[0].mb_w, pics[0].mb_h, n_streams=S, slots=parser.slots, max_pictures=S * T, lib=lib)
    hip.upload(0, pics)
    for s in range(1, S):
        for t in range(T):
            hip.clone_picture(s * T + t, t)
    streams = list(range(S))
    checked = 0
    for t in range(T):
        hip.reconstruct([s * T + t for s in streams], streams)
        if t % 10 == 9:
            hip.sync()
            for s in streams:
                if frame_sha256(*hip.read_frame(s, pics[t].desc.dst_slot)) != hashes[t]:
                    bad += 1
                    print("MISMATCH S=%d picture %d stream %d" % (S, t, s), flush=True)
                checked += 1
            print("S=%d picture %d: all streams hashed (%d so far, %d mismatches, %.0f s)" % (S, t, checked, bad, time.time() - t0), flush=True)
    print("S=%d launch %s" % (S, hip.last_launch()), flush=True)
    hip.close()
print("mismatches:", bad)
