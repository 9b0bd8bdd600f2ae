#!/usr/bin/env python3
"""One rank of a fan-out run (include/p264fan.h) over RCCL or TCP: rank 0 owns `--streams` copies of a 1080p stream (by default
BASELINE config 5's kind: Main profile, CABAC, I+P+B),
parses them, scatters the parsed pictures to the ranks owning the streams, gathers the I420 planes and checks every
picture against the reference decoder's committed hash.  bench.py starts one of these per GPU (as child processes, so
that a transport problem can never take the timed bench down); also usable by hand:

  python -m p264decoder_amd.tools.fan_bench --rank R --world N --transport rccl --uid <hex> --device D [--streams 8] [--pictures 6]
  python -m p264decoder_amd.tools.fan_bench --rank R --world N --transport tcp --port 29555
"""
import argparse
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rank", type=int, required=True)
    ap.add_argument("--world", type=int, required=True)
    ap.add_argument("--transport", default="rccl")
    ap.add_argument("--uid", default="")
    ap.add_argument("--host", default="127.0.0.1")
    ap.add_argument("--port", type=int, default=29555)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--streams", type=int, default=0, help="default: one per rank")
    ap.add_argument("--pictures", type=int, default=12)
    ap.add_argument("--workload", default="main_1080p_cabac_ipb", help="main_1080p_cabac_ipb: BASELINE config 5's streams (1080p Main, CABAC, I+P+B; checked "
                    "against the committed ORACLE hashes - the reference cannot decode them); cfg3_1080p_allp: Baseline all-P, checked against the reference's hashes")
    a = ap.parse_args()
    from p264decoder_amd import FanOut, _native
    from tests import synth_cases
    lib = _native.load()
    tr = ("rccl", bytes.fromhex(a.uid)) if a.transport == "rccl" else ("tcp", a.host, a.port)
    try:
        fan = FanOut(a.rank, a.world, tr, device=a.device, lib=lib)
    except Exception as e:                                     # the transport's own message (RCCL error string included)
        print("FANOUT " + json.dumps({"error": "rank %d: %s" % (a.rank, e), "transport": a.transport, "world": a.world}), flush=True)
        raise SystemExit(3)
    if a.rank:
        try:
            fan.worker()
        except Exception as e:
            print("FANOUT " + json.dumps({"error": "rank %d: %s" % (a.rank, e)}), flush=True)
            raise SystemExit(3)
        fan.close()
        return
    n = a.streams or a.world
    if a.workload in synth_cases.ORACLE_CASES:
        data = open(synth_cases.generate(synth_cases.ORACLE_CASES[a.workload]), "rb").read()
        hashes, pinned_by = synth_cases.oracle_golden(a.workload)[1], "oracle"
    else:
        data, hashes, pinned_by = synth_cases.stream_bytes(a.workload), synth_cases.golden(a.workload)[1], "reference"
    bad = []

    def on_frame(s, i, y, u, v):
        h = hashlib.sha256()
        for p in (y, u, v):
            h.update(p.tobytes())
        if h.hexdigest() != hashes[i]:
            bad.append((s, i))
    try:
        st = fan.root([data] * n, max_pictures=a.pictures, on_frame=on_frame)
    except Exception as e:
        print("FANOUT " + json.dumps({"error": "rank 0: %s" % e, "transport": a.transport, "world": a.world, "streams": n}), flush=True)
        raise SystemExit(3)
    fan.close()
    out = {"transport": a.transport, "world": a.world, "streams": n, "pictures": st["pictures"], "pictures_on_other_ranks": st["pictures_remote"],
           "frames_per_s": round(st["pictures"] / st["seconds"], 1), "seconds": round(st["seconds"], 3),
           "root_parse_seconds": round(st["parse_seconds"], 3), "root_parse_threads": st["parse_threads"],
           "parse_wait_seconds": round(st["parse_wait_seconds"], 3), "exchange_seconds": round(st["exchange_seconds"], 3),
           "root_reconstruct_seconds": round(st["reconstruct_seconds"], 3),
           "scattered_MB": round(st["bytes_scattered"] / 1e6, 2), "gathered_MB": round(st["bytes_gathered"] / 1e6, 2),
           "worker_rounds_on_the_device_road": st["device_road_rounds"],      # pictures into their input slots, planes out of the conversion buffers: no host bounce on the workers
           "workload": a.workload, "all_pictures_match_%s" % pinned_by: not bad,
           "what": "rank 0 parses every stream (one host thread per stream, the next round while the current one is exchanged), "
                   "scatters parsed pictures, gathers I420; parse_wait_seconds is the part of the parse that was not hidden"}
    print("FANOUT " + json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
