#!/usr/bin/env python3
"""End-to-end rate of the multi-stream pipeline (include/p264pipe.h): Annex-B bytes in host memory -> pictures in HBM,
host CAVLC parse included.  This is NOT bench.py's metric (which times the reconstruction with inputs resident in HBM);
it is the PCIe- and parse-inclusive figure DESIGN.md quotes next to it.

  python -m p264decoder_amd.tools.pipe_bench [--streams 64] [--threads 16] [--pictures 24] [--device 0|-1]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=64)
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 8)
    ap.add_argument("--pictures", type=int, default=24)
    ap.add_argument("--device", type=int, default=0, help="-1: parsers only")
    args = ap.parse_args()
    import bench
    from p264decoder_amd import Pipeline
    from tests import synth_cases
    distinct = [open(synth_cases.generate(bench.synth_args(args.pictures, 1000 + g)), "rb").read() for g in range(4)]
    pipe = Pipeline([distinct[i % 4] for i in range(args.streams)], threads=args.threads, device=args.device)
    pipe.run(max_pictures=2)                                      # warm-up: contexts, pinned buffers, first launches
    pipe.close()
    pipe = Pipeline([distinct[i % 4] for i in range(args.streams)], threads=args.threads, device=args.device)
    st = pipe.run()
    pipe.close()
    print(json.dumps({"metric": "end-to-end 1080p frames/sec (Annex-B in host memory -> pictures in HBM)",
                      "value": round(st["pictures"] / st["seconds"], 1), "unit": "frames/s",
                      "streams": args.streams, "threads": st["threads"], "pictures": st["pictures"], "device": args.device,
                      "parse_cpu_seconds": round(st["parse_seconds"], 3), "submit_seconds": round(st["submit_seconds"], 3),
                      "wall_seconds": round(st["seconds"], 3), "mbit_per_s": round(st["bytes"] * 8 / st["seconds"] / 1e6, 1)}))


if __name__ == "__main__":
    main()
