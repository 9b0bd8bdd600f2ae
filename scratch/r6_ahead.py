#!/usr/bin/env python3
"""round 6: the work-list sort on a side stream, ahead of the previous step's tail (P264AMD_SORT_AHEAD, p264hip.hip) - config 4 (I+P+B)
and config 2 rates of one process; the headline and batch_256 come from bench.py itself (scratch/r6_ahead.sh runs both settings)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                   # noqa: E402
from p264decoder_amd import Parser, _native                    # noqa: E402
from tests import synth_cases                                  # noqa: E402

lib = _native.load()
name = "main_1080p_cabac_ipb"
pics = Parser(quiet=True, lib=lib).parse_stream(open(synth_cases.generate(synth_cases.ORACLE_CASES[name]), "rb").read())
for i in range(2):
    stages = {}
    fps, digest = bench.run_batched(lib, pics, 1024, bench.MB_W, bench.MB_H, 3, stages if i else None)
    print("SORT_AHEAD=%s config4 %.0f frames/s ok=%s %s" % (os.environ.get("P264AMD_SORT_AHEAD", "default"), fps, digest == synth_cases.oracle_golden(name)[1][-1],
                                                            {k: (v["inter"], v["intra"], v["deblock"]) for k, v in stages.items()}), flush=True)
pics = Parser(quiet=True, lib=lib).parse_stream(synth_cases.stream_bytes("cfg3_1080p_ip"))
fps, digest = bench.run_batched(lib, pics, 1024, bench.MB_W, bench.MB_H, 2)
print("SORT_AHEAD=%s config3 I+P %.0f frames/s ok=%s" % (os.environ.get("P264AMD_SORT_AHEAD", "default"), fps, digest == synth_cases.golden("cfg3_1080p_ip")[1][-1]), flush=True)
