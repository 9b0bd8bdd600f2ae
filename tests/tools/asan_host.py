import sys, ctypes as C
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from p264decoder_amd import _native as N, Parser, Pipeline
lib = N.load(sys.argv[1])
from tests import synth_cases
data = open(__import__('os').path.join(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))), 'golden', 'f26.264'),'rb').read()
pics = Parser(quiet=True, lib=lib).parse_stream(data)
print("f26 pictures", len(pics))
for name in ["cif_ip","--mbw 6 --mbh 5 --frames 12 --gop 6 --seed 61 --coded 20 --maxlevel 6 --pps-alt","tiny_1x1","row_1xN","col_Nx1","wide_70","qp51","dense","mv_far","cqo_neg"]:
    p = Parser(quiet=True, lib=lib).parse_stream(synth_cases.stream_bytes(name) if name in synth_cases.CASES else open(synth_cases.generate(name), "rb").read())
    print(name, len(p))
# truncated / corrupted streams must not crash
import random
random.seed(3)
for trial in range(60):
    d = bytearray(synth_cases.stream_bytes("cif_ip"))
    for _ in range(random.randrange(1, 20)):
        d[random.randrange(40, len(d))] = random.randrange(256)
    d = bytes(d[:random.randrange(100, len(d))])
    try:
        Parser(quiet=True, lib=lib).parse_stream(d)
    except Exception as e:
        pass
print("fuzz ok")
# B streams (two lists, direct prediction from stored co-located motion, implicit weights), whole and damaged
BARGS = ["--mbw 8 --mbh 6 --frames 16 --seed 82 --refs 3 --bframes 3 --sub8x8 --implicit --coded 10 --maxlevel 6",
         "--mbw 8 --mbh 6 --frames 16 --seed 104 --refs 2 --bframes 2 --sub8x8 --implicit --coded 12 --maxlevel 8 --cabac",
         "--mbw 9 --mbh 7 --frames 8 --gop 4 --seed 101 --coded 25 --maxlevel 40 --qp-delta 6 --cabac",
         "--mbw 7 --mbh 5 --frames 13 --seed 85 --refs 4 --bframes 3 --temporal --d8inf --implicit --slices 2 --coded 8 --maxlevel 6"]
for a in BARGS:
    bs = open(synth_cases.generate(a), "rb").read()
    print("B stream", len(Parser(quiet=True, lib=lib).parse_stream(bs)))
    for trial in range(60):
        d = bytearray(bs)
        for _ in range(random.randrange(1, 20)):
            d[random.randrange(30, len(d))] = random.randrange(256)
        d = bytes(d[:random.randrange(100, len(d))])
        try:
            Parser(quiet=True, lib=lib).parse_stream(d)
        except Exception as e:
            pass
print("B fuzz ok")
# the same kind of damage through the C start-code scanner and emulation-prevention strip (pipeline, parse only)
for trial in range(40):
    d = bytearray(synth_cases.stream_bytes("cif_ip"))
    for _ in range(random.randrange(1, 30)):
        d[random.randrange(0, len(d))] = random.choice([0, 0, 1, 3, random.randrange(256)])
    d = bytes(d[:random.randrange(4, len(d))])
    try:
        pipe = Pipeline([d, d[: len(d) // 2]], threads=2, device=-1, lib=lib)
        try:
            pipe.run()
        except RuntimeError:
            pass
        pipe.close()
    except RuntimeError:
        pass
print("pipeline fuzz ok")
# picture size changing in mid-stream: the context is rebuilt while the previous picture's buffers are still held
a = open(synth_cases.generate("--mbw 6 --mbh 5 --frames 12 --gop 6 --seed 61 --coded 20 --maxlevel 6"), "rb").read()
b = open(synth_cases.generate("--mbw 8 --mbh 6 --frames 6 --gop 3 --seed 62 --coded 20 --maxlevel 6"), "rb").read()
print("size switch", len(Parser(quiet=True, lib=lib).parse_stream(a + b + a + b)))
pipe = Pipeline([a + a, b + b, a], threads=3, device=-1, lib=lib)
print(pipe.run()["pictures"]); pipe.close()
# crafted parameter sets with fields far outside their ranges
from tests.tools.bitwriter import BitWriter
for frame_bits, poc_bits, w_mb, h_mb, refs in [(60, 4, 6, 5, 1), (4, 70, 6, 5, 1), (4, 4, 1 << 20, 5, 1), (4, 4, 6, 1 << 25, 1), (4, 4, 6, 5, 1 << 30), (4, 4, 0xffffffff, 0xffffffff, 0xffffffff)]:
    w = BitWriter()
    w.u(8, 66); w.u(8, 0xc0); w.u(8, 40); w.ue(0); w.ue(frame_bits - 4); w.ue(0); w.ue(poc_bits - 4); w.ue(refs); w.u(1, 0)
    w.ue((w_mb - 1) & 0xffffffff); w.ue((h_mb - 1) & 0xffffffff); w.u(1, 1); w.u(1, 1); w.u(1, 0); w.u(1, 0); w.trailing()
    p = Parser(quiet=True, lib=lib)
    try:
        p.feed(7, 3, w.bytes())
    except Exception:
        pass
    for typ, idc, rbsp in list(N.split_annexb(lib, synth_cases.stream_bytes("cif_ip")))[1:4]:      # PPS + slices against the bad SPS
        try:
            p.feed(typ, idc, rbsp)
        except Exception:
            pass
print("crafted SPS ok")
pipe = Pipeline([data, synth_cases.stream_bytes("cif_ip")]*3, threads=4, device=-1, lib=lib)
print(pipe.run()); pipe.close()
# the fan-out's root loop, producer threads, packing (p264hip_pack_input) and the unpacked views, one rank, the CPU oracle
# plugged in as the backend (TEST INFRASTRUCTURE); then two ranks over TCP in threads of this process
import threading
from p264decoder_amd.fanout import FanOut
from tests import fan_helpers
cif = synth_cases.stream_bytes("cif_ip")
bst = open(synth_cases.generate("--mbw 22 --mbh 18 --frames 7 --seed 5 --refs 2 --bframes 2 --implicit --d8inf --coded 8 --maxlevel 8"), "rb").read()
fan = FanOut(0, 1, None, backend=fan_helpers.oracle_backend(), lib=lib)
print("fan-out, one rank:", fan.root([cif, bst, cif], max_pictures=7)["pictures"])
fan.close()
res = {}
def rank(r):
    f = FanOut(r, 2, ("tcp", "127.0.0.1", 29871), backend=fan_helpers.oracle_backend(), lib=lib)
    if r == 0:
        res["st"] = f.root([cif, bst, bst, cif], max_pictures=7)
    else:
        f.worker()
    f.close()
ts = [threading.Thread(target=rank, args=(r,)) for r in (0, 1)]
[t.start() for t in ts]; [t.join(120) for t in ts]
print("fan-out, two ranks:", res["st"]["pictures"], res["st"]["pictures_remote"])
