import sys, ctypes as C
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from p264decoder_amd import _native as N, Parser, Pipeline
lib = N.load(sys.argv[1])
from tests import synth_cases
data = open(__import__('os').path.join(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))), 'golden', 'f26.264'),'rb').read()
pics = Parser(quiet=True, lib=lib).parse_stream(data)
print("f26 pictures", len(pics))
for name in ["cif_ip","tiny_1x1","row_1xN","col_Nx1","wide_70","qp51","dense","mv_far","cqo_neg"]:
    p = Parser(quiet=True, lib=lib).parse_stream(synth_cases.stream_bytes(name))
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
pipe = Pipeline([data, synth_cases.stream_bytes("cif_ip")]*3, threads=4, device=-1, lib=lib)
print(pipe.run()); pipe.close()
