import sys, time, os
sys.path.insert(0, "/root/repo")
from p264decoder_amd import Pipeline, _native
from tests import synth_cases
args = sys.argv[1] if len(sys.argv) > 1 else "--mbw 120 --mbh 68 --frames 24 --gop 0 --seed 1000 --coded 12 --maxlevel 12 --crop-bottom 4"
lib = _native.load_library(os.environ["P264AMD_LIB"]) if os.environ.get("P264AMD_LIB") and hasattr(_native, "load_library") else None
data = open(synth_cases.generate(args), "rb").read()
best = 0
for rep in range(3):
    one = Pipeline([data], threads=1, device=-1, lib=lib) if lib else Pipeline([data], threads=1, device=-1)
    st = one.run(); one.close()
    best = max(best, st["pictures"] / st["seconds"])
print("%.1f fps single thread parse-only, %.2f MB / %d pictures" % (best, len(data) / 1e6, st["pictures"]))
