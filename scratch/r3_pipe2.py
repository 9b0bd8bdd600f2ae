import sys, os
sys.path.insert(0, os.getcwd())
from p264decoder_amd import Pipeline
from tests import synth_cases
args = "--mbw 120 --mbh 68 --frames 24 --gop 0 --seed %d --coded 12 --maxlevel 12 --crop-bottom 4"
distinct = [open(synth_cases.generate(args % (1000 + g)), "rb").read() for g in range(4)]
for threads, device in ((16, -1), (16, 0), (15, 0), (14, 0), (16, 0)):
    pipe = Pipeline([distinct[i % 4] for i in range(128)], threads=threads, device=device)
    st = pipe.run(); pipe.close()
    print("threads %d device %d: %.0f fps" % (threads, device, st["pictures"] / st["seconds"]), {k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.items()}, flush=True)
