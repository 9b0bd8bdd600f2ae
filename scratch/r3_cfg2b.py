import sys, time, os
sys.path.insert(0, os.getcwd())
from p264decoder_amd import HipReconstructor, Parser
from tests import synth_cases
pics = Parser(quiet=True).parse_stream(synth_cases.stream_bytes("cfg2_720p_intra"))[:8]
T = len(pics); S = 1024
for iw, rb, ppw in (("", "", ""), ("2", "", ""), ("8", "", ""), ("16", "", ""), ("", "3", "1"), ("", "2", "2"), ("", "1", "4"), ("", "2", "1"), ("", "1", "2")):
    for k, v in (("P264AMD_INTRA_WAVES", iw), ("P264AMD_DEBLOCK_RB_LOG2", rb), ("P264AMD_DEBLOCK_PICS_PER_WG", ppw)):
        if v: os.environ[k] = v
        else: os.environ.pop(k, None)
    hip = HipReconstructor(80, 45, n_streams=S, slots=2, max_pictures=S * T)
    hip.upload(0, pics)
    for s in range(1, S):
        for t in range(T): hip.clone_picture(s * T + t, t)
    hip.sync()
    streams = list(range(S))
    hip.reconstruct([s * T for s in streams], streams); hip.sync()
    hip.timing_enable(True); hip.timing_reset()
    t0 = time.perf_counter()
    for rep in range(2):
        for t in range(T): hip.reconstruct([s * T + t for s in streams], streams)
    hip.sync()
    dt = time.perf_counter() - t0
    tm = hip.timing_read(); hip.timing_enable(False)
    print("intra_waves=%s rb=%s ppw=%s: %.0f fps;" % (iw, rb, ppw, 2 * S * T / dt), {k: round(v[0] / max(v[1], 1), 3) for k, v in tm.items()}, flush=True)
    hip.close()
