"""single-stream rate of the drop-in API (p264_nal_decode + p264_decoder_decode per NAL), 1080p I+P"""
import time
from p264decoder_amd import Decoder, _native
from tests import synth_cases
lib = _native.load()
data = synth_cases.stream_bytes("cfg3_1080p_ip")
for rep in range(2):
    dec = Decoder(lib=lib)
    t0 = time.perf_counter(); n = sum(1 for _ in dec.decode_annexb(data)); dt = time.perf_counter() - t0
    dec.close()
    print("drop-in 1080p: %d pictures, %.1f frames/s" % (n, n / dt))
