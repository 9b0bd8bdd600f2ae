"""p264decoder_amd - MI355X-native (gfx950, hand-written HIP) macroblock reconstruction for the
p264 H.264 decoder, behind the reference's own decode API.

Layers (each a C ABI in include/*.h, served by p264decoder_amd/libp264amd.so):
  p264_dropin.h  p264_param_default / p264_nal_decode / p264_decoder_open|decode|close
  p264parse.h    host CAVLC bitstream layer -> per-picture SoA buffers
  p264hip.h      HIP kernels: inter prediction + residual, intra wavefront, deblocking wavefront
  p264pipe.h     multi-stream pipeline: threaded host parse feeding batched reconstruction
  p264fan.h      stream fan-out: one rank owns inputs and outputs, pictures scattered / planes gathered (RCCL or TCP)

Importing this package never computes on the CPU what the GPU is meant to compute; if the
shared library is missing, `_native.load()` raises.
"""
from . import _native
from .recon import HipReconstructor, ParsedPicture, Parser, P264Error, device_count
from .decoder import Decoder, param_default
from .pipeline import Pipeline
from .fanout import FanOut

__all__ = ["Decoder", "param_default", "Pipeline", "FanOut", "Parser", "ParsedPicture", "HipReconstructor", "P264Error",
           "device_count", "_native"]
