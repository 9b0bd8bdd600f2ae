"""bench.py's byte accounting (no GPU): the algorithmic bytes per stage follow SURVEY 8d, and moving the loop filter's
edge-info pass into the intra launch (P264AMD_BS_FUSED, the default) moves its 84 bytes per macroblock between the stages
without changing their sum."""
import importlib
import sys

from p264decoder_amd import Parser
from tests import synth_cases


def _bench(monkeypatch, fused):
    monkeypatch.setenv("P264AMD_BS_FUSED", fused)
    sys.modules.pop("bench", None)
    return importlib.import_module("bench")


def test_edge_info_bytes_follow_the_launch_that_does_the_work(lib, monkeypatch):
    pics = Parser(quiet=True, lib=lib).parse_stream(synth_cases.stream_bytes("cif_ip"))[1:4]
    own = _bench(monkeypatch, "0")
    a = own.algorithmic_bytes(pics)
    assert own.STAGE_KERNELS["deblock"] == "k_deblock_bs + k_deblock"
    fused = _bench(monkeypatch, "1")
    b = fused.algorithmic_bytes(pics)
    assert "edge-info" in fused.STAGE_KERNELS["intra"] and fused.STAGE_KERNELS["deblock"] == "k_deblock"
    n_mb = sum(p.desc.mb_w * p.desc.mb_h for p in pics)
    assert a["deblock"] == n_mb * 852 and b["deblock"] == n_mb * 768
    assert b["intra"] - a["intra"] == n_mb * 84
    assert a["inter"] == b["inter"] and a["inter_read"] == b["inter_read"]
    assert sum(a[k] for k in ("inter", "intra", "deblock")) == sum(b[k] for k in ("inter", "intra", "deblock"))
    sys.modules.pop("bench", None)
