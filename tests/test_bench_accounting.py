"""bench.py's byte accounting (no GPU): the algorithmic bytes per stage follow SURVEY 8d, and moving the loop filter's
edge-info pass into the intra launch (the default in batches of P pictures; p264hip_last_launch says which launch ran it)
moves its 84 bytes per macroblock between the stages without changing their sum."""
import bench
from p264decoder_amd import Parser
from tests import synth_cases


def test_edge_info_bytes_follow_the_launch_that_does_the_work(lib):
    pics = Parser(quiet=True, lib=lib).parse_stream(synth_cases.stream_bytes("cif_ip"))[1:4]
    a = bench.algorithmic_bytes(pics, False)
    assert bench.stage_kernels(False)["deblock"] == "k_deblock_bs + k_deblock"
    b = bench.algorithmic_bytes(pics, True)
    assert "edge-info" in bench.stage_kernels(True)["intra"] and bench.stage_kernels(True)["deblock"] == "k_deblock"
    n_mb = sum(p.desc.mb_w * p.desc.mb_h for p in pics)
    assert a["deblock"] == n_mb * 852 and b["deblock"] == n_mb * 768
    assert b["intra"] - a["intra"] == n_mb * 84
    assert a["inter"] == b["inter"] and a["inter_read"] == b["inter_read"]
    assert sum(a[k] for k in ("inter", "intra", "deblock")) == sum(b[k] for k in ("inter", "intra", "deblock"))


def test_the_accounting_does_not_read_the_environment(monkeypatch):
    """(round 4 parsed P264AMD_BS_FUSED itself, differently from the library's atoi: a value like '' booked the edge-info bytes
    on the wrong stage.)  The library is asked instead."""
    import inspect
    src = inspect.getsource(bench)
    assert "P264AMD_BS_FUSED" not in src.replace('P264AMD_BS_FUSED=0 gives it its own launch', "")


def test_a_counter_summary_of_other_kernels_is_not_replayed(monkeypatch, capsys):
    """roofline.traffic is replayed from profiles/traffic_latest.json only while that summary belongs to THIS build's kernels
    (fingerprint over csrc/hip); after any kernel change it is reported as STALE (null + a line on stderr), never silently."""
    import json
    import os
    tj = json.load(open(os.path.join(bench.ROOT, "profiles", "traffic_latest.json")))
    S = int(tj["pictures_per_launch"])
    monkeypatch.setattr(bench, "kernel_fingerprint", lambda: tj["kernels_sha256"])
    t, src = bench.static_traffic("inter", S)
    assert t == tj["inter"] and src.startswith("static: profiles/")
    assert bench.static_traffic("inter", S + 1)[0] is None
    monkeypatch.setattr(bench, "kernel_fingerprint", lambda: "0" * 64)
    t, src = bench.static_traffic("inter", S)
    assert t is None and src.startswith("STALE") and "NOT replayed" in capsys.readouterr().err


def test_kernel_fingerprint_follows_the_sources(tmp_path, monkeypatch):
    a = bench.kernel_fingerprint()
    assert len(a) == 64 and a == bench.kernel_fingerprint()


def test_the_committed_counter_summary_belongs_to_these_kernels():
    """profiles/traffic_latest.json is the fall-back of roofline.traffic where the run cannot count for itself (N > 1, no
    rocprofv3): it must have been collected on the kernels in the tree (profiles/collect.sh + summarize.py after the last change
    under csrc/hip)."""
    import json
    import os
    tj = json.load(open(os.path.join(bench.ROOT, "profiles", "traffic_latest.json")))
    assert tj.get("kernels_sha256") == bench.kernel_fingerprint(), "csrc/hip changed since profiles/%s was collected: run profiles/collect.sh + summarize.py" % tj.get("source")
