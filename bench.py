#!/usr/bin/env python3
"""bench.py - throughput of the MI355X macroblock-reconstruction hot path.

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): synthetic
1920x1080 (coded 1920x1088, 8160 MBs) Baseline CAVLC stream, 1 IDR + P pictures only
(the "all-P-slice" stream of the north star), written by tools/synth264.  S independent
streams are decoded side by side on one GPU - consecutive P pictures of one stream depend on
each other, independent streams are the parallel axis.  A *step* = one pass of the hot path
(inter prediction + residual, intra, deblocking) over one batch = the next picture of each of
the S streams.  All parsed inputs (the host CAVLC parse is CPU work by design) are resident in
HBM before the timed region; every stream has its own private copy of its inputs and its own
frame store.  The IDR picture and W P pictures are the untimed warm-up, then exactly K P
pictures per stream are timed.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--streams S]
  N>1: one rank per GPU, streams sharded across ranks, no data-path collective (weak scaling).  Either the caller starts the
       ranks (python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...: RANK / WORLD_SIZE in the
       environment) or bench.py does: `python bench.py --gpus N` without WORLD_SIZE starts that same command as a child
       process - the parent never touches the GPU - and relays rank 0's JSON line.  A WORLD_SIZE that differs from --gpus
       is an error (exit 2), never a silent one-rank run.

Prints ONE JSON line (rank 0).  Besides the contract's fields: `roofline` (the dominant stage), `kernels` (every stage with
its own fraction of the HBM roofline), `cpu_baseline` (the real reference decoder on one host core, N=1 only), a
`golden_check` (the timed output of a golden-seeded stream hashed against the committed reference hash) and `extras`
(never `value`: config 2, config 3 I+P, the parse- and PCIe-inclusive pipeline, the single-stream drop-in API; N=1 only).
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
MB_W, MB_H = 120, 68
N_MB = MB_W * MB_H
DISTINCT = 4                   # distinct synthetic streams per rank; the S streams cycle through private copies of them
def stage_kernels(edge_info_fused):
    """Which kernels a stage's time covers.  Round 4: in batches of P pictures the loop filter's edge-info pass (the boundary
    strengths: 84 B of records and vectors per macroblock in) runs as extra workgroups of the k_intra_sparse launch, where it
    overlaps the intra work (kernel_intra.h); P264AMD_BS_FUSED=0 gives it its own launch (k_deblock_bs) back.  Which of the two
    ran is asked of the library (p264hip_last_launch), not guessed from the environment."""
    k = {"inter": "k_mc_sort + k_mc", "intra": "k_intra_sparse",      # (the timed launches hold P pictures only: the sparse build of k_intra)
         "deblock": "k_deblock_bs + k_deblock"}
    if edge_info_fused:
        k.update(intra="k_intra_sparse (intra macroblocks + the loop filter's edge-info pass as extra workgroups)", deblock="k_deblock")
    return k


def synth_args(frames, seed):
    return "--mbw %d --mbh %d --frames %d --gop 0 --seed %d --coded 12 --maxlevel 12 --crop-bottom 4" % (MB_W, MB_H, frames, seed)


def algorithmic_bytes(pics, edge_info_fused=True):
    """Bytes that must cross HBM once per launch, per stage (DESIGN.md 'Roofline').  The edge-info pass's input is
    booked on the stage whose launch does the work (edge_info_fused: inside the intra launch).  The MC figure is SURVEY 8d's
    836 B per inter macroblock (384 B reference + 64 B motion + 4 B type read, 384 B written), its read part 452 B; the residual
    input the MC kernels also consume (16 B record + 32 B per coded block, the seam's dense block format) is reported
    separately as `inter_with_residual` and never enters `roofline.frac`."""
    import numpy as np
    inter = inter_read = inter_resid = intra = deblock = 0
    for p in pics:
        rec = p.mb_records()
        is_intra = rec["mb_type"] <= 2
        blocks = np.array([bin(int(m) & 0x3ffffff).count("1") for m in rec["coef_mask"]])
        n_inter = int((~is_intra).sum())
        n_intra = int(is_intra.sum())
        inter += n_inter * 836
        inter_read += n_inter * 452
        inter_resid += n_inter * (836 + 16) + int(blocks[~is_intra].sum()) * 32
        # intra: 384 B written + modes (16 B) + MB record (16 B) + coded blocks
        intra += n_intra * (384 + 32) + int(blocks[is_intra].sum()) * 32
        # deblock: 384 B read + 384 B written + side tables (16 B record, 64 B motion, 4 B refs).  With the edge-info pass inside the
        # intra launch the side tables are read there; what crosses HBM for them here is then the 16 B of edge info per macroblock
        # written by that pass and read by k_deblock (not algorithmic: left out on both sides)
        deblock += len(rec) * (768 + (0 if edge_info_fused else 84))
        if edge_info_fused:
            intra += len(rec) * 84
    return {"inter": inter, "intra": intra, "deblock": deblock, "inter_read": inter_read, "inter_with_residual": inter_resid}


def cpu_quota():
    """CPUs this process may use: the cgroup quota where there is one (cpu.max), else the visible CPUs."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            return max(1, int(int(q) / int(per)))
    except (OSError, ValueError):
        pass
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 8


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def run_batched(lib, pics, S, mb_w, mb_h, slots, stages=None):
    """All pictures of one parsed stream on S streams side by side (private clones), inputs resident: pictures/s and the
    hash of stream S-1's last picture.  stages: a dict that receives, per slice type, the stage times of one more (untimed)
    pass over the pictures, measured with HIP events on the context's stream like the metric's."""
    from p264decoder_amd import HipReconstructor
    from tests.conftest import frame_sha256
    T = len(pics)
    hip = HipReconstructor(mb_w, mb_h, n_streams=S, slots=slots, max_pictures=S * T, lib=lib)
    hip.upload(0, pics)
    for s in range(1, S):
        for t in range(T):
            hip.clone_picture(s * T + t, t)
    hip.sync()
    streams = list(range(S))
    hip.reconstruct([s * T for s in streams], streams)           # the first picture once, untimed: allocations, first launches
    hip.sync()
    t0 = time.perf_counter()
    for t in range(T):
        hip.reconstruct([s * T + t for s in streams], streams)
    hip.sync()
    dt = time.perf_counter() - t0
    digest = frame_sha256(*hip.read_frame(S - 1, pics[-1].desc.dst_slot))
    if stages is not None:
        hip.timing_enable(True)
        acc = {}
        for t in range(T):
            hip.timing_reset()
            hip.reconstruct([s * T + t for s in streams], streams)
            hip.sync()
            a = acc.setdefault("IPB"[(2, 0, 1).index(pics[t].desc.slice_type)], {"launches": 0, "inter": 0.0, "intra": 0.0, "deblock": 0.0, "reconstruct": 0.0})
            a["launches"] += 1
            for k, (ms, cnt) in hip.timing_read().items():
                a[k] += ms
        hip.timing_enable(False)
        for a in acc.values():
            for k in ("inter", "intra", "deblock", "reconstruct"):
                a[k] = round(a[k] / a["launches"], 4)
        stages.update(acc)
    hip.close()
    return S * T / dt, digest


def bipred_bytes(pics):
    """SURVEY 8d's count for the MC stage of B pictures: per 8x8 quadrant and list used 96 bytes of reference samples read, per
    macroblock and list used 64 + 4 bytes of motion, 384 bytes written per inter macroblock (one list: the 836 bytes of a P
    macroblock).  Returns (bytes per B launch and stream, inter macroblocks, share of quadrants that use both lists)."""
    import numpy as np
    total = n_inter = 0
    bi = quads = 0
    nb = 0
    for p in pics:
        if p.desc.slice_type != 1:
            continue
        nb += 1
        rec = p.mb_records()
        inter = rec["mb_type"] > 2                                  # (_native: MB_IPCM = 2)
        r0 = p.ref_idx.reshape(-1, 4)[inter] >= 0
        r1 = p.ref_idx_l1.reshape(-1, 4)[inter] >= 0
        used0, used1 = r0 | ~r1, r1                                   # (no list at all reads list 0)
        total += int(used0.sum() + used1.sum()) * 96 + int(used0.any(axis=1).sum() + used1.any(axis=1).sum()) * 68 + int(inter.sum()) * 384
        n_inter += int(inter.sum())
        bi += int((used0 & used1).sum())
        quads += int(inter.sum()) * 4
    return (total // max(nb, 1), n_inter // max(nb, 1), bi / max(quads, 1))


def small_batch_leg(lib, S=256, K=20, Wm=3):
    """The metric's workload at SURVEY 8d's own batch: S = 256 independent 1080p pictures per launch (the headline runs 2048).
    Same stream, same step, same stage timing; stream 0 is the golden stream and its last picture is hashed against the real
    reference decoder's.  Reported under extras.batch_256 - never `value`."""
    from p264decoder_amd import HipReconstructor, Parser
    from tests import synth_cases
    from tests.conftest import frame_sha256
    T = 1 + Wm + K
    golden_hashes = synth_cases.golden("cfg3_1080p_allp")[1]
    parsed = []
    for g in range(DISTINCT):
        path = synth_cases.generate("cfg3_1080p_allp") if g == 0 else synth_cases.generate(synth_args(T, 1000 + g))
        parsed.append(Parser(quiet=True, lib=lib).parse_stream(open(path, "rb").read(), limit=T))
    hip = HipReconstructor(MB_W, MB_H, n_streams=S, slots=2, max_pictures=S * T, lib=lib)
    for s in range(min(S, DISTINCT)):
        hip.upload(s * T, parsed[s])
    for s in range(DISTINCT, S):
        for t in range(T):
            hip.clone_picture(s * T + t, (s % DISTINCT) * T + t)
    streams = list(range(S))
    for t in range(1 + Wm):
        hip.reconstruct([s * T + t for s in streams], streams)
    hip.sync()
    hip.timing_enable(True)
    hip.timing_reset()
    t0 = time.perf_counter()
    for t in range(1 + Wm, T):
        hip.reconstruct([s * T + t for s in streams], streams)
    hip.sync()
    elapsed = time.perf_counter() - t0
    timing = hip.timing_read()
    hip.timing_enable(False)
    launch = hip.last_launch()
    fused = launch["edge_info_fused"] > 0
    ok = frame_sha256(*hip.read_frame(0, parsed[0][-1].desc.dst_slot)) == golden_hashes[T - 1]
    hip.close()
    alg = algorithmic_bytes([parsed[s % DISTINCT][T - 1] for s in streams], fused)
    stages = {}
    for name in ("inter", "intra", "deblock"):
        ms, cnt = timing[name]
        if cnt:
            avg = ms / cnt
            stages[name] = {"kernels": stage_kernels(fused)[name], "avg_ms": round(avg, 4), "algorithmic_bytes": alg[name],
                            "frac_of_hbm_peak": round(alg[name] / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    return {"value": round(S * K / elapsed, 1), "unit": "frames/s", "streams": S, "pictures_per_launch": S, "steps": K, "ms_per_step": round(elapsed / K * 1e3, 4),
            "stages": stages, "launch": launch, "last_picture_matches_reference": ok,
            "what": "the metric's all-P workload at SURVEY 8d's batch of 256 pictures per launch (one picture per compute unit): stage times and their "
                    "fractions of the 8 TB/s roofline at the same algorithmic bytes per macroblock as the headline (MC: 836 B per inter macroblock)"}


def upload_inclusive_leg(lib, S=512, K=10, Wm=2, compact=True):
    """The metric's workload with the host->HBM copies of the parsed inputs inside the timed region (the seam hands over host
    buffers): every step uploads the S pictures it then reconstructs, one copy per picture out of pinned memory.  compact: the
    pictures travel in the compact link format (include/p264hip.h: one vector per partition, Intra4x4 modes where there are
    any, 8-bit levels where they fit) and are expanded on the device by ONE kernel per step (k_expand_compact, inside the timed
    region); else in the layout of an input slot (p264hip_upload_packed: what rounds 4 - 5 reported).  No parse, no packing
    (the blocks are made before the clock starts).  Reported under extras.upload_inclusive - never `value`."""
    import ctypes as C
    from p264decoder_amd import HipReconstructor, Parser
    from tests import synth_cases
    from tests.conftest import frame_sha256
    T = 1 + Wm + K
    golden_hashes = synth_cases.golden("cfg3_1080p_allp")[1]
    parsed = []
    for g in range(DISTINCT):
        path = synth_cases.generate("cfg3_1080p_allp") if g == 0 else synth_cases.generate(synth_args(T, 1000 + g))
        parsed.append(Parser(quiet=True, lib=lib).parse_stream(open(path, "rb").read(), limit=T))
    # the blocks in pinned memory (one per distinct picture; every stream's copy of it is its own transfer)
    pinned, total = [], 0
    for g in range(DISTINCT):
        row = []
        for t in range(T):
            pk = HipReconstructor.pack_compact(parsed[g][t], lib=lib) if compact else HipReconstructor.pack(parsed[g][t], lib=lib)
            ptr = lib.p264hip_host_alloc(pk.size)
            if not ptr:
                raise RuntimeError("p264hip_host_alloc failed")
            C.memmove(ptr, pk.ctypes.data, pk.size)
            row.append((ptr, pk.size))
        pinned.append(row)
    hip = HipReconstructor(MB_W, MB_H, n_streams=S, slots=2, max_pictures=S * 2, lib=lib)
    streams = list(range(S))
    up = lib.p264hip_upload_compact if compact else lib.p264hip_upload_packed

    def step(t):
        nonlocal total
        base = (t & 1) * S                                   # two sets of input slots: the copies of step t + 1 never touch step t's
        for s in streams:
            ptr, n = pinned[s % DISTINCT][t]
            hip._chk(up(hip.h, base + s, C.byref(parsed[s % DISTINCT][t].desc), ptr, n), "p264hip_upload_compact" if compact else "p264hip_upload_packed")
            total += n
        hip.reconstruct([base + s for s in streams], streams)
    for t in range(1 + Wm):
        step(t)
    hip.sync()
    total = 0
    t0 = time.perf_counter()
    for t in range(1 + Wm, T):
        step(t)
    hip.sync()
    elapsed = time.perf_counter() - t0
    ok = frame_sha256(*hip.read_frame(0, parsed[0][-1].desc.dst_slot)) == golden_hashes[T - 1]
    hip.close()
    for row in pinned:
        for ptr, _ in row:
            lib.p264hip_host_free(ptr)
    return {"value": round(S * K / elapsed, 1), "unit": "frames/s", "streams": S, "steps": K, "input_bytes_per_picture": int(total / (S * K)),
            "format": "compact link format, expanded on the device (k_expand_compact) inside the timed region" if compact else "slot layout (p264hip_upload_packed)",
            "upload_GBps": round(total / elapsed / 1e9, 2), "last_picture_matches_reference": ok,
            "what": "the metric's all-P workload with every picture's parsed input (one block in pinned host memory) copied host -> HBM inside the timed "
                    "region, %d pictures per launch; no parse" % S}


def extras(lib, baseline_stream=None):
    """Figures that are NOT the metric (never `value`): the other single-GPU configurations of BASELINE.json and the
    end-to-end rates, each on a bounded run."""
    from p264decoder_amd import Decoder, Parser
    from tests import synth_cases
    from tests.conftest import frame_sha256
    out = {}
    try:
        out["batch_256"] = small_batch_leg(lib)
    except Exception as e:                                    # never let an extra take the metric down
        out["batch_256"] = {"error": str(e)}
    try:
        out["upload_inclusive"] = upload_inclusive_leg(lib)
        out["upload_inclusive"]["slot_layout"] = {k: v for k, v in upload_inclusive_leg(lib, compact=False).items() if k in ("value", "input_bytes_per_picture", "upload_GBps", "format")}
    except Exception as e:
        out["upload_inclusive"] = {"error": str(e)}
    # config 2: 1280x720 Baseline CAVLC, I slices only (intra + IDCT path), 10 pictures x 1024 streams (as many streams as
    # the metric's run: with 256 the two row-wavefront kernels leave most of the chip idle - 97 k against 144 k frames/s)
    pics = Parser(quiet=True, lib=lib).parse_stream(synth_cases.stream_bytes("cfg2_720p_intra"))[:10]
    fps, digest = run_batched(lib, pics, 1024, 80, 45, 2)
    out["config2_720p_intra_only"] = {"value": round(fps, 1), "unit": "frames/s", "streams": 1024, "pictures_per_stream": len(pics),
                                       "last_picture_matches_reference": digest == synth_cases.golden("cfg2_720p_intra")[1][len(pics) - 1]}
    # config 3 as specified: I+P, GOP 30, 60 pictures x 1024 streams
    pics = Parser(quiet=True, lib=lib).parse_stream(synth_cases.stream_bytes("cfg3_1080p_ip"))
    fps, digest = run_batched(lib, pics, 1024, MB_W, MB_H, 2)
    out["config3_1080p_i_plus_p_gop30"] = {"value": round(fps, 1), "unit": "frames/s", "streams": 1024, "pictures_per_stream": len(pics),
                                            "last_picture_matches_reference": digest == synth_cases.golden("cfg3_1080p_ip")[1][-1]}
    # BASELINE config 4 ("next"): 1920x1088 Main profile, CABAC, I + P + B pictures (two B pictures between
    # reference pictures, implicit weights, direct prediction); 13 pictures x 1024 streams.
    # The reference cannot decode B pictures: the check is against the committed ORACLE hash (parity with the reference unpinned)
    try:
        name = "main_1080p_cabac_ipb"
        data = open(synth_cases.generate(synth_cases.ORACLE_CASES[name]), "rb").read()
        pics = Parser(quiet=True, lib=lib).parse_stream(data)
        # one parser thread on this stream, in C (the pipeline's parse-only mode, as single_thread_parse_fps below; until round 4 this
        # figure was timed around the Python wrapper's parse_stream and was mostly the wrapper: 96 against the real 250 - 300)
        from p264decoder_amd import Pipeline
        one = Pipeline([data * 4], threads=1, device=-1, lib=lib)
        st1 = one.run()
        one.close()
        parse_fps = st1["pictures"] / st1["seconds"]
        stages = {}
        fps, digest = run_batched(lib, pics, 1024, MB_W, MB_H, 3, stages)
        bytes_b, inter_b, bi_share = bipred_bytes(pics)
        if "B" in stages and stages["B"]["inter"] > 0:
            gbps = bytes_b * 1024 / (stages["B"]["inter"] * 1e-3) / 1e9
            stages["B"]["mc_roofline"] = {"bound": "hbm", "algorithmic_bytes_per_launch": bytes_b * 1024, "achieved": round(gbps, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                          "frac": round(gbps / HBM_PEAK_GBS, 4), "inter_mb_per_picture": inter_b, "quadrants_from_both_lists": round(bi_share, 3),
                                          "kernels": "k_mc_sort_b + k_mc + k_mc_second"}
        out["config4_1080p_main_cabac_ipb"] = {"value": round(fps, 1), "unit": "frames/s", "streams": 1024, "pictures_per_stream": len(pics),
                                                "stage_ms_per_launch": stages,
                                                "b_pictures_per_stream": sum(1 for p in pics if p.desc.slice_type == 1),
                                                "last_picture_matches_oracle": digest == synth_cases.oracle_golden(name)[1][-1],
                                                "cabac_parse_fps_one_thread": round(parse_fps, 1),
                                                "what": "BASELINE config 4: 1920x1080 Main profile, CABAC, I+P+B (two B pictures between reference pictures, implicit weights, "
                                                        "direct prediction), deblocking; reconstruction of resident inputs as in `value`.  Bi-predicted blocks take two passes "
                                                        "through the MC stage (k_mc, k_mc_second).  Pinned to the CPU oracle only - the reference decodes neither CABAC nor B pictures"}
    except Exception as e:
        out["config4_1080p_main_cabac_ipb"] = {"error": str(e)}
    # end to end: Annex-B bytes in host memory -> pictures in HBM, host CAVLC parse and PCIe uploads included.  The parse is
    # CPU work: what this figure can reach is set by the host cores this process may use (the cgroup CPU quota where there
    # is one - a one-GPU share of the box is 16 CPUs), so the quota, the parse-only ceiling (device = -1: the same threads
    # without the GPU) and the rate per parser thread are reported next to it.
    try:
        from p264decoder_amd import Pipeline
        quota = cpu_quota()
        distinct = [open(synth_cases.generate(synth_args(72, 1000 + g)), "rb").read() for g in range(4)]   # (72 pictures per stream: the context's creation behind the first round is 2 % of the run, not 7)
        n_streams = 128

        def run(threads, device):
            pipe = Pipeline([distinct[i % 4] for i in range(n_streams)], threads=threads, device=device, lib=lib)
            st = pipe.run()
            pipe.close()
            return st
        run(min(2 * quota, 64), 0)                             # untimed: allocations, first launches
        best, table = None, {}
        for threads in sorted({quota, min(quota + quota // 4, 128), min(2 * quota, 128)}):   # (a few more threads than CPUs even out the rounds' ends)
            pst, est = run(threads, -1), run(threads, 0)
            table[str(threads)] = {"parse_only_fps": round(pst["pictures"] / pst["seconds"], 1), "end_to_end_fps": round(est["pictures"] / est["seconds"], 1),
                                   "fps_per_parser_thread": round(est["pictures"] / est["parse_seconds"], 1)}
            if best is None or est["pictures"] / est["seconds"] > best[1]["pictures"] / best[1]["seconds"]:
                best = (threads, est)
        one = Pipeline([distinct[0]], threads=1, device=-1, lib=lib)
        st1 = one.run()
        one.close()
        threads, st = best
        fps = st["pictures"] / st["seconds"]
        out["end_to_end_pipeline"] = {"value": round(fps, 1), "unit": "frames/s", "streams": n_streams, "host_threads": threads, "cpu_quota": quota, "cpus_visible": os.cpu_count(),
                                      "by_threads": table, "single_thread_parse_fps": round(st1["pictures"] / st1["seconds"], 1),
                                      "upload_GBps": round(st["bytes_uploaded"] / st["seconds"] / 1e9, 2) if "bytes_uploaded" in st else None,
                                      "main_thread_waits_s": {"for_the_parsers": round(st["wait_parse_seconds"], 3), "for_the_device": round(st["wait_device_seconds"], 3), "of": round(st["seconds"], 3)},
                                      "what": "Annex-B in host memory -> CAVLC parse on the host threads (into registered huge pages) -> DMA uploads -> batched reconstruction; pictures stay in HBM; "
                                              "bound by the host parse: compare parse_only_fps (same threads, no GPU)"}
        # ... and the reference beside it: as many reference processes as the pipeline had parser threads, on the same cores
        try:
            if baseline_stream:
                out["end_to_end_pipeline"]["cpu_baseline_n"] = cpu_baseline_n(baseline_stream, threads)
        except Exception as e:
            out["end_to_end_pipeline"]["cpu_baseline_n"] = {"error": str(e)}
        # the same for config 4's kind of stream (Main profile, CABAC, I + P + B): the CABAC parse is the slower one
        try:
            main = open(synth_cases.generate(synth_cases.ORACLE_CASES["main_1080p_cabac_ipb"]), "rb").read()
            st4 = pst4 = None
            for th4 in sorted({quota, min(quota + quota // 4, 128), min(2 * quota, 128)}):   # (64 streams: its own best thread count)
                pipe = Pipeline([main * 2] * 64, threads=th4, device=0, lib=lib)
                e4 = pipe.run()
                pipe.close()
                if st4 is None or e4["pictures"] / e4["seconds"] > st4["pictures"] / st4["seconds"]:
                    pipe = Pipeline([main * 2] * 64, threads=th4, device=-1, lib=lib)
                    st4, pst4, threads = e4, pipe.run(), th4
                    pipe.close()
            out["end_to_end_pipeline_config4"] = {"value": round(st4["pictures"] / st4["seconds"], 1), "unit": "frames/s", "streams": 64, "host_threads": threads, "cpu_quota": quota,
                                                  "parse_only_fps": round(pst4["pictures"] / pst4["seconds"], 1), "fps_per_parser_thread": round(st4["pictures"] / st4["parse_seconds"], 1),
                                                  "what": "as end_to_end_pipeline, on 64 copies of the config-4 stream (1080p Main profile, CABAC, I + P + B) decoded twice"}
        except Exception as e:
            out["end_to_end_pipeline_config4"] = {"error": str(e)}
    except Exception as e:                                    # never let an extra take the metric down
        out["end_to_end_pipeline"] = {"error": str(e)}
    # the drop-in API, one stream, picture by picture with the I420 download (p264_decoder_decode)
    try:
        data = synth_cases.stream_bytes("cfg3_1080p_ip")
        dec = Decoder(lib=lib)
        t0 = time.perf_counter()
        n = sum(1 for _ in dec.decode_annexb(data))
        dt = time.perf_counter() - t0
        dec.close()
        out["single_stream_dropin_api"] = {"value": round(n / dt, 1), "unit": "frames/s", "pictures": n,
                                           "what": "p264_nal_decode + p264_decoder_decode per NAL, host parse, upload, reconstruction and I420 download per picture"}
    except Exception as e:
        out["single_stream_dropin_api"] = {"error": str(e)}
    return out


def fanout_leg(rank, local_rank, world, lib):
    """BASELINE config 5 (one rank owns inputs and outputs, RCCL send / recv over xGMI): every rank starts ONE child process
    on its GPU (python -m p264decoder_amd.tools.fan_bench) with a communicator of its own and waits for it with a time
    limit - whatever happens in there cannot disturb the timed result above.  Rank 0 returns the root child's report."""
    import torch.distributed as dist
    from p264decoder_amd import fanout
    # (rehearsal on a box with fewer GPUs than ranks - P264AMD_BENCH_DEVICE pins every rank to one GPU - : two ranks cannot share
    # a device in one RCCL communicator, so the same protocol runs over the TCP transport with its device entry points on)
    tcp = bool(os.environ.get("P264AMD_BENCH_DEVICE")) or os.environ.get("P264AMD_BENCH_FAN_TRANSPORT") == "tcp"
    uid = [None]
    if rank == 0:
        try:
            if tcp:
                import socket
                with socket.socket() as s:
                    s.bind(("127.0.0.1", 0))
                    uid[0] = "port:%d" % s.getsockname()[1]
            else:
                uid[0] = fanout.rccl_unique_id(lib).hex()
        except Exception as e:
            uid[0] = "error: %s" % e
    dist.broadcast_object_list(uid, src=0)
    if uid[0].startswith("error"):
        return {"error": uid[0]}
    cmd = [sys.executable, "-m", "p264decoder_amd.tools.fan_bench", "--rank", str(rank), "--world", str(world),
           "--device", str(local_rank), "--streams", str(max(world, 8)), "--pictures", "12"]
    cmd += ["--transport", "tcp", "--port", uid[0][5:]] if tcp else ["--transport", "rccl", "--uid", uid[0]]
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID", "GROUP_RANK", "ROLE_RANK"):
        env.pop(k, None)
    if tcp:
        env["P264AMD_FAN_TCP_DEVICE"] = "1"
    res = {"error": "no report"}
    try:
        p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT, env=env)
        try:
            out, err = p.communicate(timeout=120)
        except subprocess.TimeoutExpired:
            p.kill()
            out, err = p.communicate()
            res = {"error": "rank %d: fan-out child timed out" % rank}
        for line in out.splitlines():
            if line.startswith("FANOUT "):
                res = json.loads(line[7:])                 # the report, or {"error": the transport's / RCCL's own message}
        if res.get("error") == "no report" and p.returncode not in (0, None):
            res = {"error": "rank %d: child exited with %s: %s" % (rank, p.returncode, err.strip().splitlines()[-1] if err.strip() else "")}
        elif res.get("error") == "no report" and rank and p.returncode == 0:
            res = {"ok": True}                            # a worker that left in step has nothing to report
    except Exception as e:                                    # noqa: BLE001
        res = {"error": "rank %d: %r" % (rank, e)}
    # the root's report plus whatever the other ranks have to say (a worker's RCCL error is the interesting one when the
    # root only sees a time-out)
    every = [None] * world
    dist.all_gather_object(every, res)
    if rank == 0:
        worker_errors = [r["error"] for r in every[1:] if isinstance(r, dict) and r.get("error")]
        if worker_errors:
            res = dict(res, worker_errors=worker_errors)
    return res


def cpu_baseline(stream_path, n_pictures):
    """The reference's own CPU path on the host cores of this box, 1 core (it is single-threaded),
    on a bounded sample of the same workload.  kind = "reference" when oracle/_ref (the real
    reference, built in the build container) travelled with the repo, else "port" (our oracle)."""
    driver = os.path.join(ROOT, "oracle", "_ref", "p264ref_driver")
    if os.path.exists(driver):
        loops = 25                                          # ~12 s of single-core work at ~50 frames/s
        try:
            out = subprocess.run([driver, "time", stream_path, str(loops)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                                 text=True, timeout=600).stdout.split()
            return {"value": round(float(out[out.index("fps") + 1]), 3), "unit": "frames/s", "cores": 1, "kind": "reference", "cpu_model": cpu_model(),
                    "includes_parse": True,
                    "note": "the reference cannot reconstruct without parsing: this rate includes its CAVLC parse (12-14 % of its time, SURVEY 6), "
                            "`value` (inputs resident in HBM) does not - compare with extras.end_to_end_pipeline for parse-inclusive rates",
                    "sample": "%d-picture 1920x1088 all-P stream decoded %dx by the reference decoder (parse + reconstruction)" % (n_pictures, loops)}
        except Exception:
            pass
    from p264decoder_amd import Parser
    from tests import oracle_bind
    ora = oracle_bind.load()
    pics = Parser(quiet=True).parse_stream(open(stream_path, "rb").read())
    store = oracle_bind.FrameStore(MB_W, MB_H, 2)
    t0 = time.time()
    n = 0
    while time.time() - t0 < 12.0:
        for p in pics:
            oracle_bind.reconstruct(ora, store, p)
            n += 1
    return {"value": round(n / (time.time() - t0), 3), "unit": "frames/s", "cores": 1, "kind": "port", "cpu_model": cpu_model(), "includes_parse": False,
            "sample": "%d 1920x1088 pictures through the scalar oracle (reconstruction only, parse excluded)" % n}


def cpu_baseline_n(stream_path, n_procs, loops=12):
    """stream-parallel figures sit next to N independent reference processes on N cores (the reference is
    single-threaded, core/core.c:48, so N streams are N processes).  n_procs copies of oracle/_ref/p264ref_driver decode the
    same all-P 1080p stream `loops` times each, started together; value = all their pictures over the wall clock from the
    first start to the last exit.  None where the real reference did not travel (no port stands in here)."""
    driver = os.path.join(ROOT, "oracle", "_ref", "p264ref_driver")
    if not os.path.exists(driver):
        return None
    t0 = time.perf_counter()
    procs = [subprocess.Popen([driver, "time", stream_path, str(loops)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for _ in range(n_procs)]
    frames, per = 0, []
    for p in procs:
        out = p.communicate(timeout=900)[0].split()
        frames += int(out[out.index("frames") + 1])
        per.append(float(out[out.index("fps") + 1]))
    dt = time.perf_counter() - t0
    return {"value": round(frames / dt, 2), "unit": "frames/s", "cores": n_procs, "processes": n_procs, "kind": "reference", "cpu_model": cpu_model(), "cpu_quota": cpu_quota(),
            "fps_per_process_min_max": [round(min(per), 2), round(max(per), 2)], "includes_parse": True,
            "sample": "%d reference decoder processes side by side, each decoding the bench's 1920x1088 all-P stream %d times (parse + reconstruction), %d pictures in %.1f s"
                      % (n_procs, loops, frames, dt)}


def kernel_fingerprint():
    """SHA-256 over the HIP sources (csrc/hip/*, names and contents) and their compiler options: what a counter summary under profiles/ was collected on.
    profiles/summarize.py stamps traffic_latest.json with it; a summary of other kernels is STALE and is not replayed."""
    import hashlib
    d = os.path.join(ROOT, "p264decoder_amd", "csrc", "hip")
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith((".h", ".hip")):
            h.update(f.encode() + b"\0" + open(os.path.join(d, f), "rb").read() + b"\0")
    # (and how they are compiled: since round 6 a translation unit can have options of its own - build.py: HIP_EXTRA)
    from p264decoder_amd import build as _build
    h.update(repr((_build.HIP_ARCH, sorted(_build.HIP_EXTRA.items()))).encode())
    return h.hexdigest()


def live_counters(S, edge_info_fused, counters=("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU"), budget_s=200):
    """HBM traffic and vector instructions of THIS build on THIS box: one rocprofv3 --pmc pass per counter (never combined with
    a trace; MI355X_MICROARCH.md's HBM section) around a short child run of this same script - the same streams, the same
    batch, 1 warm-up + 2 timed P launches, no extras (so the child never gets here).  Per stage: bytes per launch =
    (2 x FETCH_SIZE + WRITE_SIZE) KB (gfx950: FETCH_SIZE counts a 128-byte request as 64), averaged over the P launches.
    Returns None when rocprofv3 is missing, refuses, or runs out of its time budget - the caller then falls back to the
    committed summary if that is of the same kernels."""
    import csv
    import glob
    import shutil
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None
    tmp = tempfile.mkdtemp(prefix="p264amd_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    t_end = time.time() + budget_s
    agg = {}
    try:
        for ctr in counters:
            left = t_end - time.time()
            if left < 20:
                return None
            d = os.path.join(tmp, ctr)
            cmd = [prof, "--pmc", ctr, "--kernel-include-regex", "^(void )?k_", "--output-format", "csv", "-d", d, "--",
                   sys.executable, os.path.abspath(__file__), "--steps", "2", "--warmup", "1", "--streams", str(S), "--no-extras", "--no-cpu-baseline"]
            try:
                r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, cwd="/tmp", env=env, timeout=left)
            except subprocess.TimeoutExpired:
                return None
            found = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not found:
                return None
            for row in csv.DictReader(open(max(found, key=os.path.getmtime))):
                k = row["Kernel_Name"].split("(")[0]
                k = k[5:] if k.startswith("void ") else k
                agg.setdefault(k, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)

    def avg(k, c):
        # (the P launches only: the first launch of k_deblock is the IDR picture's; k_mc* and k_intra_sparse have none for it)
        v = agg.get(k, {}).get(c, [])
        v = v[1:] if k.startswith("k_deblock") and len(v) > 1 else v
        return sum(v) / len(v) if v else 0.0
    stage_kernels_ = {"inter": ("k_mc_sort", "k_mc"), "intra": ("k_intra_sparse",), "deblock": ("k_deblock",) if edge_info_fused else ("k_deblock", "k_deblock_bs<false>")}
    out = {"launches_averaged": len(agg.get("k_mc", {}).get(counters[0], []))}
    for stage, ks in stage_kernels_.items():
        out[stage] = {"traffic": int(sum(2 * avg(k, "FETCH_SIZE") + avg(k, "WRITE_SIZE") for k in ks) * 1024) if "FETCH_SIZE" in counters else None,
                      "valu": int(sum(avg(k, "SQ_INSTS_VALU") for k in ks)) if "SQ_INSTS_VALU" in counters else None,
                      "per_kernel": {k: {c: round(avg(k, c)) for c in counters} for k in ks}}
    return out


def static_traffic(stage, S):
    """HBM traffic cannot be counted inside this run (PMC counters need rocprofv3 passes of their own): it is replayed
    from the committed summary of the same command profiled on the same code (profiles/collect.sh), and says so."""
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if not os.path.exists(tpath):
        return None, None
    try:
        tj = json.load(open(tpath))
        if tj.get("kernels_sha256") != kernel_fingerprint():
            # (the driver's line must not carry a traffic figure of kernels that are no longer the ones that ran)
            sys.stderr.write("bench.py: profiles/traffic_latest.json was collected on other kernels (csrc/hip changed since %s): NOT replayed - "
                             "run profiles/collect.sh + summarize.py\n" % tj.get("source"))
            return None, "STALE: profiles/%s was collected on other kernels than this build's (csrc/hip fingerprint differs); not replayed" % tj.get("source")
        if S == int(tj.get("pictures_per_launch", 1024)):              # (counted for the default batch: scaled to nothing else)
            return tj.get(stage), "static: profiles/%s, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --steps 3`, %s" % (tj.get("source"), tj.get("formula"))
        return None, "none: profiles/%s was collected at %d pictures per launch, this run has %d" % (tj.get("source"), int(tj.get("pictures_per_launch", 1024)), S)
    except Exception:
        return None, None


def measured_copy_bandwidth(torch):
    """Achievable HBM ceiling on this box next to the 8 TB/s vendor peak: device-to-device copy of 1 GiB,
    bytes read + bytes written per second."""
    n = 1 << 30
    a = torch.empty(n, dtype=torch.uint8, device="cuda")
    b = torch.empty_like(a)
    a.fill_(1)
    b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    gbps = 2.0 * n * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del a, b
    torch.cuda.empty_cache()
    return round(gbps, 1)


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves.  This process has not touched the GPU (no
    HIP call, not even torch imported) and never will: the ranks are fresh children of `python -m torch.distributed.run`
    (a subprocess, not an exec), rank 0's JSON line is relayed on stdout, everything else on stderr."""
    import socket
    rc, line = 1, None
    for attempt in range(2):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + argv
        env = dict(os.environ)
        env.setdefault("OMP_NUM_THREADS", "1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT, env=env)
        out, err = p.communicate()
        rc = p.returncode
        sys.stderr.write(err)
        for l in out.splitlines():
            if l.startswith("{") and '"metric"' in l:
                line = l
            else:
                sys.stderr.write(l + "\n")
        # (the port was free when it was picked and may be taken when the rendezvous binds it: once more on exactly that)
        if rc == 0 or not any(w in err for w in ("ddress already in use", "EADDRINUSE", "failed to bind")):
            break
    if rc != 0 or line is None:
        raise SystemExit("bench.py --gpus %d: the ranks failed (exit code %s, %s)" % (n, rc, "no JSON line" if line is None else "JSON line present"))
    if json.loads(line).get("n_gpus") != n:
        raise SystemExit("bench.py --gpus %d: the job reports n_gpus = %s" % (n, json.loads(line).get("n_gpus")))
    print(line, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    # 2048 streams = 8 pictures per CU and launch (round 4: 1024 -> 2048 streams +5 % frames/s, 3072 / 4096 no more; the wavefront
    # kernels balance their bands better with more pictures per workgroup).  26 pictures x 2048 streams of parsed input = 80 GB.
    ap.add_argument("--streams", type=int, default=2048, help="independent 1080p streams per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the non-metric figures (config 2, config 3 I+P, pipeline, drop-in API)")
    ap.add_argument("--no-live-counters", action="store_true", help="do not run the rocprofv3 --pmc child passes (HBM traffic, vector instructions) behind the timed region")
    ap.add_argument("--no-fanout", action="store_true", help="N > 1: skip the config-5 fan-out leg (also P264AMD_BENCH_FANOUT=0).  By default the leg runs at "
                    "N = 2 only - two more GPU processes beside the two ranks - so that a transport that has never met a second device cannot take "
                    "the N = 4 / 8 scaling lines down with it; P264AMD_BENCH_FANOUT=1 runs it at any N > 1")
    ap.add_argument("--only-upload-inclusive", action="store_true", help="print extras.upload_inclusive (both formats) and nothing else")
    ap.add_argument("--only-batch-256", action="store_true", help="print extras.batch_256 (the metric's workload at 256 pictures per launch) and nothing else")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:    # no launcher around us: be the launcher (before anything touches the GPU)
        return launch_ranks(args.gpus, sys.argv[1:])
    from p264decoder_amd import shard
    rank, local_rank, world = shard.env_rank()
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE = %d: start one rank per GPU (or run without a launcher and let "
                         "bench.py start them)\n" % (args.gpus, world))
        raise SystemExit(2)
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the reconstruction path")
    # (rehearsal knobs for a box with fewer GPUs than ranks: P264AMD_BENCH_DEVICE pins every rank to one GPU,
    # P264AMD_BENCH_BACKEND=gloo replaces RCCL for the barrier and the clock; the driver's runs use neither)
    if os.environ.get("P264AMD_BENCH_DEVICE"):
        local_rank = int(os.environ["P264AMD_BENCH_DEVICE"])
    backend = os.environ.get("P264AMD_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    shard.init(backend, torch.device("cuda", local_rank))      # "nccl" is RCCL on ROCm; barrier + clock only

    from p264decoder_amd import HipReconstructor, Parser, _native
    from tests import synth_cases
    lib = _native.load()

    if args.only_upload_inclusive:
        if rank == 0:
            print(json.dumps({"extras": {"upload_inclusive": {"compact": upload_inclusive_leg(lib), "slot_layout": upload_inclusive_leg(lib, compact=False)}}}), flush=True)
        return
    if args.only_batch_256:                               # (profiles/collect.sh: the kernel trace and the counters of this leg alone)
        if rank == 0:
            print(json.dumps({"extras": {"batch_256": small_batch_leg(lib, K=args.steps, Wm=args.warmup)}}), flush=True)
        return
    S, K, Wm = args.streams, args.steps, args.warmup
    T = 1 + Wm + K                                        # pictures per stream: IDR + warm-up + timed
    # ---- set-up (untimed): write + parse DISTINCT streams, make every stream's inputs resident ----
    # Stream 0 of every rank is the golden all-P stream (tests/golden/synth_cfg3_1080p_allp.sha256: per-picture hashes of
    # the REAL reference decoder) as long as it is long enough: its timed output is checked against those hashes below.
    golden_case = "cfg3_1080p_allp" if T <= 30 else "cfg3_1080p_allp_300"      # (the same stream, 30 / 300 pictures of it)
    golden_hashes = synth_cases.golden(golden_case)[1]
    use_golden = T <= len(golden_hashes)
    synth_extra = os.environ.get("P264AMD_BENCH_SYNTH_EXTRA", "")     # experiments only (e.g. "--mvmax 0"): no golden stream then
    if synth_extra or os.environ.get("P264AMD_BENCH_NO_GOLDEN"):     # timing experiments with deliberately wrong kernels (scratch/)
        use_golden = False
    paths, parsed = [], []
    for g in range(DISTINCT):
        path = synth_cases.generate(golden_case) if (g == 0 and use_golden) else synth_cases.generate(synth_args(T, 1000 + 16 * rank + g) + (" " + synth_extra if synth_extra else ""))
        paths.append(path)
        pics = Parser(quiet=True, lib=lib).parse_stream(open(path, "rb").read(), limit=T)
        assert len(pics) == T and all(p.desc.slice_type == 0 for p in pics[1:])
        parsed.append(pics)
    hip = HipReconstructor(MB_W, MB_H, n_streams=S, slots=2, max_pictures=S * T, device=local_rank, lib=lib)
    for s in range(min(S, DISTINCT)):
        hip.upload(s * T, parsed[s])
    for s in range(DISTINCT, S):
        for t in range(T):
            hip.clone_picture(s * T + t, (s % DISTINCT) * T + t)
    hip.sync()
    streams = list(range(S))

    def step(t):
        hip.reconstruct([s * T + t for s in streams], streams)

    for t in range(1 + Wm):                               # IDR + W P pictures
        step(t)
    hip.sync()
    hip.timing_enable(True)
    hip.timing_reset()
    torch.cuda.synchronize()
    shard.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(1 + Wm, T):
        step(t)
    hip.sync()
    torch.cuda.synchronize()
    shard.barrier()
    torch.cuda.synchronize()
    elapsed = shard.max_over_ranks(time.perf_counter() - t0, device="cuda" if backend == "nccl" else "cpu")
    timing = hip.timing_read()
    hip.timing_enable(False)
    launch = hip.last_launch()                             # the shapes the timed launches had; which launch ran the edge-info pass
    EDGE_INFO_FUSED = launch["edge_info_fused"] > 0
    STAGE_KERNELS = stage_kernels(EDGE_INFO_FUSED)

    # ---- the timed output against the real reference decoder: the last picture of stream 0 and of its last clone must
    #      hash to what the reference produced for that picture of the golden stream (committed fixture) ----
    from tests.conftest import frame_sha256
    last = parsed[0][-1].desc.dst_slot
    golden_check = {"stream": golden_case, "picture": T - 1, "checked": False}
    if use_golden:
        clone = S - 1 - (S - 1) % DISTINCT                  # the last stream that decodes stream 0's pictures
        for s in sorted({0, clone}):
            got = frame_sha256(*hip.read_frame(s, last))
            if got != golden_hashes[T - 1]:
                raise SystemExit("bench.py: stream %d picture %d differs from the reference decoder (%s != %s)" % (s, T - 1, got[:16], golden_hashes[T - 1][:16]))
        golden_check.update(checked=True, streams_checked=sorted({0, clone}), sha256=golden_hashes[T - 1][:16] + "...",
                            source="tests/golden/synth_%s.sha256 (oracle/_ref, the real reference decoder)" % golden_case)

    # what every rank did (N > 1: the streams each rank owns are its own - S per rank, weak scaling - and every rank checks its
    # own golden stream against the reference's hash)
    ranks = shard.gather_objects({"rank": rank, "device": local_rank, "streams": S, "frames": S * K, "golden_checked": golden_check["checked"],
                                  "first_global_stream": rank * S})
    copy_gbps = measured_copy_bandwidth(torch) if rank == 0 else None
    if rank == 0:
        frames = sum(r["frames"] for r in ranks)
        assert frames == S * K * world and len(ranks) == world
        fps = frames / elapsed
        alg = algorithmic_bytes([parsed[s % DISTINCT][t] for s in streams for t in (T - 1,)], EDGE_INFO_FUSED)   # one representative step
        kernels = {}
        for name in ("inter", "intra", "deblock"):
            ms, cnt = timing[name]
            if cnt:
                avg = ms / cnt
                gbps = alg[name] / (avg * 1e-3) / 1e9
                kernels[name] = {"kernels": STAGE_KERNELS[name], "avg_ms": round(avg, 4), "launches": int(cnt), "algorithmic_bytes": alg[name],
                                 "GBps": round(gbps, 1), "frac_of_hbm_peak": round(gbps / HBM_PEAK_GBS, 4)}
                if name == "inter":                       # the read-only fraction (452 B/MB) and, separately, the figure with the residual input
                    kernels[name]["bytes_per_inter_mb"] = 836
                    kernels[name]["frac_read"] = round(alg["inter_read"] / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                    kernels[name]["algorithmic_bytes_with_residual"] = alg["inter_with_residual"]
                    kernels[name]["frac_with_residual"] = round(alg["inter_with_residual"] / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        dom = max(kernels, key=lambda k: kernels[k]["avg_ms"])

        traffic_of = lambda stage: static_traffic(stage, S)   # noqa: E731

        def valu_of(stage):
            """Vector-instruction issue of the stage's kernels: instructions per launch from the committed counter summary (like the
            traffic: counters cannot be read inside this run), against this run's stage time - wave instructions x 4 cycles on every
            SIMD of the device.  The kernels of this path are bound by instruction issue as much as by HBM."""
            try:
                tj = json.load(open(os.path.join(ROOT, "profiles", "traffic_latest.json")))
                if S != int(tj.get("pictures_per_launch", 1024)) or tj.get("kernels_sha256") != kernel_fingerprint():
                    return None
                pmc = json.load(open(os.path.join(ROOT, "profiles", tj["source"])))
                names = {"inter": ("k_mc_sort", "k_mc"), "intra": ("k_intra_sparse",), "deblock": ("k_deblock",) if EDGE_INFO_FUSED else ("k_deblock", "k_deblock_bs<false>")}[stage]
                insts = sum(int(pmc[k]["SQ_INSTS_VALU"]) for k in names)
                prop = torch.cuda.get_device_properties(local_rank)
                # (the clock under these kernels' load: 2.24 GHz by GRBM_GUI_ACTIVE / 8 XCDs / kernel time, scratch/r5_pmc2.sh - not the
                # 2400 MHz of hipDeviceProp_t.clockRate, which rounds 1 - 4 had assumed)
                simds, hz = prop.multi_processor_count * 4, 2.24e9
                return {"wave_instructions_per_launch": insts, "issue_cycles_frac": round(insts * 4.0 / (kernels[stage]["avg_ms"] * 1e-3 * hz * simds), 3),
                        "simds": simds, "clock_MHz": round(hz / 1e6), "clock_source": "measured under load in round 5 (GRBM_GUI_ACTIVE), static here",
                        "source": "static: profiles/%s (SQ_INSTS_VALU), 4 cycles per wave instruction" % tj["source"]}
            except Exception:
                return None

        def roofline_of(stage):
            traffic, traffic_source = traffic_of(stage)
            r = {"kernel": STAGE_KERNELS[stage], "stage": stage, "bound": "hbm", "achieved": kernels[stage]["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": round(kernels[stage]["GBps"] / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_unit": "bytes per launch", "traffic_source": traffic_source,
                 "algorithmic_bytes_per_launch": kernels[stage]["algorithmic_bytes"], "avg_ms": kernels[stage]["avg_ms"], "measured_copy_GBps": copy_gbps}
            if traffic:
                r["traffic_over_algorithmic"] = round(traffic / kernels[stage]["algorithmic_bytes"], 3)
            r["valu"] = valu_of(stage)
            if stage == "inter":
                r["bytes_per_inter_mb"] = 836
                r["frac_read"] = kernels[stage]["frac_read"]
                r["frac_with_residual"] = kernels[stage]["frac_with_residual"]
                if traffic:
                    r["traffic_over_algorithmic_with_residual"] = round(traffic / kernels[stage]["algorithmic_bytes_with_residual"], 3)
            return r
        out = {
            "metric": "1080p decoded frames/sec", "value": round(fps, 2), "unit": "frames/s",
            "n_gpus": world, "steps": K, "warmup": Wm, "ms_per_step": round(elapsed / K * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int16", "data": "synthetic",
            "config": {"workload": "BASELINE config 3: 1920x1080 (coded 1920x1088) Baseline CAVLC all-P stream (1 IDR + P), "
                                   "reconstruction hot path (MC + residual, intra, deblock), parsed inputs resident in HBM",
                       "streams_per_gpu": S, "pictures_per_step": S * world, "mb_per_picture": N_MB, "parallelism": "stream-parallel x%d" % world},
            "macroblocks_per_s": round(fps * N_MB, 0),
            # `roofline` = the stage that takes longest (the contract's "dominant kernel"); `roofline_mc` = the motion-compensation
            # stage, which the north star names, whichever is longer
            "roofline": roofline_of(dom),
            "roofline_mc": roofline_of("inter") if "inter" in kernels else None,
            "kernels": kernels,
            "reconstruct_call_ms": round(timing["reconstruct"][0] / max(timing["reconstruct"][1], 1), 4),
            "golden_check": golden_check,
            "ranks": ranks,
            "launch": launch,
            "build": {"timing_build": bool(lib.p264hip_build_info() & 1)},
        }
        if not args.no_cpu_baseline and world == 1:          # rank 0 at N=1 only: a reported baseline, not part of the scaling runs
            out["cpu_baseline"] = cpu_baseline(paths[0], T)
    hip.close()
    if rank == 0 and world == 1 and not args.no_extras and not args.no_live_counters:
        # counters of this build on this box, in child processes after the timed region (never inside it)
        try:
            live = live_counters(S, EDGE_INFO_FUSED)
        except Exception as e:                              # noqa: BLE001 - the committed summary stays
            live = None
            sys.stderr.write("bench.py: live counters failed: %r\n" % (e,))
        if live:
            prop = torch.cuda.get_device_properties(local_rank)
            for r in (out["roofline"], out["roofline_mc"]):
                if not r or not live[r["stage"]]["traffic"]:
                    continue
                lv, st = live[r["stage"]], r["stage"]
                r.update(traffic=lv["traffic"], traffic_over_algorithmic=round(lv["traffic"] / kernels[st]["algorithmic_bytes"], 3),
                         traffic_source="live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate, no trace) of a child `bench.py --steps 2 --warmup 1 --streams %d` on this box "
                                        "after the timed region, (2*FETCH_SIZE + WRITE_SIZE) KB averaged over %d P launches" % (S, live["launches_averaged"]),
                         counters_per_kernel=lv["per_kernel"])
                if st == "inter":
                    r["traffic_over_algorithmic_with_residual"] = round(lv["traffic"] / kernels[st]["algorithmic_bytes_with_residual"], 3)
                if lv["valu"]:
                    simds, hz = prop.multi_processor_count * 4, 2.24e9
                    r["valu"] = {"wave_instructions_per_launch": lv["valu"], "issue_cycles_frac": round(lv["valu"] * 4.0 / (kernels[st]["avg_ms"] * 1e-3 * hz * simds), 3),
                                 "simds": simds, "clock_MHz": round(hz / 1e6), "clock_source": "measured under load in round 5 (GRBM_GUI_ACTIVE), static here",
                                 "source": "live: SQ_INSTS_VALU pass of the same child run, 4 cycles per wave instruction"}
    fan = None
    fan_env = os.environ.get("P264AMD_BENCH_FANOUT", "")
    if world > 1 and not args.no_fanout and fan_env != "0" and (world == 2 or fan_env == "1"):
        fan = fanout_leg(rank, local_rank, world, lib)       # after the timed region, in child processes, never `value`
    if rank == 0:
        if not args.no_extras and world == 1:
            out["extras"] = extras(lib, None if args.no_cpu_baseline else paths[0])
        if fan is not None:
            out.setdefault("extras", {})["fanout_config5"] = fan
        print(json.dumps(out), flush=True)
    shard.barrier()
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
