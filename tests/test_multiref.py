"""Beyond the reference's safe subset: two reference frames with a reference index per partition
and several slices per picture.  The reference decoder is broken
here (Appendix A-Q5: out-of-bounds scan8 index), so nothing can be pinned against it; instead
  * CPU: the parser's motion vectors and reference indices are checked against the stream writer's own record of what it
    coded (tools/synth264 --dump-mv): two independent implementations of the H.264 8.4.1.3 predictor and the list-0
    construction (sliding window, PicNum order) have to agree;
  * GPU: the HIP path against the CPU oracle, picture by picture - both take reference slots per 8x8 quadrant, which
    exercises the per-item reference offsets of the motion-compensation kernels and the reference test of the boundary strengths."""
import os
import subprocess

import numpy as np
import pytest

from p264decoder_amd import Parser, _native as N
from tests import synth_cases

ARGS = "--mbw 11 --mbh 9 --frames 10 --gop 0 --seed 41 --refs 2 --coded 8 --maxlevel 6"
# several slices per picture (the reference handles one, decoder/decoder.c:516-523): slice boundaries in the middle of
# macroblock rows change every neighbour-availability pattern of the intra predictors and the vector / nC / mode predictors
SLICED = ["--mbw 11 --mbh 9 --frames 8 --gop 4 --seed 43 --slices 4 --coded 10 --maxlevel 6",
          "--mbw 7 --mbh 6 --frames 9 --gop 0 --seed 44 --slices 5 --refs 2 --coded 12 --maxlevel 6",
          "--mbw 9 --mbh 7 --frames 8 --gop 4 --seed 45 --slices 3 --deblock-idc 2 --coded 14 --maxlevel 8"]   # no filtering across slices


# sub-8x8 partitions (8x4, 4x8, 4x4; the reference mis-decodes them, A-Q4) and list-0 reordering (ignored by the reference,
# decoder/lists.c:146-149): spec-driven, pinned by the writer's record and by HIP == oracle
SUB8X8 = ["--mbw 11 --mbh 9 --frames 8 --gop 4 --seed 46 --sub8x8 --coded 10 --maxlevel 6",
          "--mbw 9 --mbh 8 --frames 9 --gop 0 --seed 47 --sub8x8 --refs 2 --slices 2 --mvmax 40 --coded 12 --maxlevel 6"]
REORDER = "--mbw 10 --mbh 8 --frames 12 --gop 0 --seed 48 --refs 2 --reorder --sub8x8 --coded 10 --maxlevel 6"


MMCO = ["--mbw 8 --mbh 6 --frames 40 --gop 14 --seed 71 --refs 3 --mmco --coded 8 --maxlevel 6",
        "--mbw 7 --mbh 5 --frames 36 --gop 0 --seed 72 --refs 4 --mmco --sub8x8 --coded 8 --maxlevel 6",
        # operation 5 too: everything but the current picture goes, which then counts as frame_num 0 (7.4.3, 8.2.1) - the
        # pictures behind it (at least num_ref_frames P pictures before the next one) build their lists against that
        "--mbw 7 --mbh 5 --frames 60 --gop 0 --seed 73 --refs 3 --mmco5 --coded 8 --maxlevel 6"]



def make(tmp_path, args=ARGS):
    synth_cases.ensure_tool()
    stream, dump = str(tmp_path / "mr.264"), str(tmp_path / "mr.mv")
    subprocess.run([synth_cases.TOOL, stream] + args.split() + ["--dump-mv", dump], check=True)
    return open(stream, "rb").read(), np.fromfile(dump, dtype=np.uint8)


def test_parser_against_writer(lib, tmp_path):
    data, dump = make(tmp_path)
    parser = Parser(quiet=True, lib=lib)
    pics = parser.parse_stream(data)
    n = pics[0].n_mb
    per = n * 64 + n * 16
    assert len(pics) == 10 and parser.slots == 3 and len(dump) == per * len(pics)
    both = 0
    for i, p in enumerate(pics):
        blob = dump[i * per:(i + 1) * per]
        mv = blob[:n * 64].view(np.int16).reshape(n, 16, 2)
        rf = blob[n * 64:].view(np.int8).reshape(n, 16)
        assert p.desc.n_ref == min(i, 2)
        if i >= 2:                                            # list 0 = the two previous pictures, most recent first
            assert p.desc.ref_slot[0] == pics[i - 1].desc.dst_slot and p.desc.ref_slot[1] == pics[i - 2].desc.dst_slot
        inter = p.mb_records()["mb_type"] > N.MB_IPCM
        assert np.array_equal(p.mv.reshape(n, 16, 2)[inter], mv[inter]), "picture %d: vectors" % i
        assert np.array_equal(p.ref_idx.reshape(n, 4)[inter], rf[:, [0, 2, 8, 10]][inter]), "picture %d: reference indices" % i
        both += int(((rf == 1).any(axis=1) & (rf == 0).any(axis=1)).sum())
    assert both > 20                                          # macroblocks mixing both references exist


@pytest.mark.parametrize("args", SLICED)
def test_sliced_pictures_parser_against_writer(lib, tmp_path, args):
    data, dump = make(tmp_path, args)
    pics = Parser(quiet=True, lib=lib).parse_stream(data)
    n = pics[0].n_mb
    per = n * 64 + n * 16
    assert len(dump) == per * len(pics)
    patterns = set()
    for i, p in enumerate(pics):
        blob = dump[i * per:(i + 1) * per]
        mv = blob[:n * 64].view(np.int16).reshape(n, 16, 2)
        rf = blob[n * 64:].view(np.int8).reshape(n, 16)
        rec = p.mb_records()
        inter = rec["mb_type"] > N.MB_IPCM
        assert np.array_equal(p.mv.reshape(n, 16, 2)[inter], mv[inter]), "picture %d: vectors" % i
        assert np.array_equal(p.ref_idx.reshape(n, 4)[inter], rf[:, [0, 2, 8, 10]][inter]), "picture %d: reference indices" % i
        patterns |= set(rec["avail"].tolist())
    assert len(patterns) >= 7                                 # far more neighbour patterns than a single slice produces


@pytest.mark.parametrize("args", SUB8X8)
def test_sub8x8_partitions_parser_against_writer(lib, tmp_path, args):
    data, dump = make(tmp_path, args)
    pics = Parser(quiet=True, lib=lib).parse_stream(data)
    n = pics[0].n_mb
    per = n * 64 + n * 16
    assert len(dump) == per * len(pics)
    shapes = set()
    for i, p in enumerate(pics):
        blob = dump[i * per:(i + 1) * per]
        mv = blob[:n * 64].view(np.int16).reshape(n, 16, 2)
        rf = blob[n * 64:].view(np.int8).reshape(n, 16)
        rec = p.mb_records()
        inter = rec["mb_type"] > N.MB_IPCM
        got = p.mv.reshape(n, 16, 2)
        assert np.array_equal(got[inter], mv[inter]), "picture %d: vectors" % i
        assert np.array_equal(p.ref_idx.reshape(n, 4)[inter], rf[:, [0, 2, 8, 10]][inter]), "picture %d: reference indices" % i
        # which sub-partition shapes occur: per 8x8 quadrant, are the two rows / the two columns of vectors different?
        for m in np.nonzero(rec["mb_type"] == N.MB_P_8x8)[0]:
            v = got[m].reshape(4, 4, 2)
            for qy in (0, 2):
                for qx in (0, 2):
                    q = v[qy:qy + 2, qx:qx + 2]
                    rows = not np.array_equal(q[0], q[1])
                    cols = not np.array_equal(q[:, 0], q[:, 1])
                    shapes.add((rows, cols))
    assert shapes == {(False, False), (True, False), (False, True), (True, True)}      # 8x8, 8x4, 4x8, 4x4 all present


def test_list0_reordering_parser_against_writer(lib, tmp_path):
    data, dump = make(tmp_path, REORDER)
    parser = Parser(quiet=True, lib=lib)
    pics = parser.parse_stream(data)
    n = pics[0].n_mb
    per = n * 64 + n * 16 + 1
    assert len(pics) == 12 and len(dump) == per * len(pics)
    swapped = 0
    for i, p in enumerate(pics):
        blob = dump[i * per:(i + 1) * per]
        mv = blob[:n * 64].view(np.int16).reshape(n, 16, 2)
        rf = blob[n * 64:n * 80].view(np.int8).reshape(n, 16)
        reordered = int(blob[-1])
        inter = p.mb_records()["mb_type"] > N.MB_IPCM
        assert np.array_equal(p.mv.reshape(n, 16, 2)[inter], mv[inter]), "picture %d: vectors" % i
        assert np.array_equal(p.ref_idx.reshape(n, 4)[inter], rf[:, [0, 2, 8, 10]][inter]), "picture %d: reference indices" % i
        if i >= 2:
            want = [pics[i - 1].desc.dst_slot, pics[i - 2].desc.dst_slot]
            if reordered:
                want.reverse()
                swapped += 1
            assert [p.desc.ref_slot[0], p.desc.ref_slot[1]] == want, "picture %d: list 0" % i
    assert 2 <= swapped <= 8


@pytest.mark.gpu
@pytest.mark.parametrize("args", [ARGS] + SLICED + SUB8X8 + [REORDER] + MMCO)
def test_two_references_hip_vs_oracle(lib, oracle, tmp_path, args):
    from p264decoder_amd import HipReconstructor
    from tests import oracle_bind
    data, _ = make(tmp_path, args)
    parser = Parser(quiet=True, lib=lib)
    pics = parser.parse_stream(data)
    mb_w, mb_h = pics[0].mb_w, pics[0].mb_h
    store = oracle_bind.FrameStore(mb_w, mb_h, parser.slots)
    hip = HipReconstructor(mb_w, mb_h, n_streams=1, slots=parser.slots, max_pictures=1, lib=lib)
    for i, p in enumerate(pics):
        want = oracle_bind.reconstruct(oracle, store, p)
        hip.submit(0, p)
        got = hip.read_frame(0, p.desc.dst_slot)
        for plane, (a, b) in enumerate(zip(got, want)):
            assert np.array_equal(a, b), "picture %d plane %d differs" % (i, plane)
    hip.close()


@pytest.mark.parametrize("args", MMCO)
def test_adaptive_marking_and_long_term_parser_against_writer(lib, tmp_path, args):
    """memory_management_control_operation 1, 2, 3, 6 and long-term IDR pictures (the reference parses
    the commands and ignores them, decoder/lists.c:183-187): the writer keeps its own model of the frame store and records
    list 0 of every picture as picture numbers; the parser's list 0 must name the frame-store slots those pictures were
    decoded into - and those slots must still hold them."""
    data, dump = make(tmp_path, args)
    parser = Parser(quiet=True, lib=lib)
    pics = parser.parse_stream(data)
    n = pics[0].n_mb
    base = n * 64 + n * 16
    slot_pic, at, multi, longs = {}, 0, 0, 0
    assert parser.slots == int(args.split("--refs ")[1].split()[0]) + 1
    for i, p in enumerate(pics):
        blob = dump[at:at + base]
        mv = blob[:n * 64].view(np.int16).reshape(n, 16, 2)
        rf = blob[n * 64:].view(np.int8).reshape(n, 16)
        cnt = int(dump[at + base])
        want = [int(dump[at + base + 1 + 2 * k]) | int(dump[at + base + 2 + 2 * k]) << 8 for k in range(cnt)]
        at += base + 1 + 2 * cnt
        inter = p.mb_records()["mb_type"] > N.MB_IPCM
        assert np.array_equal(p.mv.reshape(n, 16, 2)[inter], mv[inter]), "picture %d: vectors" % i
        assert np.array_equal(p.ref_idx.reshape(n, 4)[inter], rf[:, [0, 2, 8, 10]][inter]), "picture %d: reference indices" % i
        assert p.desc.n_ref == cnt, "picture %d: list length" % i
        got = [slot_pic[p.desc.ref_slot[k]] for k in range(cnt)]
        assert got == want, "picture %d: list 0 is pictures %s, the writer meant %s" % (i, got, want)
        assert p.desc.dst_slot not in [p.desc.ref_slot[k] for k in range(cnt)]
        multi += cnt > 2
        longs += any(w < i - 4 for w in want)             # an entry older than any sliding window of this size would keep
        slot_pic[p.desc.dst_slot] = i
    assert at == len(dump) and multi > 5 and longs > (3 if "--mmco5" not in args else 0)
