"""CPU half of the seam fuzz: the random-picture builder (tests/seam_fuzz.py) produces well-formed p264hip_picture_t input -
the oracle consumes it, the result is deterministic for a seed, and the packed coefficient stream is laid out as
include/p264hip.h says.  (The comparison with the HIP kernels is tests/test_gpu_seam_fuzz.py.)"""
import numpy as np

from p264decoder_amd import _native as N
from tests import oracle_bind, seam_fuzz


def run(oracle, seed):
    rng = np.random.default_rng(seed)
    store = oracle_bind.FrameStore(7, 5, 3)
    for s in range(3):
        for dst, src in zip(store[s], seam_fuzz.random_frame(rng, 7, 5)):
            dst[:] = src
    out = []
    for i in range(4):
        pic = seam_fuzz.make_picture(rng, 7, 5, p_picture=(i != 1), n_ref=2, slots=3, dst_slot=i % 3, level_style="mixed", slices=2)
        rec = pic.rec
        # coefficient stream: indices are consecutive, block counts match the masks
        want_index = 0
        for m in range(pic.n_mb):
            assert rec["coef_index"][m] == want_index
            mask = int(rec["coef_mask"][m])
            want_index += bin(mask & 0x3ffffff).count("1")
        assert want_index == pic.desc.n_coef_blocks and len(pic.coefs) >= 16 * want_index
        inter = rec["mb_type"] > N.MB_IPCM
        assert (pic.ref_idx.reshape(-1, 4)[~inter] == -1).all()
        got = oracle_bind.reconstruct(oracle, store, pic)
        out.append([p.copy() for p in got])
    return out


def test_builder_is_deterministic_and_wellformed(oracle):
    a, b = run(oracle, 5), run(oracle, 5)
    for x, y in zip(a, b):
        for p, q in zip(x, y):
            assert np.array_equal(p, q)
    c = run(oracle, 6)
    assert any(not np.array_equal(p, q) for x, y in zip(a, c) for p, q in zip(x, y))


def test_fuzz_configurations_reach_the_paths_they_are_meant_for(oracle):
    """The oracle's coverage counters over the GPU test's configurations (tests/test_gpu_seam_fuzz.py asserts the same
    behind the comparison with the HIP kernels): the loop filter changes samples, edges between macroblocks of different QP
    get filtered, dequantised coefficients wrap their int16 store where the configuration asks for it."""
    import ctypes as C
    from tests.test_gpu_seam_fuzz import CONFIGS
    for name, mb_w, mb_h, n_pics, kw in CONFIGS:
        rng = np.random.default_rng(sum(map(ord, name)) * 7919)
        slots = kw["slots"]
        store = oracle_bind.FrameStore(mb_w, mb_h, slots)
        for s in range(slots):
            for dst, src in zip(store[s], seam_fuzz.random_frame(rng, mb_w, mb_h, "smooth" if "smooth" in name else "noise")):
                dst[:] = src
        oracle.oracle_stats_reset()
        for i in range(n_pics):
            oracle_bind.reconstruct(oracle, store, seam_fuzz.make_picture(rng, mb_w, mb_h, p_picture=(i != 2), dst_slot=i % slots, **kw))
        st = (C.c_longlong * 8)()
        oracle.oracle_stats_get(st)
        assert st[2] > 0 and st[3] > 0 and st[4] > 0 and st[5] > 0, (name, list(st))
        if kw["qp_mode"] in ("random", "two") and mb_w * mb_h >= 9:
            assert st[6] > 0, (name, list(st))
        if kw["level_style"] in ("wrap", "mixed"):
            assert st[7] > 0, (name, list(st))
        if kw.get("mirror_l1"):
            oracle.oracle_bs_by_picture.restype = C.c_longlong
            assert oracle.oracle_bs_by_picture() > 20, (name, oracle.oracle_bs_by_picture())
