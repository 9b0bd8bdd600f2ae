"""The CABAC arithmetic decoding engine of the host entropy layer (csrc/host/cabac.h; SURVEY 8f rank 4, 8c "partial
pins") against the reference: tests/golden/kat_cabac.npz holds random bin sequences as the REAL reference's encoder wrote
them (core/cabac.c:907-1018, recorded by tests/golden/make_kat_cabac.py); the product must decode every bin back - context
initialisation for I and P/B tables, every cabac_init_idc and slice QP, decisions, bypass and terminate bins.  CPU only."""
import ctypes as C
import os

import numpy as np
import pytest


@pytest.fixture(scope="module")
def kat():
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kat_cabac.npz"))


def decode(lib, data, is_i, idc, qp, ops):
    lib.p264cabac_decode_ops.restype = C.c_int
    lib.p264cabac_decode_ops.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    ops = np.ascontiguousarray(ops, np.int16)
    data = np.ascontiguousarray(data, np.uint8)
    bins = np.zeros(len(ops), np.uint8)
    rc = lib.p264cabac_decode_ops(data.ctypes.data, len(data), is_i, idc, qp, ops.ctypes.data, len(ops), bins.ctypes.data)
    return rc, bins


def test_engine_decodes_what_the_reference_encoded(lib, kat):
    seen_qp, seen_tab = set(), set()
    for ops, bins, (is_i, idc, qp, n), data in zip(kat["ops"], kat["bins"], kat["par"].tolist(), kat["data"]):
        rc, got = decode(lib, data[:n], is_i, idc, qp, ops)
        assert rc == 0, "read past the end of the data (qp %d)" % qp
        bad = np.nonzero(got != bins)[0]
        assert len(bad) == 0, "I %d idc %d qp %d: bin %d of %d differs (op %d)" % (is_i, idc, qp, bad[0], len(ops), ops[bad[0]])
        seen_qp.add(qp)
        seen_tab.add((is_i, 0 if is_i else idc))
    assert seen_qp == set(range(52)) and seen_tab == {(1, 0), (0, 0), (0, 1), (0, 2)}
    assert (kat["ops"] == -1).sum() > 1000 and (kat["ops"] == -2).sum() > 100


def test_engine_is_sensitive_to_its_inputs(lib, kat):
    """The same bytes with the wrong context table, QP or a flipped byte do NOT give the bins back: the test above is not
    satisfied by accident."""
    ops, bins, (is_i, idc, qp, n), data = kat["ops"][1], kat["bins"][1], kat["par"][1].tolist(), kat["data"][1]
    assert (decode(lib, data[:n], is_i, idc, qp, ops)[1] == bins).all()
    assert not (decode(lib, data[:n], 1 - is_i, idc, qp, ops)[1] == bins).all()
    assert not (decode(lib, data[:n], is_i, idc, (qp + 17) % 52, ops)[1] == bins).all()
    d2 = data[:n].copy(); d2[5] ^= 0x10
    assert not (decode(lib, d2, is_i, idc, qp, ops)[1] == bins).all()
    rc, _ = decode(lib, data[:8], is_i, idc, qp, ops)
    assert rc == 1                                          # truncated data: reported, no crash
    assert lib.p264cabac_decode_ops(None, 0, 0, 0, 0, None, 0, None) == -1


def test_context_initialisation_covers_every_context(lib):
    """All 460 contexts, I and P tables, give a valid first decision at every QP (the reference initialises only the first
    436, core/cabac.c:835; the rest are the field-coding contexts of 9.3.1.1 and must not be left undefined here)."""
    data = np.arange(64, dtype=np.uint8)
    for is_i in (0, 1):
        for qp in (0, 26, 51):
            rc, bins = decode(lib, data, is_i, 1, qp, np.arange(460, dtype=np.int16))
            assert rc == 0 and set(bins.tolist()) <= {0, 1}
