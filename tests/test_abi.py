"""The C-ABI library loads on a CPU-only box, exports every symbol include/lisreg.h declares, and refuses to
compute without a HIP device (no silent fallback).  No compute calls here."""
import ctypes as C
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_exports_every_declared_symbol():
    import lisreg
    hdr = open(os.path.join(ROOT, "include", "lisreg.h")).read()
    declared = sorted(set(re.findall(r"^\s*(?:int|void\*?|const char\*)\s+(lisreg_[a-z_0-9]+)\s*\(", hdr, re.M)))
    assert len(declared) >= 25
    L = lisreg.lib()
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing, missing
    assert sorted(declared) == sorted(lisreg.ABI_SYMBOLS)


def test_struct_layouts_match_header():
    import lisreg
    # lisreg_params: 12 scalars + float[32] + 6 scalars; lisreg_item holds two pointers (8-byte aligned)
    assert C.sizeof(lisreg.Params) == 4 * (13 + 32 + 6)
    assert C.sizeof(lisreg.Stats) == 24 and C.sizeof(lisreg.Imu) == 12
    assert C.sizeof(lisreg.Item) == 56


def test_default_params_are_the_reference_literals(oracle):
    import lisreg
    for v in (1, 2, 3):
        a, b = lisreg.default_params(v), oracle.default_params(v)
        for name, _ in a._fields_:
            if name == "label_score":
                assert list(a.label_score) == list(b.label_score)
            else:
                assert getattr(a, name) == getattr(b, name), (v, name)
    p = lisreg.default_params(1)
    assert (p.max_iters, p.knn_sq_thresh, p.min_corr, p.surf_min, p.edge_min) == (15, 1.0, 50, 100, -1)
    assert lisreg.lib().lisreg_default_params(7, C.byref(p)) == lisreg.ERR_ARG


def test_host_helpers_match_oracle(oracle):
    import lisreg
    rng = np.random.default_rng(0)
    Lo = oracle.lib()
    for _ in range(20):
        T = rng.uniform(-1, 1, 6).astype(np.float32)
        M = np.zeros(12, np.float32)
        Lo.orc_pose_to_matrix(T.ctypes.data_as(C.POINTER(C.c_float)), M.ctypes.data_as(C.POINTER(C.c_float)))
        assert np.array_equal(lisreg.pose_to_matrix(T).ravel(), M)
        for imu in (None, (1, 0.1, -0.2), (1, 0.3, 1.45), (0, 0.3, 0.1)):
            for variant in (1, 3):
                po, pg = oracle.default_params(variant), lisreg.default_params(variant)
                To = T.copy()
                Lo.orc_transform_update(C.byref(po), C.byref(oracle.Imu(*imu)) if imu else None,
                                        To.ctypes.data_as(C.POINTER(C.c_float)))
                Tg = lisreg.transform_update(pg, lisreg.Imu(*imu) if imu else None, T)
                assert np.allclose(Tg, To, atol=1e-6)


def test_no_device_fails_loudly():
    import lisreg
    L = lisreg.lib()
    if L.lisreg_device_count() > 0:
        return          # on the GPU box this check is covered by the gpu tests constructing a context
    h = C.c_void_p()
    assert L.lisreg_create(0, C.byref(h)) == lisreg.ERR_HIP and not h.value
    assert b"no HIP device" in L.lisreg_last_error(None)
    try:
        lisreg.Context(0)
        raise AssertionError("Context() must raise without a device")
    except lisreg.LisregError as e:
        assert e.code == lisreg.ERR_HIP
