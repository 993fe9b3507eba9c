"""Committed golden vectors for the SURVEY.md §8 f rows (tests/golden/next/next_rows.npz, made by make_golden_next.py from
numpy / scipy mirrors only): the C oracle must reproduce them on the CPU, the HIP library on the GPU.  Integer / index
results exactly; floating-point results within the stated tolerances."""
import os

import numpy as np
import pytest

f32 = np.float32
PATH = os.path.join(os.path.dirname(__file__), "golden", "next", "next_rows.npz")


@pytest.fixture(scope="module")
def G():
    return np.load(PATH)


def _cloud(G, name, dtype):
    n = len(G[f"{name}.x"])
    c = np.zeros(n, dtype)
    for f in dtype.names:
        c[f] = G[f"{name}.{f}"]
    return c


def _pose_diff(A, B):
    A, B = np.asarray(A, np.float64), np.asarray(B, np.float64)
    R = A[:3, :3].T @ B[:3, :3]
    v = 0.5 * np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    return float(np.arctan2(np.linalg.norm(v), (np.trace(R) - 1) / 2)), float(np.abs(A[:3, 3] - B[:3, 3]).max())


class _Impl:
    """One interface over the two implementations under test."""

    def __init__(self, kind, oracle=None, ctx=None):
        self.kind, self.o, self.c = kind, oracle, ctx

    def voxel(self, c, leaf):
        return self.o.voxel_grid(c, leaf) if self.kind == "oracle" else self.c.voxel_downsample(c, leaf)

    def features(self, raw, dk_args):
        if self.kind == "oracle":
            p = self.o.FeatureParams(8, 240, 1, 0.0, 70.0, 1.0, 0.1)
            r = self.o.extract_features(raw, p)
            dk = self.o.make_deskew(*dk_args)
            return r, self.o.deskew_points(raw, dk, r["deskewed"])
        import lisreg
        from lisreg import synth
        p = lisreg.FeatureParams(8, 240, 1, 0.0, 70.0, 1.0, 0.1)
        plain = self.c.extract_features(raw, p)
        dsk = self.c.extract_features(raw, p, lisreg.make_deskew(*dk_args))
        # recover index lists from the (unique) time stamps of the returned points
        t2i = {float(t): i for i, t in enumerate(raw["time"])}
        r = {k: np.array([t2i[float(t)] for t in v["time"]], np.int32) for k, v in plain.items()}
        return r, synth.pcl_xyz(dsk["deskewed"])

    def map_ops(self, m, q, box):
        if self.kind == "oracle":
            _, d2 = self.o.nearest(m, q)
            kept, _ = self.o.dynamic_filter(m, q, 25.0, 0.3, 1.0, 0.05)
            return d2, kept, self.o.bbx_filter(q, box), self.o.bbx_filter(q, box, True), self.o.cloud_bounds(q)
        self.c.map_index_set(30, m)
        _, d2 = self.c.nearest(30, q)
        kept, _ = self.c.dynamic_filter(30, q, 25.0, 0.3, 1.0, 0.05)
        return d2, kept, self.c.bbx_filter(q, box), self.c.bbx_filter(q, box, True), self.c.cloud_bounds(q)

    def icp(self, tgt, src):
        if self.kind == "oracle":
            return self.o.icp_align(tgt, src, self.o.icp_default_params(0)), self.o.icp_gn_match(tgt, src, 12, 4.0, np.eye(4, dtype=f32))
        import lisreg
        self.c.map_index_set(31, tgt)
        return self.c.icp_align(31, src, lisreg.icp_default_params(0)), self.c.icp_gn_match(31, src, 12, 4.0, np.eye(4, dtype=f32))


def _check_all(G, impl):
    from lisreg import synth
    # f-1
    cv = _cloud(G, "vox_in", synth.PCL_DTYPE)
    st, ds = impl.voxel(cv, float(G["vox_leaf"]))
    assert st == 0 and len(ds) == len(G["vox_centroid"])
    assert np.array_equal(ds["label"], G["vox_label"])
    assert np.abs(synth.pcl_xyz(ds) - G["vox_centroid"][:, :3]).max() < 2e-5 and np.abs(ds["intensity"] - G["vox_centroid"][:, 3]).max() < 2e-3
    # f-2
    raw = _cloud(G, "feat_in", synth.XYZIRT_DTYPE)
    r, dxyz = impl.features(raw, (G["imu_time"], G["imu_rot"][:, 0], G["imu_rot"][:, 1], G["imu_rot"][:, 2], 100.0))
    for k in ("deskewed", "corner", "surface", "corner_sharp", "surface_sharp"):
        assert np.array_equal(r[k], G[f"feat_{k}"]), k
    assert np.abs(dxyz - G["deskew_xyz"]).max() < 5e-5
    # f-3
    m, q = _cloud(G, "map", synth.PCL_DTYPE), _cloud(G, "map_q", synth.PCL_DTYPE)
    d2, kept, inside, outside, bounds = impl.map_ops(m, q, G["box"])
    assert np.allclose(d2, G["nn_d2"], rtol=3e-7, atol=0)
    same = lambda a, b: len(a) == len(b) and all(np.array_equal(a[f], b[f]) for f in b.dtype.names)
    assert same(kept, q[G["dyn_keep"]]) and same(inside, q[G["box_inside"]]) and same(outside, q[~G["box_inside"]])
    assert np.array_equal(bounds, G["bounds"])
    # f-4
    tgt, src = _cloud(G, "icp_tgt", synth.PCL_DTYPE), _cloud(G, "icp_src", synth.PCL_DTYPE)
    ri, rg = impl.icp(tgt, src)
    assert ri["converged"] and ri["iters"] == int(G["icp_iters"]) and ri["state"] == int(G["icp_state"])
    dr, dt = _pose_diff(ri["T"], G["icp_T"])
    assert dr < 2e-4 and dt < 1e-3, (dr, dt)
    dr, dt = _pose_diff(rg["T"], G["gn_T"])
    assert rg["steps_applied"] == 12 and dr < 2e-4 and dt < 2e-3, (dr, dt)
    assert abs(rg["fitness"] - float(G["gn_fitness"])) < 5e-3 * float(G["gn_fitness"])


def test_fixture_present(G):
    assert len(G.files) > 40


def test_oracle_reproduces_next_row_goldens(oracle, G):
    _check_all(G, _Impl("oracle", oracle=oracle))


@pytest.mark.gpu
def test_hip_reproduces_next_row_goldens(gpu_ctx, G):
    _check_all(G, _Impl("hip", ctx=gpu_ctx))
