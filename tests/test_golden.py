"""Committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py from the numpy mirror):
the C oracle must reproduce them on CPU; the HIP library must reproduce them on the GPU (bar: 1e-3 m / 1e-3 rad)."""
import glob
import os

import numpy as np
import pytest

from helpers import copy_params, pose_err

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


def load(path):
    from lisreg import synth
    z = np.load(path)
    def cloud(a):
        return synth.to_pcl(np.ascontiguousarray(a[:, :3]), a[:, 3].astype(np.uint16))
    return z, dict(tgt_corner=cloud(z["tgt_corner"]), tgt_surf=cloud(z["tgt_surf"]), src_corner=cloud(z["src_corner"]),
                   src_surf=cloud(z["src_surf"]), T_init=z["T_init"])


def check(z, T, st, tr, tol_pose, tol_ncorr):
    iters, deg, n_last, status = [int(v) for v in z["stats"]]
    assert st["status"] == status and st["iters"] == iters and st["degenerate"] == deg
    rot, trn = pose_err(T, z["T_expected"])
    assert rot <= tol_pose and trn <= tol_pose, (rot, trn)
    g = z["trace"]
    assert len(tr) == len(g)
    for k in range(len(g)):
        assert abs(tr[k, 0] - g[k, 0]) <= tol_ncorr
        assert tr[k, 55] == g[k, 55]
        r, t = pose_err(tr[k, 49:55], g[k, 49:55])
        assert r <= tol_pose and t <= tol_pose, (k, r, t)
        if g[k, 55]:
            assert np.abs(tr[k, 1:37] - g[k, 1:37]).max() <= 2e-3 * np.abs(g[k, 1:37]).max()


def test_golden_files_present():
    assert len(GOLDEN) >= 5


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_oracle_reproduces_golden(oracle, path):
    z, case = load(path)
    p = oracle.default_params(int(z["variant"]))
    p.fixed_iters = int(z["fixed_iters"])
    p.use_imu_blend = 0
    T, st, tr = oracle.align(case["tgt_corner"], case["tgt_surf"], case["src_corner"], case["src_surf"], case["T_init"], p,
                             degenerate_in=int(z["degenerate_in"]))
    check(z, T, st, tr, tol_pose=5e-5, tol_ncorr=2)


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_hip_reproduces_golden(oracle, gpu_ctx, path):
    import lisreg
    z, case = load(path)
    p = lisreg.default_params(int(z["variant"]))
    p.fixed_iters = int(z["fixed_iters"])
    p.use_imu_blend = 0
    gpu_ctx.set_target(case["tgt_corner"], case["tgt_surf"])
    T, st, tr = gpu_ctx.align(case["src_corner"], case["src_surf"], case["T_init"], p)
    check(z, T, st, tr, tol_pose=1e-3, tol_ncorr=3)
