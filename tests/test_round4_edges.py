"""Round-4 edge cases: in-place voxel grids (lisreg_localmap_extract grids a class cloud onto itself; the single-cloud
fallback of lisreg_voxel_downsample_multi wrote the output over records other threads were still reading), the search_mode
environment override, and front-end 0 refusing the tie / exact options it does not implement."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _records(n, seed, spread=40.0, labelled=True):
    rng = np.random.default_rng(seed)
    rec = np.zeros((n, 4), np.float32)
    rec[:, :3] = rng.uniform(-spread, spread, (n, 3)).astype(np.float32)
    rec[:, 2] *= 0.1
    if labelled:
        rec[:, 3] = rng.integers(0, 20, n).astype(np.uint32).view(np.float32)
    return rec


@pytest.mark.parametrize("n,leaf", [(150000, 0.4), (120001, 1.0), (5000, 0.2)])
def test_single_cloud_voxel_grid_in_place_equals_out_of_place(gpu_ctx, n, leaf):
    import lisreg
    rec = _records(n, 4100 + n)
    src = lisreg.DeviceArray(rec)
    dst = lisreg.DeviceArray(np.zeros_like(rec))
    rc, m = gpu_ctx.voxel_downsample_device(src.ptr, n, leaf, dst.ptr, n)
    assert rc == 0 and 0 < m < n
    want = dst.download(m)
    rc2, m2 = gpu_ctx.voxel_downsample_device(src.ptr, n, leaf, src.ptr, n)        # out == in
    assert rc2 == 0 and m2 == m
    assert np.array_equal(src.download(m).view(np.uint32), want.view(np.uint32))


def test_multi_cloud_grid_with_one_live_class_in_place(gpu_ctx):
    """five class slots, only one of them holds points (> 100 k): lisreg_voxel_downsample_multi takes its one-by-one path, with
    in[k] == out[k] exactly as lisreg_localmap_extract calls it."""
    import lisreg
    n = 130000
    rec = _records(n, 4200)
    ref_in = lisreg.DeviceArray(rec); ref_out = lisreg.DeviceArray(np.zeros_like(rec))
    rc, m = gpu_ctx.voxel_downsample_device(ref_in.ptr, n, 0.4, ref_out.ptr, n)
    assert rc == 0
    want = ref_out.download(m)
    live = lisreg.DeviceArray(rec)
    empty = lisreg.DeviceArray(np.zeros((1, 4), np.float32))
    ptrs = [empty.ptr, empty.ptr, live.ptr, empty.ptr, empty.ptr]
    counts = [0, 0, n, 0, 0]
    got = gpu_ctx.voxel_downsample_multi_device(ptrs, counts, [0.4] * 5, ptrs, [1, 1, n, 1, 1])
    assert got == [0, 0, m, 0, 0]
    assert np.array_equal(live.download(m).view(np.uint32), want.view(np.uint32))


def test_localmap_extract_with_a_single_class_over_100k_points(oracle, gpu_ctx):
    """A sliding map that only ever received ground points (> 100 k of them): extract crops and re-grids the one class in place;
    the class cloud and the surf target must equal the host restatement's."""
    import lisreg
    import replay_oracle as ro
    from lisreg import synth
    rng = np.random.default_rng(4300)
    n = 140000
    xyz = np.zeros((n, 3), np.float32)
    xyz[:, 0] = rng.uniform(-35, 35, n); xyz[:, 1] = rng.uniform(-35, 35, n); xyz[:, 2] = rng.normal(-1.7, 0.02, n)
    ground = synth.to_pcl(xyz, np.full(n, 9, np.uint16))
    none = ground[:0]
    clouds = dict(dynamic=none, ground=ground, building=none, pole=none, outlier=none)
    P = lisreg.localmap_default_params()
    T = np.zeros(6, np.float32)
    lm = ro.LocalMapOracle(ground.dtype)
    lm.insert([clouds[c] for c in ro.CLASSES], T)
    gpu_ctx.localmap_reset(7)
    info = gpu_ctx.localmap_insert(7, [clouds[c] for c in ro.CLASSES], T, P)
    assert info["n"] == [len(c) for c in lm.cls]
    T2 = np.array([0, 0, 0.01, 0.5, 0.2, 0], np.float32)
    tc, ts, isect = lm.extract(T2)
    info = gpu_ctx.localmap_extract(7, T2, P, target_slot=0)
    assert info["n_target_corner"] == len(tc) == 0 and info["n_target_surf"] == len(ts) > 1000

    def rec(cloud):
        out = np.zeros((len(cloud), 4), np.float32)
        out[:, 0], out[:, 1], out[:, 2] = cloud["x"], cloud["y"], cloud["z"]
        out[:, 3] = cloud["label"].astype(np.uint32).view(np.float32)
        return out
    assert np.array_equal(gpu_ctx.localmap_get(7, 6).view(np.uint32), rec(ts).view(np.uint32))
    for c in range(5):
        assert np.array_equal(gpu_ctx.localmap_get(7, c).view(np.uint32), rec(lm.cls[c]).view(np.uint32)), c


def test_front_end_0_refuses_canonical_ties_and_exact(gpu_ctx):
    import lisreg
    from lisreg import synth
    case = synth.make_case(h=16, w=450, m_points=20000, scan_seed=4400)
    gpu_ctx.set_target(case["tgt_corner"], case["tgt_surf"])
    p = lisreg.default_params(1)
    try:
        gpu_ctx.set_option("search_mode", 0)
        T, st, _ = gpu_ctx.align(case["src_corner"], case["src_surf"], case["T_init"], p)      # plain: fine
        assert st["status"] == 0
        for opt in ("canonical_ties", "exact_arithmetic"):
            gpu_ctx.set_option(opt, 1)
            with pytest.raises(lisreg.LisregError):
                gpu_ctx.align(case["src_corner"], case["src_surf"], case["T_init"], p)
            gpu_ctx.set_option(opt, 0)
    finally:
        gpu_ctx.set_option("canonical_ties", 0); gpu_ctx.set_option("exact_arithmetic", 0); gpu_ctx.set_option("search_mode", 4)


def test_search_mode_environment_override_is_validated():
    """LISREG_SEARCH_MODE outside {0, 1, 3, 4, 5} is ignored with a message (it used to select the graph kernel without a graph)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = f"""
import sys
sys.path.insert(0, {os.path.join(root, 'lis-slam_amd')!r})
import lisreg
from lisreg import synth
case = synth.make_case(h=16, w=450, m_points=20000, scan_seed=4500)
ctx = lisreg.Context(0)
assert ctx.get_option("search_mode") == 4
ctx.set_target(case["tgt_corner"], case["tgt_surf"])
T, st, _ = ctx.align(case["src_corner"], case["src_surf"], case["T_init"], lisreg.default_params(1))
assert st["status"] == 0
ctx.close()
print("OK")
"""
    env = dict(os.environ, LISREG_SEARCH_MODE="2")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, r.stderr
    assert "LISREG_SEARCH_MODE=2 ignored" in r.stderr


@pytest.mark.gpu
def test_cell_rows_that_do_not_fit_their_buffers_fall_back_to_the_walk():
    """search_mode 5 sizes its row buffers from the first classification of a target; a later rebuild inside a run (rebuild_targets_each_run)
    whose cloud asks for more rows gives the cells past the capacity no row (table entry -1) and their queries take the cell walk.  Forced
    here by under-sizing the buffers (LISREG_CROW_CAP_PERCENT=40): same neighbours, same bits as the cell walk; and an empty / a tiny target."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = f"""
import sys
import numpy as np
sys.path.insert(0, {os.path.join(root, 'lis-slam_amd')!r})
import lisreg
from lisreg import synth
case = synth.make_case(h=16, w=450, m_points=20000, scan_seed=4600, trans=0.4, rot_deg=2.5)
p = lisreg.default_params(1)
out = {{}}
for mode in (1, 5):
    ctx = lisreg.Context(0)
    ctx.set_option("search_mode", mode); ctx.set_option("canonical_ties", 1)
    ctx.set_target(case["tgt_corner"], case["tgt_surf"])
    out[mode] = ctx.align(case["src_corner"], case["src_surf"], case["T_init"], p)
    if mode == 5:
        g = ctx.target_cell_rows(0, 1)
        assert (g["table"] == -1).sum() > 1000 and (g["table"] >= 0).sum() > 1000, ((g["table"] == -1).sum(), (g["table"] >= 0).sum())
        # a target of four points (fewer than five: no correspondences at all) and an empty corner cloud
        ctx.set_target(case["tgt_corner"][:0], case["tgt_surf"][:4])
        T, st, _ = ctx.align(case["src_corner"], case["src_surf"], case["T_init"], p)
        assert st["status"] != 0 and np.array_equal(T, case["T_init"])
        # forty points: rows exist, almost every query finds fewer than five neighbours inside tau
        ctx.set_target(case["tgt_corner"][:5], case["tgt_surf"][:40])
        T5, st5, tr5 = ctx.align(case["src_corner"], case["src_surf"], case["T_init"], p)
        c1 = lisreg.Context(0); c1.set_option("search_mode", 1); c1.set_option("canonical_ties", 1)
        c1.set_target(case["tgt_corner"][:5], case["tgt_surf"][:40])
        T1, st1, tr1 = c1.align(case["src_corner"], case["src_surf"], case["T_init"], p)
        assert st1 == st5 and np.array_equal(T1, T5) and np.array_equal(tr1, tr5)
        c1.close()
    ctx.close()
(T1, s1, tr1), (T5, s5, tr5) = out[1], out[5]
assert s1 == s5 and s1["status"] == 0 and np.array_equal(T1, T5) and np.array_equal(tr1, tr5)
print("OK")
"""
    env = dict(os.environ, LISREG_CROW_CAP_PERCENT="40")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, r.stderr[-2000:]


@pytest.mark.gpu
def test_auto_declines_cell_rows_that_exceed_their_memory_bound(gpu_ctx):
    """search_mode auto picks the cell rows from 170 query-iterations per target point on — unless the rows a target asks for exceed
    "cell_rows_max_mb" (a cloud scattered through space instead of lying on surfaces asks for up to 125 centre rows per point): then the
    batch takes the graph scan, with the same result.  With search_mode 5 set by the caller the buffers are capped instead (the cells past
    them walk): same result again."""
    import lisreg
    from lisreg import synth
    tc, ts = synth.make_submap(20000, 42)
    scans = [synth.make_scan(32, 900, 1000 + i) for i in range(14)]       # 14 x 27 k points x 10 iterations / 20 k target points = 190
    cases = [dict(src_corner=s["corner"], src_surf=s["surf"]) for s in scans]
    T0 = np.array([synth.perturb_pose(s["T_true"], np.random.default_rng(9 + i)) for i, s in enumerate(scans)], np.float32)
    p = lisreg.default_params(1); p.fixed_iters = 10
    out = {}
    for name, mode, mb in (("auto", 4, 16384), ("auto_small", 4, 8), ("forced_small", 5, 8), ("walk", 1, 16384)):
        c = lisreg.Context(0)
        c.set_option("search_mode", mode); c.set_option("cell_rows_max_mb", mb); c.set_option("canonical_ties", 1); c.set_option("lanes_per_query", 1)
        c.set_target(tc, ts)
        out[name] = c.align_batch(cases, T0, p)
        fe = c.front_end()
        assert fe == {"auto": 5, "auto_small": 3, "forced_small": 5, "walk": 1}[name], (name, fe)
        if name == "forced_small":
            g = c.target_cell_rows(0, 1)
            assert g["n_rows"] <= 8 * 1048576 // 1032 and (g["table"] == -1).any()
        c.close()
    for name in ("auto_small", "forced_small", "walk"):
        assert np.array_equal(out["auto"][0], out[name][0]) and out["auto"][1] == out[name][1], name


@pytest.mark.gpu
def test_cell_rows_on_a_tall_grid_take_the_plain_classification(gpu_ctx):
    """A target 100 m tall (200+ cells in z) does not fit the LDS tile of the cell-row classification: the per-cell kernel takes over.
    Same neighbours as the cell walk (bit-identical registration with canonical ties); the rows cover their radius."""
    import lisreg
    from lisreg import synth
    from scipy.spatial import cKDTree
    case = synth.make_case(h=16, w=450, m_points=20000, scan_seed=4700, trans=0.3, rot_deg=2.0)
    rng = np.random.default_rng(5)
    tower = synth.to_pcl(np.stack([rng.uniform(-1, 1, 3000), rng.uniform(-1, 1, 3000), rng.uniform(0, 100, 3000)], 1).astype(np.float32))
    ts = synth.concat_clouds([case["tgt_surf"], tower])
    p = lisreg.default_params(1); p.fixed_iters = 5
    out = {}
    for mode in (1, 5):
        c = lisreg.Context(0)
        c.set_option("search_mode", mode); c.set_option("canonical_ties", 1); c.set_option("lanes_per_query", 1)
        c.set_target(case["tgt_corner"], ts)
        out[mode] = c.align(case["src_corner"], case["src_surf"], case["T_init"], p)
        if mode == 5:
            g = c.target_cell_rows(0, 1); idx = c.target_index(0, 1)
            assert idx["nz"] > 150
            pts = idx["sorted"][:, :3].astype(np.float64); tree = cKDTree(pts)
            nz, ny = idx["nz"], idx["ny"]
            have = np.flatnonzero(g["table"] >= 0)
            for cid in np.random.default_rng(1).choice(have, 300, replace=False):
                r = int(g["table"][cid]) >> 8                  # the cell's centre row
                h = np.array([cid // (nz * ny), (cid // nz) % ny, cid % nz], np.float64)
                m = idx["origin"].astype(np.float64) + (h + 0.5) * idx["cell"]
                want = set(tree.query_ball_point(m, float(np.sqrt(g["rho2"][r])) * (1 - 1e-5)))
                assert want <= set(g["ids"][r][:g["count"][r]].tolist())
        c.close()
    (T1, s1, tr1), (T5, s5, tr5) = out[1], out[5]
    assert s1 == s5 and s1["status"] == 0 and np.array_equal(T1, T5) and np.array_equal(tr1, tr5)
