"""Exact-neighbour parity of the search front-ends (SURVEY.md "hard parts": the grid / graph search must return the TRUE five
nearest, as pcl::KdTreeFLANN::nearestKSearch(5) does, odomEstimationNode.cpp:655 / :774).

The library's test hook (option "dump_neighbors", lisreg_get_neighbors) hands back, for every source point, the original
target indices the last GN iteration used.  They are compared with an independent exact search (scipy cKDTree in float64 on
the float32-transformed points): the SETS must be identical, except where two candidates are equidistant to within float32
rounding of the squared distance (then either is a correct answer of a float32 kNN; such queries are counted and bounded).
The cell walk (mode 1) and the k-NN graph scan (mode 3) are also compared with each other."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _reference_sets(tgt_xyz, q, tau):
    from scipy.spatial import cKDTree
    tree = cKDTree(tgt_xyz.astype(np.float64))
    d, i = tree.query(q.astype(np.float64), k=7, workers=8)
    return d, i


def _check(ids_gpu, src_xyz, tgt_xyz, T, tau, what):
    """ids_gpu [5, n]; returns number of rounding-level ties tolerated"""
    import lisreg
    M = lisreg.pose_to_matrix(np.asarray(T, np.float32)).astype(np.float64)
    q = src_xyz.astype(np.float64) @ M[:, :3].T + M[:, 3]
    n = len(q)
    if len(tgt_xyz) < 5:
        assert (ids_gpu < 0).all()
        return 0
    d, i = _reference_sets(tgt_xyz, q.astype(np.float32), tau)
    found_ref = d[:, 4] ** 2 < tau
    found_gpu = ids_gpu[4] >= 0
    bad = 0
    # The device evaluates q = M p and d^2 in float32 (a couple of ulps of the coordinates apart from this float64 check),
    # so two candidates whose squared distances differ by less than 2 d (3 ulp(|q|)) + 1e-6 d^2 may legitimately swap.
    ulp = np.spacing(np.float32(max(1.0, float(np.abs(q).max())))).astype(np.float64)
    # found / not found may only differ when the 5th distance sits on tau
    diff_found = np.nonzero(found_ref != found_gpu)[0]
    for k in diff_found:
        assert abs(d[k, 4] ** 2 - tau) <= 2.0 * d[k, 4] * 3.0 * ulp + 1e-6 * tau, (what, "found flag", k, d[k, 4] ** 2)
        bad += 1
    both = np.nonzero(found_ref & found_gpu)[0]
    g = np.sort(ids_gpu[:, both].T, axis=1)
    r = np.sort(i[both, :5], axis=1)
    mism = both[np.nonzero((g != r).any(1))[0]]
    for k in mism:
        gs, rs = set(ids_gpu[:, k].tolist()), set(i[k, :5].tolist())
        extra, missing = sorted(gs - rs), sorted(rs - gs)
        assert len(extra) == len(missing), (what, k)
        de = np.sort(np.linalg.norm(tgt_xyz[extra].astype(np.float64) - q[k], axis=1) ** 2)
        dm = np.sort(np.linalg.norm(tgt_xyz[missing].astype(np.float64) - q[k], axis=1) ** 2)
        assert np.all(np.abs(de - dm) <= 2.0 * np.sqrt(dm) * 3.0 * ulp + 1e-6 * dm), (what, "wrong neighbour", k, de, dm)
        bad += 1
    assert bad <= max(2, 1e-3 * n), (what, bad, n)
    return bad


@pytest.mark.parametrize("h,w,m_points,variant", [(16, 450, 20000, 1), (32, 900, 60000, 2), (64, 1800, 200000, 1), (128, 2048, 1000000, 1)])
def test_neighbour_sets_are_exact(h, w, m_points, variant):
    import lisreg
    from lisreg import synth
    case = synth.make_case(h=h, w=w, m_points=m_points, scan_seed=4321, labelled=variant != 1)
    tau = lisreg.default_params(variant).knn_sq_thresh
    src = [synth.pcl_xyz(case["src_corner"]), synth.pcl_xyz(case["src_surf"])]
    tgt = [synth.pcl_xyz(case["tgt_corner"]), synth.pcl_xyz(case["tgt_surf"])]
    nc, ns = len(src[0]), len(src[1])
    dumps = {}
    for iters in (1, 4):
        for mode in (1, 3, 5):
            ctx = lisreg.Context(0)
            ctx.set_option("search_mode", mode); ctx.set_option("dump_neighbors", 1)
            ctx.set_target(case["tgt_corner"], case["tgt_surf"])
            p = lisreg.default_params(variant); p.fixed_iters = iters
            T, st, tr = ctx.align(case["src_corner"], case["src_surf"], case["T_init"], p)
            assert ctx.front_end() == mode and st["iters"] == iters
            ids = ctx.neighbors(nc + ns)[:5]
            ctx.close()
            T_search = case["T_init"] if iters == 1 else tr[iters - 2, 49:55]      # the pose the last iteration searched at
            ties = _check(ids[:, :nc], src[0], tgt[0], T_search, tau, (mode, iters, "corner"))
            ties += _check(ids[:, nc:], src[1], tgt[1], T_search, tau, (mode, iters, "surf"))
            dumps[(mode, iters)] = (ids, tr, ties)
        a, b = dumps[(1, iters)], dumps[(3, iters)]
        # the two front-ends against each other: identical neighbours in identical order (up to the same rounding-level ties)
        differing = int((a[0] != b[0]).any(0).sum())
        print(f"{h}x{w} vs {m_points}: iteration {iters - 1}: rounding-level ties vs float64 search: walk {a[2]}, graph {b[2]}; "
              f"queries where walk and graph differ: {differing} of {nc + ns}")
        assert differing <= a[2] + b[2] + 2, (iters, differing)
        if differing == 0:
            assert np.array_equal(a[1], b[1])
        cells = dumps[(5, iters)]                              # cell rows (search_mode 5) against the walk, the same way
        differing = int((a[0] != cells[0]).any(0).sum())
        print(f"{h}x{w} vs {m_points}: iteration {iters - 1}: cell rows: rounding-level ties {cells[2]}; queries where walk and cell rows differ: {differing}")
        assert differing <= a[2] + cells[2] + 2, (iters, differing)
        if differing == 0:
            assert np.array_equal(a[1], cells[1])


@pytest.mark.gpu
def test_equal_distances_resolve_by_original_index_in_every_front_end(oracle):
    """Option "canonical_ties".  A target in which EVERY point exists twice (exact copies, far apart in the cloud) and a mirror-symmetric part: each query meets
    exact distance ties inside its five neighbours and at the fifth / sixth place.  The cell walk with one and with eight lanes per
    query, the graph scan, and the exact-arithmetic build must return the same five ORIGINAL indices in the same order — the smaller
    index first among equals — and the registration must then agree bit for bit between them, and in its integer outputs with the
    oracle (whose k-NN resolves equal distances the same way)."""
    import lisreg
    from lisreg import synth
    from helpers import copy_params
    case = synth.make_case(h=16, w=450, m_points=12000, scan_seed=4711, trans=0.2, rot_deg=1.0)

    def doubled(cloud, seed):
        rng = np.random.default_rng(seed)
        both = synth.concat_clouds([cloud, cloud])
        return both[rng.permutation(len(both))]
    tc, ts = doubled(case["tgt_corner"], 1), doubled(case["tgt_surf"], 2)
    p = lisreg.default_params(1)
    p.fixed_iters = 3
    n = len(case["src_corner"]) + len(case["src_surf"])
    out = {}
    for name, mode, lanes, exact in (("walk1", 1, 1, 0), ("walk8", 1, 0, 0), ("graph", 3, 1, 0), ("cells", 5, 1, 0), ("exact", 1, 1, 1)):
        c = lisreg.Context(0)
        c.set_option("search_mode", mode); c.set_option("lanes_per_query", lanes); c.set_option("exact_arithmetic", exact)
        c.set_option("canonical_ties", 1)                      # (implied by exact_arithmetic)
        c.set_option("dump_neighbors", 1)
        c.set_target(tc, ts)
        T, st, tr = c.align(case["src_corner"], case["src_surf"], case["T_init"], p)
        out[name] = (T, st, tr, c.neighbors(n)[:5].copy())
        c.close()
    nb = out["walk1"][3]
    found = nb[4] >= 0
    assert found.sum() > 0.5 * n
    # every query with five neighbours has ties: its neighbours come in exact pairs
    txyz = {0: synth.pcl_xyz(tc), 1: synth.pcl_xyz(ts)}
    nc = len(case["src_corner"])
    for q in np.flatnonzero(found)[:200]:
        xyz = txyz[0 if q < nc else 1][nb[:, q]]
        pairs = sum(np.array_equal(xyz[k], xyz[k + 1]) for k in range(4))
        assert pairs >= 2, (q, xyz)
        for k in range(4):
            if np.array_equal(xyz[k], xyz[k + 1]):
                assert nb[k, q] < nb[k + 1, q]                              # the smaller original index first among equals
    for name in ("walk8", "graph", "cells"):
        # (queries with fewer than five neighbours inside tau contribute nothing; the partial lists they keep are not canonicalised)
        assert np.array_equal(out[name][3][4] >= 0, found), name
        bad = np.flatnonzero((out[name][3][:, found] != nb[:, found]).any(0))
        assert len(bad) == 0, (name, len(bad), out[name][3][:, found][:, bad[:3]], nb[:, found][:, bad[:3]])
        assert np.array_equal(out[name][0], out["walk1"][0]) and np.array_equal(out[name][2], out["walk1"][2]), name
    assert np.array_equal(out["exact"][3][:, found], nb[:, found])
    p_o = copy_params(p, oracle.Params)
    To, so, tro = oracle.align(tc, ts, case["src_corner"], case["src_surf"], case["T_init"], p_o)
    assert np.array_equal(out["exact"][2][:, 0], tro[:, 0])
    assert max(np.abs(out["exact"][0] - To).max(), 0) <= 2e-6
