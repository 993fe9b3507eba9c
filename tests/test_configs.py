"""BASELINE.json configs, each at its FULL size, HIP path (through the C ABI) against the CPU oracle.

  configs[0]  one 64x1800 scan vs a 50 k-point submap, 10 fixed GN iterations      -> test_cfg0_single_scan_50k
  configs[1]  batch of 64x1800 scans vs one shared 200 k-point submap              -> test_cfg1_batch_vs_oracle
  configs[2]  sequential replay with the semantic mask                             -> tests/test_replay.py (synthetic drive;
                                                                                      KITTI itself: tools/kitti_replay.py)
  configs[3]  256 independent registrations per GPU, each against its OWN target   -> test_cfg3_own_targets_256
  configs[4]  128x2048 scans vs a 1 M-point submap, 30 fixed GN iterations         -> test_cfg4_dense_1m

Synthetic KITTI-shape data (no dataset in this environment); the bar is BASELINE.md §4: pose within 1e-3 rad / 1e-3 m of
the oracle after the same iteration count, per-iteration correspondence counts within the handful of threshold
straddlers fp32 contraction can flip."""
import numpy as np
import pytest

from helpers import copy_params, pose_err

pytestmark = pytest.mark.gpu

TOL = 1e-3
ORACLE_THREADS = 16


def _dev(arr):
    import lisreg
    return lisreg.DeviceArray(lisreg.pack_device_records(arr))


def test_cfg0_single_scan_50k(oracle, gpu_ctx):
    """configs[0] on synthetic data: ONE 64x1800 scan vs a 50 k-point submap, 10 fixed iterations; pose, every iteration's pose
    and the per-iteration correspondence counts against the oracle."""
    import lisreg
    from lisreg import synth
    case = synth.make_case(h=64, w=1800, m_points=50000, scan_seed=1000)
    p_o = oracle.default_params(1); p_o.fixed_iters = 10
    p = copy_params(p_o, lisreg.Params)
    To, so, tro = oracle.align(case["tgt_corner"], case["tgt_surf"], case["src_corner"], case["src_surf"], case["T_init"], p_o,
                               n_threads=ORACLE_THREADS)
    gpu_ctx.set_target(case["tgt_corner"], case["tgt_surf"])
    T, st, tr = gpu_ctx.align(case["src_corner"], case["src_surf"], case["T_init"], p)
    assert st["status"] == so["status"] == 0 and st["iters"] == so["iters"] == 10 and len(tr) == len(tro) == 10
    assert max(pose_err(T, To)) <= TOL
    for k in range(10):
        assert max(pose_err(tr[k, 49:55], tro[k, 49:55])) <= TOL, k
        assert abs(tr[k, 0] - tro[k, 0]) <= max(3, 0.001 * tro[k, 0]), (k, tr[k, 0], tro[k, 0])


def test_cfg1_batch_vs_oracle(oracle, gpu_ctx):
    """configs[1] at full size: a device-resident batch of 64x1800 scans against one shared 200 k-point submap, 10 fixed
    iterations, index rebuilt inside the run (what bench.py times) — 12 of the items against the oracle, and the batch
    against single calls bit for bit."""
    import lisreg
    from lisreg import synth
    tc, ts = synth.make_submap(200000, 42)
    n = 12
    scans = [synth.make_scan(64, 1800, 1000 + i) for i in range(n)]
    T0 = np.array([synth.perturb_pose(s["T_true"], np.random.default_rng(1000 + i + 7919)) for i, s in enumerate(scans)], np.float32)
    p_o = oracle.default_params(1); p_o.fixed_iters = 10
    p = copy_params(p_o, lisreg.Params)
    tcd, tsd = _dev(tc), _dev(ts)
    recs = [(_dev(s["corner"]), _dev(s["surf"])) for s in scans]
    gpu_ctx.set_target_device(tcd.ptr, len(tc), tsd.ptr, len(ts))
    items = [dict(corner_ptr=a.ptr, n_corner=a.shape[0], surf_ptr=b.ptr, n_surf=b.shape[0]) for a, b in recs]
    gpu_ctx.set_option("rebuild_targets_each_run", 1)
    gpu_ctx.set_option("canonical_ties", 1)          # equal distances resolved by (distance, original index): bits independent of the front-end
    try:
        gpu_ctx.batch_prepare_device(items, T0, p)
        gpu_ctx.batch_run()
        T, st = gpu_ctx.batch_fetch()
    finally:
        gpu_ctx.set_option("rebuild_targets_each_run", 0)
    worst = 0.0
    for i, s in enumerate(scans):
        To, so, _ = oracle.align(tc, ts, s["corner"], s["surf"], T0[i], p_o, n_threads=ORACLE_THREADS, max_trace=1)
        assert st[i]["status"] == so["status"] == 0 and st[i]["iters"] == so["iters"] == 10
        assert abs(st[i]["n_corr_last"] - so["n_corr_last"]) <= max(3, 0.001 * so["n_corr_last"])
        worst = max(worst, *pose_err(T[i], To))
    assert worst <= TOL, worst
    # batch == single calls, bit for bit, with the front-end left at auto: the batch qualifies for the graph scan, a single
    # registration takes the eight-lanes-per-query cell walk — equal distances are resolved by (distance, original index) in all of
    # them with the option "canonical_ties" (LISREG_CANONICAL_FIVE), so a frame's bits do not depend on the batch it sits in
    assert gpu_ctx.get_option("front_end") == 3           # 12 scans: 69 query-iterations per target point — the graph scan (64 scans: the cell rows)
    # the same batch through the cell rows (what auto picks for the 64 scans of bench.py): the same bits
    gpu_ctx.set_option("search_mode", 5); gpu_ctx.set_option("rebuild_targets_each_run", 1)
    try:
        gpu_ctx.batch_prepare_device(items, T0, p)
        assert gpu_ctx.get_option("front_end") == 5
        gpu_ctx.batch_run()
        T5, st5 = gpu_ctx.batch_fetch()
    finally:
        gpu_ctx.set_option("search_mode", 4); gpu_ctx.set_option("rebuild_targets_each_run", 0)
    assert np.array_equal(T5, T) and st5 == st
    try:
        gpu_ctx.set_target(tc, ts)
        for i in (0, n - 1):
            Ts, ss, _ = gpu_ctx.align(scans[i]["corner"], scans[i]["surf"], T0[i], p)
            assert gpu_ctx.get_option("front_end") == 1
            assert np.array_equal(Ts, T[i]) and ss == st[i]
    finally:
        gpu_ctx.set_option("canonical_ties", 0)


def test_cfg1_all_64_scans_of_the_bench_vs_oracle(oracle, gpu_ctx):
    """The batch `bench.py` times, all of it: 64 device-resident 64x1800 scans against the shared 200 k-point submap,
    10 fixed iterations, index and cell rows rebuilt inside the run, every option at its default (front-end by auto = the cell rows,
    rows filtered by the batch's query marks, equal distances NOT canonicalised) — every one of the 64 poses against the oracle's, and a
    second run of the prepared batch against the first bit for bit."""
    import lisreg
    from lisreg import synth
    tc, ts = synth.make_submap(200000, 42)
    n = 64
    scans = [synth.make_scan(64, 1800, 1000 + i) for i in range(n)]
    T0 = np.array([synth.perturb_pose(s["T_true"], np.random.default_rng(1000 + i + 7919)) for i, s in enumerate(scans)], np.float32)
    p_o = oracle.default_params(1); p_o.fixed_iters = 10
    p = copy_params(p_o, lisreg.Params)
    tcd, tsd = _dev(tc), _dev(ts)
    recs = [(_dev(s["corner"]), _dev(s["surf"])) for s in scans]
    gpu_ctx.set_target_device(tcd.ptr, len(tc), tsd.ptr, len(ts))
    items = [dict(corner_ptr=a.ptr, n_corner=a.shape[0], surf_ptr=b.ptr, n_surf=b.shape[0]) for a, b in recs]
    gpu_ctx.set_option("rebuild_targets_each_run", 1)
    try:
        gpu_ctx.batch_prepare_device(items, T0, p)
        assert gpu_ctx.get_option("front_end") == 5
        gpu_ctx.batch_run()
        T, st = gpu_ctx.batch_fetch()
        assert gpu_ctx.get_option("row_reach_now") == 1
        gpu_ctx.batch_run()
        T2, st2 = gpu_ctx.batch_fetch()
    finally:
        gpu_ctx.set_option("rebuild_targets_each_run", 0)
    assert np.array_equal(T, T2) and st == st2
    worst_r = worst_t = 0.0
    for i, s in enumerate(scans):
        To, so, _ = oracle.align(tc, ts, s["corner"], s["surf"], T0[i], p_o, n_threads=ORACLE_THREADS, max_trace=1)
        assert st[i]["status"] == so["status"] == 0 and st[i]["iters"] == so["iters"] == 10
        assert abs(st[i]["n_corr_last"] - so["n_corr_last"]) <= max(3, 0.001 * so["n_corr_last"])
        r, t = pose_err(T[i], To)
        worst_r, worst_t = max(worst_r, r), max(worst_t, t)
    print(f"[cfg1, 64 of 64] worst pose difference vs the oracle: {worst_r:.2e} rad, {worst_t:.2e} m")
    assert worst_r <= TOL and worst_t <= TOL


def test_cfg3_own_targets_256(oracle, gpu_ctx):
    """configs[3] shape on one GPU: 256 loop-closure style registrations, each against its OWN 200 k-point target (distinct
    seeds, one slot per item, all 512 indexes rebuilt inside the run); 8 sampled items against the oracle, and duplicates of
    an item (same scan, same target seed) must agree bit for bit wherever they sit in the batch."""
    import lisreg
    from lisreg import synth
    ctx = lisreg.Context(0)
    n, n_scans, n_maps = 256, 16, 32
    scans = [synth.make_scan(64, 1800, 3000 + i) for i in range(n_scans)]
    maps = [synth.make_submap(200000, 500 + m) for m in range(n_maps)]
    srec = [(_dev(s["corner"]), _dev(s["surf"])) for s in scans]
    T0s = [synth.perturb_pose(s["T_true"], np.random.default_rng(77 + i)) for i, s in enumerate(scans)]
    keep, items, T0 = [], [], []
    for i in range(n):
        m, s = i % n_maps, (i * 7) % n_scans
        a, b = _dev(maps[m][0]), _dev(maps[m][1])          # every item owns its copy of the target in its own slot
        keep.append((a, b))
        ctx.set_target_device(a.ptr, a.shape[0], b.ptr, b.shape[0], slot=i)
        items.append(dict(corner_ptr=srec[s][0].ptr, n_corner=srec[s][0].shape[0], surf_ptr=srec[s][1].ptr,
                          n_surf=srec[s][1].shape[0], target=i))
        T0.append(T0s[s])
    T0 = np.array(T0, np.float32)
    p_o = oracle.default_params(1); p_o.fixed_iters = 10
    p = copy_params(p_o, lisreg.Params)
    ctx.set_option("rebuild_targets_each_run", 1)
    ctx.batch_prepare_device(items, T0, p)
    assert ctx.front_end() == 1                            # one-shot targets: the cell walk, no graph build
    ctx.batch_run()
    T, st = ctx.batch_fetch()
    assert all(s["status"] == 0 and s["iters"] == 10 for s in st)
    for i in (0, 37, 74, 111, 148, 185, 222, 255):
        m, s = i % n_maps, (i * 7) % n_scans
        To, so, _ = oracle.align(maps[m][0], maps[m][1], scans[s]["corner"], scans[s]["surf"], T0[i], p_o,
                                 n_threads=ORACLE_THREADS, max_trace=1)
        assert max(pose_err(T[i], To)) <= TOL, i
        assert abs(st[i]["n_corr_last"] - so["n_corr_last"]) <= max(3, 0.001 * so["n_corr_last"])
    first = {}
    for i in range(n):
        key = (i % n_maps, (i * 7) % n_scans)
        if key in first:
            assert np.array_equal(T[i], T[first[key]]) and st[i] == st[first[key]], (i, first[key])
        else:
            first[key] = i
    ctx.close()


def test_cfg4_dense_1m(oracle, gpu_ctx):
    """configs[4]: 128x2048 scans against a 1 M-point submap, 30 fixed iterations — one item against the oracle over all 30
    iterations, an 8-item device batch against single calls, in the three search front-ends (cell walk, graph scan, cell rows)."""
    import lisreg
    from lisreg import synth
    tc, ts = synth.make_submap(1_000_000, 42)
    scans = [synth.make_scan(128, 2048, 5000 + i) for i in range(8)]
    T0 = np.array([synth.perturb_pose(s["T_true"], np.random.default_rng(31 + i)) for i, s in enumerate(scans)], np.float32)
    p_o = oracle.default_params(1); p_o.fixed_iters = 30
    p = copy_params(p_o, lisreg.Params)
    To, so, tro = oracle.align(tc, ts, scans[0]["corner"], scans[0]["surf"], T0[0], p_o, n_threads=ORACLE_THREADS, max_trace=30)
    tcd, tsd = _dev(tc), _dev(ts)
    recs = [(_dev(s["corner"]), _dev(s["surf"])) for s in scans]
    items = [dict(corner_ptr=a.ptr, n_corner=a.shape[0], surf_ptr=b.ptr, n_surf=b.shape[0]) for a, b in recs]
    results = {}
    for mode in (1, 3, 5):
        ctx = lisreg.Context(0)
        ctx.set_option("search_mode", mode)
        ctx.set_option("canonical_ties", 1)
        ctx.set_target_device(tcd.ptr, len(tc), tsd.ptr, len(ts))
        ctx.batch_prepare_device(items, T0, p)
        assert ctx.front_end() == mode
        ctx.batch_run()
        T, st = ctx.batch_fetch()
        assert all(s["status"] == 0 and s["iters"] == 30 for s in st)
        assert max(pose_err(T[0], To)) <= TOL
        Ts, ss, tr = ctx.align(scans[0]["corner"], scans[0]["surf"], T0[0], p)
        assert np.array_equal(Ts, T[0]) and ss == st[0] and len(tr) == 30
        for k in range(30):
            assert max(pose_err(tr[k, 49:55], tro[k, 49:55])) <= TOL, (mode, k)
            assert abs(tr[k, 0] - tro[k, 0]) <= max(3, 0.001 * tro[k, 0]), (mode, k, tr[k, 0], tro[k, 0])
        results[mode] = (T, st)
        ctx.close()
    # same neighbours, same order -> same bits: 63 M query-iterations against a 1 M-point map do meet candidates at exactly equal
    # float distance; with "canonical_ties" both front-ends resolve them by (distance, original index)
    assert np.array_equal(results[1][0], results[3][0])
    assert results[1][1] == results[3][1]
    assert np.array_equal(results[1][0], results[5][0]) and results[1][1] == results[5][1]
