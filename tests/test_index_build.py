"""The target search index (uniform grid, records ordered by (cell, original index), cell_start) in both device build forms —
the bucket sort (one global atomic per point) and the slab form (x-slab partition + one workgroup per slab in LDS) — against a
numpy restatement of the ordering, bit for bit.  Covers many targets per launch sequence (own-target batches, BASELINE configs[3]),
empty and tiny targets, a cloud that lives in ONE x-slab (the slab too big for LDS: global slot tables), and clouds with points
outside the grid's clamped range.  This replaces the two pcl::KdTreeFLANN::setInputCloud calls per registration
(/root/reference/src/node/odomEstimationNode.cpp:602-603)."""
import numpy as np
import pytest


def _expected(idx, cloud_xyz):
    """numpy restatement: cell of every point (float32 arithmetic as on the device), stable order by cell then index."""
    o, inv = idx["origin"], np.float32(1.0) / np.float32(idx["cell"])
    dims = (idx["nx"], idx["ny"], idx["nz"])
    c = []
    for a in range(3):
        v = np.floor((cloud_xyz[:, a].astype(np.float32) - o[a]) * inv).astype(np.int64)
        c.append(np.clip(v, 0, dims[a] - 1))
    cell = (c[0] * dims[1] + c[1]) * dims[2] + c[2]
    order = np.lexsort((np.arange(len(cell)), cell))
    cs = np.zeros(idx["n_cells"] + 1, np.int64)
    np.add.at(cs, cell + 1, 1)
    return order, np.cumsum(cs)


def _check(ctx, slot, kind, cloud):
    import lisreg
    idx = ctx.target_index(slot, kind)
    xyz = lisreg.synth.pcl_xyz(cloud)
    assert idx["n"] == len(cloud)
    if len(cloud) == 0:
        return
    order, cs = _expected(idx, xyz)
    assert np.array_equal(idx["cell_start"], cs), "cell_start"
    assert np.array_equal(idx["sorted"][:, 3].view(np.int32), order.astype(np.int32)), "record order (cell, original index)"
    assert np.array_equal(idx["sorted"][:, :3], xyz[order].astype(np.float32))


def _clouds(seed):
    from lisreg import synth
    rng = np.random.default_rng(seed)
    tc, ts = synth.make_submap(30000 + 1000 * seed, 300 + seed)
    wall = np.zeros((40000, 3), np.float32)                     # one x-slab, 40 k points: beyond the LDS capacity of a slab
    wall[:, 0] = 3.0 + rng.uniform(0, 0.2, len(wall)); wall[:, 1] = rng.uniform(-30, 30, len(wall)); wall[:, 2] = rng.uniform(-1, 6, len(wall))
    thick = np.concatenate([rng.uniform(-40, 40, (20000, 3)).astype(np.float32) * np.array([1, 1, 0.1], np.float32), wall[:25000]])
    return tc, ts, synth.to_pcl(wall, np.zeros(len(wall), np.uint16)), synth.to_pcl(thick, np.zeros(len(thick), np.uint16))


@pytest.mark.gpu
@pytest.mark.parametrize("form", [0, 1])
def test_batched_index_build_is_exactly_cell_then_index_order(gpu_ctx, form):
    import lisreg
    from lisreg import synth
    tc, ts, wall, thick = _clouds(1)
    tc2, ts2, _, _ = _clouds(2)
    empty = synth.to_pcl(np.zeros((0, 3), np.float32), np.zeros(0, np.uint16))
    tiny = synth.to_pcl(np.array([[0, 0, 0], [0.1, 0, 0], [5, 5, 1], [5, 5.01, 1], [-3, 2, 0]], np.float32), np.zeros(5, np.uint16))
    targets = [(tc, ts), (tc2, wall), (tiny, thick), (empty, ts2)]
    for s, (a, b) in enumerate(targets):
        gpu_ctx.set_target(a, b, slot=s)
    sc = synth.make_scan(16, 300, 77)
    items = [dict(src_corner=sc["corner"], src_surf=sc["surf"], target=s) for s in range(len(targets))]
    p = lisreg.default_params(1); p.fixed_iters = 2
    gpu_ctx.set_option("index_build", form); gpu_ctx.set_option("rebuild_targets_each_run", 1)
    try:
        gpu_ctx.align_batch(items, np.tile(sc["T_true"].astype(np.float32), (len(targets), 1)), p)
        assert gpu_ctx.get_option("index_build_now") == form
        for s, (a, b) in enumerate(targets):
            _check(gpu_ctx, s, 0, a)
            _check(gpu_ctx, s, 1, b)
    finally:
        gpu_ctx.set_option("index_build", 2); gpu_ctx.set_option("rebuild_targets_each_run", 0)


@pytest.mark.gpu
def test_single_target_build_matches_too(gpu_ctx):
    tc, ts, wall, thick = _clouds(3)
    gpu_ctx.set_target(tc, thick, slot=0)
    _check(gpu_ctx, 0, 0, tc)
    _check(gpu_ctx, 0, 1, thick)


@pytest.mark.gpu
def test_wide_sparse_map_still_takes_the_strip_form(gpu_ctx):
    """A 700 m x 700 m map: its grid has far more (ix, iy-run) strips than the partition histogram holds at the default strip
    length, so the strips are made longer; still the strip form, still exact."""
    import lisreg
    from lisreg import synth
    rng = np.random.default_rng(5)
    xyz = np.concatenate([rng.uniform(-350, 350, (120000, 2)), rng.uniform(-2, 8, (120000, 1))], 1).astype(np.float32)
    big = synth.to_pcl(xyz, np.zeros(len(xyz), np.uint16))
    tc, ts, _, _ = _clouds(4)
    gpu_ctx.set_target(tc, big, slot=0)
    sc = synth.make_scan(16, 300, 78)
    p = lisreg.default_params(1); p.fixed_iters = 1
    gpu_ctx.set_option("index_build", 1); gpu_ctx.set_option("rebuild_targets_each_run", 1)
    try:
        gpu_ctx.align_batch([dict(src_corner=sc["corner"], src_surf=sc["surf"], target=0)], sc["T_true"].astype(np.float32)[None], p)
        assert gpu_ctx.get_option("index_build_now") == 1
        idx = gpu_ctx.target_index(0, 1)
        assert idx["nx"] * idx["ny"] > 8192 * 16                  # the default 2048-cell strips would not have fitted
        _check(gpu_ctx, 0, 1, big)
        _check(gpu_ctx, 0, 0, tc)
    finally:
        gpu_ctx.set_option("index_build", 2); gpu_ctx.set_option("rebuild_targets_each_run", 0)


@pytest.mark.gpu
@pytest.mark.parametrize("m_points,kind", [(60000, 1), (8000, 1), (60000, 0)])
def test_graph_rows_cover_their_radius(gpu_ctx, m_points, kind):
    """The k-NN graph behind search front-end 3 (lisreg_get_target_graph): for every row, (1) the entries are other points, each once,
    in ascending distance from the row's point (to 2^-16 relative), (2) the coordinates stored in the row are the listed points' own, bit for bit (the
    scan never gathers them), (3) EVERY target point closer than rho is listed — the guarantee the 5-NN certificate rests on —
    checked against a float64 kd-tree, (4) padded entries carry the point's own coordinates and id -1."""
    from scipy.spatial import cKDTree
    from lisreg import synth
    tc, ts = synth.make_submap(m_points, 77)
    gpu_ctx.set_target(tc, ts)
    idx = gpu_ctx.target_index(0, kind); g = gpu_ctx.target_graph(0, kind)
    n, k = idx["n"], g["k"]
    pts32 = idx["sorted"][:, :3]; pts = pts32.astype(np.float64)
    assert g["ids"].shape == (n, k) and (g["count"] >= 0).all() and (g["count"] <= k).all()
    col = np.arange(k)[None, :]
    listed = col < g["count"][:, None]
    assert ((g["ids"] >= 0) == listed).all()                                   # a prefix of real entries, then padding
    own = np.broadcast_to(pts32[:, None, :], g["xyz"].shape)
    assert np.array_equal(g["xyz"][~listed], own[~listed])                     # (4)
    safe = np.where(listed, g["ids"], 0)
    assert np.array_equal(g["xyz"][listed], pts32[safe][listed])               # (2)
    assert (g["ids"][listed] != np.broadcast_to(np.arange(n)[:, None], (n, k))[listed]).all()
    d = np.linalg.norm(g["xyz"].astype(np.float64) - pts[:, None, :], axis=2)
    dd = np.where(listed, d, np.inf)
    with np.errstate(invalid="ignore"):                                        # inf / inf in the padding
        ratio = dd[:, 1:] / np.maximum(dd[:, :-1], 1e-12)
    # (1) ascending up to the key quantisation: since round 5 the graph's rows go through the 32-bit-key sort of the cell rows (the squared
    # distance's float bits with the low 7 bits replaced by the lane), exact to 2^-16 relative — inside the scan's millimetre of slack
    assert (ratio[listed[:, 1:]] >= 1 - 2.0 ** -15).all()
    # (a listed entry is inside rho up to the key quantisation: keys are compared with their low 7 mantissa bits cleared)
    assert (dd[listed] <= np.sqrt(g["rho2"].astype(np.float64))[:, None].repeat(k, 1)[listed] * (1 + 2.0 ** -15)).all()
    tree = cKDTree(pts)
    rng = np.random.default_rng(1)
    for s in rng.choice(n, min(n, 4000), replace=False):
        rho = float(np.sqrt(g["rho2"][s]))
        want = set(tree.query_ball_point(pts[s], rho * (1 - 1e-5))) - {int(s)}
        have = set(g["ids"][s][:g["count"][s]].tolist())
        assert len(have) == g["count"][s]                                      # each once
        assert want <= have, (s, rho, sorted(want - have)[:4])                 # (3)


@pytest.mark.gpu
@pytest.mark.parametrize("m_points,kind", [(60000, 1), (8000, 1), (60000, 0), (400000, 1)])
def test_cell_rows_cover_their_radius(gpu_ctx, m_points, kind):
    """The cell rows behind search front-end 5 (lisreg_get_target_cell_rows).  Table: -2 only where no target point lies in the 5 x 5 x 5 cell
    block, every other cell has a centre row and one row per octant of its mask, rows laid end to end.  For every row, about its centre
    (cell centre, or octant centre = cell corner + 0.25 / 0.75 of the edge): (1) the entries are target points, each once, ascending in
    distance to 2^-16 relative (the build sorts quantised keys), (2) the stored coordinates are the listed points' own, bit for bit,
    (3) EVERY target point closer than rho is listed — the guarantee the certificate rests on — against a float64 kd-tree, (4) padded
    entries carry the centre's coordinates and id -1, (5) an octant with a target point inside its box has a row."""
    from scipy.spatial import cKDTree
    from lisreg import synth
    tc, ts = synth.make_submap(m_points, 78)
    gpu_ctx.set_target(tc, ts)
    g = gpu_ctx.target_cell_rows(0, kind); idx = gpu_ctx.target_index(0, kind)      # (building the rows gives the grid its two-cell margin)
    n, k, R = idx["n"], g["k"], g["n_rows"]
    nx, ny, nz, cell, org = idx["nx"], idx["ny"], idx["nz"], np.float32(idx["cell"]), idx["origin"].astype(np.float32)
    pts32 = idx["sorted"][:, :3]; pts = pts32.astype(np.float64)
    tab = g["table"]
    assert tab.shape == (nx * ny * nz,) and ((tab >= 0) | (tab == -2)).all()      # (-1 only when the buffers are too small: not here)
    # occupancy -> which cells may be -2
    cs = idx["cell_start"]; occ = (np.diff(cs) > 0).reshape(nx, ny, nz)
    from scipy.ndimage import maximum_filter
    near = maximum_filter(occ.astype(np.uint8), size=5, mode="constant", cval=0).astype(bool).reshape(-1)
    assert np.array_equal(tab >= 0, near)
    have = np.flatnonzero(tab >= 0)
    base = tab[have] >> 8; mask = tab[have] & 255
    cnt = 1 + np.array([bin(int(m)).count("1") for m in mask])
    order = np.argsort(base)
    assert base[order][0] == 0 and np.array_equal(base[order][1:], np.cumsum(cnt[order])[:-1]) and base[order][-1] + cnt[order][-1] == R
    # (5): cells that hold a point have the octant that point is in
    pc = np.floor((pts32 - org) / cell).astype(np.int64)
    pc = np.minimum(np.maximum(pc, 0), np.array([nx - 1, ny - 1, nz - 1]))
    frac = (pts32 - org) / cell - pc
    inner = ((frac > 0.01) & (frac < 0.49)) | ((frac > 0.51) & (frac < 0.99))   # clear of the octant faces (float rounding)
    ok = inner.all(1)
    octv = (frac[:, 0] >= 0.5).astype(int) + 2 * (frac[:, 1] >= 0.5) + 4 * (frac[:, 2] >= 0.5)
    cid = (pc[:, 0] * ny + pc[:, 1]) * nz + pc[:, 2]
    assert ((tab[cid[ok]] >> octv[ok]) & 1).all()
    # rows -> centres
    centre = np.zeros((R, 3), np.float32)
    for c_, b_, m_ in zip(have, base, mask):
        iz = c_ % nz; iy = (c_ // nz) % ny; ix = c_ // (nz * ny)
        h = np.array([ix, iy, iz], np.float32)
        centre[b_] = (org.astype(np.float64) + (h.astype(np.float64) + 0.5) * np.float64(cell)).astype(np.float32)
        slot = 1
        for o in range(8):
            if (m_ >> o) & 1:
                f = np.array([0.75 if o & 1 else 0.25, 0.75 if o & 2 else 0.25, 0.75 if o & 4 else 0.25], np.float32)
                centre[b_ + slot] = (org.astype(np.float64) + (h.astype(np.float64) + f.astype(np.float64)) * np.float64(cell)).astype(np.float32)
                slot += 1
    col = np.arange(k)[None, :]
    listed = col < g["count"][:, None]
    assert (g["count"] >= 0).all() and (g["count"] <= k).all() and ((g["ids"] >= 0) == listed).all()
    # (4) (the device forms a centre with one fused multiply-add; float64 here, rounded once: the same value up to double rounding)
    assert np.abs(g["xyz"][~listed] - np.broadcast_to(centre[:, None, :], g["xyz"].shape)[~listed]).max(initial=0) <= 4e-6
    safe = np.where(listed, g["ids"], 0)
    assert np.array_equal(g["xyz"][listed], pts32[safe][listed])               # (2)
    d = np.linalg.norm(g["xyz"].astype(np.float64) - centre.astype(np.float64)[:, None, :], axis=2)
    dd = np.where(listed, d, np.inf)
    with np.errstate(invalid="ignore"):
        ratio = dd[:, 1:] / np.maximum(dd[:, :-1], 1e-12)
    assert (ratio[listed[:, 1:]] >= 1 - 2.0 ** -15).all()                      # (1) ascending up to the key quantisation
    tree = cKDTree(pts)
    rng = np.random.default_rng(2)
    for r in rng.choice(R, min(R, 4000), replace=False):
        rho = float(np.sqrt(g["rho2"][r]))
        want = set(tree.query_ball_point(centre[r].astype(np.float64), rho * (1 - 1e-5)))
        got = g["ids"][r][:g["count"][r]].tolist()
        assert len(set(got)) == len(got)                                       # each once
        assert want <= set(got), (r, rho, sorted(want - set(got))[:4])         # (3)
