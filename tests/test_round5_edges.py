"""Round-5 edge cases of the cell-row index (search front-end 5): buffers that are re-used across targets of different sizes and across
batch runs that rebuild their targets, now that the row build hands the octant masks back as zero instead of a memset in front of every
classification (lisreg_index.hip: launch_crow_classify / launch_crow_build, `omask_zero`)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rows(ctx, kind):
    g = ctx.target_cell_rows(0, kind)
    return g["table"].copy(), g["count"].copy(), g["rho2"].copy(), g["ids"].copy()


def test_cell_rows_of_a_reused_slot_equal_those_of_a_fresh_context(gpu_ctx):
    """One context, one slot, a sequence of targets of different sizes (bigger, smaller, bigger again: the mask buffer is re-used, re-allocated,
    re-used with a shorter and a longer grid), with registrations and whole batch runs (which rebuild the index and its rows inside the run)
    in between: after every step the slot's table, row counts, coverage radii and listed ids equal those a FRESH context builds for the same
    target — a mask that was not handed back as zero would OR stale octants into the next classification and change the table."""
    import lisreg
    from lisreg import synth
    p = lisreg.default_params(1); p.fixed_iters = 3
    c = lisreg.Context(0)
    c.set_option("search_mode", 5); c.set_option("canonical_ties", 1); c.set_option("lanes_per_query", 1)
    sizes = [(30000, 11), (80000, 12), (12000, 13), (80000, 14), (30000, 11)]
    scan = synth.make_scan(16, 300, 4321)
    T0 = synth.perturb_pose(scan["T_true"], np.random.default_rng(3))
    items = [dict(src_corner=scan["corner"], src_surf=scan["surf"], T_init=T0)] * 3
    for m, seed in sizes:
        tc, ts = synth.make_submap(m, seed)
        c.set_target(tc, ts)
        Tb, stb = c.align_batch(items, np.array([T0] * 3), p)          # prepare + run: index and rows rebuilt inside the run
        Tb2, stb2 = c.align_batch(items, np.array([T0] * 3), p)        # ... and again on the masks the first run handed back
        assert np.array_equal(Tb, Tb2) and stb == stb2
        f = lisreg.Context(0)
        f.set_option("search_mode", 5); f.set_option("canonical_ties", 1); f.set_option("lanes_per_query", 1)
        f.set_target(tc, ts)
        Tf, stf = f.align_batch(items, np.array([T0] * 3), p)
        assert np.array_equal(Tb, Tf) and stb == stf, (m, seed)
        for kind in (0, 1):
            a, b = _rows(c, kind), _rows(f, kind)
            for x, y in zip(a, b):
                assert np.array_equal(x, y), (m, seed, kind)
        f.close()
    c.close()
