"""the pose gather through the library's OWN RCCL entry points (lisreg_comm_unique_id / lisreg_comm_init /
lisreg_gather_results) with more than one rank — one process per GPU, the unique id handed over through a file, no torch anywhere.
Needs >= 2 visible devices; skipped on the 1-GPU test box (where tests/test_gpu_batch.py::test_native_rccl_gather_single_rank covers the
bootstrap and the all-gather call with one rank)."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

RANK = textwrap.dedent("""
    import os, sys, time
    import numpy as np
    sys.path.insert(0, os.path.join(sys.argv[1], "lis-slam_amd"))
    import lisreg
    from lisreg import synth
    rank, nranks, idfile = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    ctx = lisreg.Context(rank)                                   # one process per GPU
    if rank == 0:
        uid = lisreg.comm_unique_id()
        with open(idfile + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(idfile + ".tmp", idfile)
    else:
        t0 = time.time()
        while not os.path.exists(idfile):
            if time.time() - t0 > 60:
                raise SystemExit("no unique id")
            time.sleep(0.05)
        uid = open(idfile, "rb").read()
    ctx.comm_init(rank, nranks, uid)
    print(f"rank {rank}: device {rank}, comm_nranks {ctx.get_option('comm_nranks')}", flush=True)
    # every rank registers ITS share of the items (block partition) against the same submap
    tc, ts = synth.make_submap(20000, seed=42)
    n_local = 3
    cases = [synth.make_case(h=16, w=450, m_points=20000, scan_seed=1500 + rank * n_local + k) for k in range(n_local)]
    D = lisreg.DeviceArray
    recs = [(D(lisreg.pack_device_records(c["src_corner"])), D(lisreg.pack_device_records(c["src_surf"]))) for c in cases]
    ctx.set_target(tc, ts)
    p = lisreg.default_params(1); p.fixed_iters = 4
    items = [dict(corner_ptr=a.ptr, n_corner=a.shape[0], surf_ptr=b.ptr, n_surf=b.shape[0]) for a, b in recs]
    ctx.batch_prepare_device(items, np.array([c["T_init"] for c in cases]), p)
    ctx.batch_run()
    out = D(np.zeros((nranks * n_local, 12), np.float32))
    ctx.gather_results(ctx.result_device_ptr, n_local, out.ptr)
    T, st = ctx.batch_fetch()
    got = lisreg.device_to_host(out.ptr, (nranks * n_local, 12))
    assert np.array_equal(got[rank * n_local:(rank + 1) * n_local, :6], T)
    np.save(f"{idfile}.rank{rank}.npy", got)
    ctx.comm_destroy()
    ctx.close()
""")


def _device_count():
    sys.path.insert(0, os.path.join(ROOT, "lis-slam_amd"))
    import lisreg
    return lisreg.device_count()


@pytest.mark.gpu
def test_native_rccl_gather_two_ranks(tmp_path):
    n = _device_count()
    if n < 2:
        pytest.skip(f"{n} HIP device(s) visible: the two-rank gather needs two")
    idfile = str(tmp_path / "uid")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = [subprocess.Popen([sys.executable, "-c", RANK, ROOT, str(r), "2", idfile], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert all("comm_nranks 2" in o for o in outs), outs
    g0, g1 = np.load(f"{idfile}.rank0.npy"), np.load(f"{idfile}.rank1.npy")
    assert np.array_equal(g0, g1)                                 # every rank holds every rank's block
    assert np.all(g0[:, 6] == 4)                                  # 4 fixed iterations everywhere


def test_rank_script_is_importable_text():
    compile(RANK, "<rank>", "exec")                               # CPU suite: the rank program at least parses
