"""N > 1 path on CPU: world_size-2 gloo run of the sharding + result all-gather used by bench.py on RCCL.
Each rank registers its shard with the CPU oracle (stand-in for its GPU), the gathered block must equal the
single-process result for all items, in item order."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n_items, out_path):
    for p in (os.path.join(ROOT, "lis-slam_amd"), os.path.join(ROOT, "oracle")):
        sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    import oracle_ctypes as oc
    from lisreg import shard, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard.shard_range(n_items, rank, world)
    tc, ts = synth.make_submap(6000, 42)
    p = oc.default_params(1); p.fixed_iters = 3
    Ts, sts = [], []
    for i in range(lo, hi):
        sc = synth.make_scan(8, 200, 3000 + i)
        T0 = synth.perturb_pose(sc["T_true"], np.random.default_rng(i))
        T, st, _ = oc.align(tc, ts, sc["corner"], sc["surf"], T0, p, max_trace=1)
        Ts.append(T); sts.append(st)
    local = torch.from_numpy(shard.pack_results(np.array(Ts).reshape(-1, 6), sts))
    allr = shard.gather_results(local, n_items, world)
    if rank == 0:
        np.save(out_path, allr.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions():
    from lisreg import shard
    for n in (0, 1, 5, 64, 2048, 2049):
        for w in (1, 2, 3, 8):
            r = [shard.shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.timeout(300)
def test_two_rank_gather_equals_single_process(tmp_path):
    import torch.multiprocessing as mp
    n_items = 5                      # ragged: 3 + 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    out2 = str(tmp_path / "w2.npy")
    mp.spawn(_worker, args=(2, port, n_items, out2), nprocs=2, join=True)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    out1 = str(tmp_path / "w1.npy")
    mp.spawn(_worker, args=(1, port, n_items, out1), nprocs=1, join=True)
    a, b = np.load(out2), np.load(out1)
    assert a.shape == (n_items, 12) and np.array_equal(a, b)
    assert np.all(a[:, 6] == 3) and np.all(np.isin(a[:, 11], (0, 2)))
