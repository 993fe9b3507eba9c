#!/usr/bin/env python3
"""bench.py — scan-to-submap registrations/sec on MI355X (BASELINE.json metric).

Workload (BASELINE.json configs[1], SURVEY.md §8d cfg 2): per GPU a batch of 64 synthetic 64x1800 scans (every
valid pixel a feature: ~4.4 k edge + ~110.8 k planar points) against one shared 200 k-point submap, semantic mask
off, fixed 10 Gauss-Newton iterations.  A "step" = one pass of the hot path over that batch with the inputs
already resident in HBM: target index build (the reference rebuilds both kd-trees per registration,
odomEstimationNode.cpp:602-603; here once per batch because the submap is shared), 10 x {correspondence +
normal-equation kernel, solve kernel}, finalize (sources stay in caller order: scan order is already coherent).  Multi-GPU: one process per GPU, independent
batches per rank (weak scaling), one RCCL all-gather of the 64 x 12-float result blocks per step.

Prints ONE JSON line (rank 0).  `roofline.achieved` = algorithmic bytes per launch of the dominant kernel
(96 B x point-iterations in the launch) / its average duration measured with HIP events on the
library's stream inside the timed region.  `cpu_baseline` = the CPU oracle (a port of the reference path,
kd-tree build included) timed on this host on a bounded sample of the same scans.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "lis-slam_amd"))

import numpy as np
import torch

import lisreg
from lisreg import synth, synth_torch

H, W, M_SUBMAP, BATCH, ITERS = 64, 1800, 200_000, 64, 10
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
ALG_BYTES_PER_POINT_ITER = 96  # 16 B source read + 5 x 16 B neighbour gather


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--cpu-regs", type=int, default=24, help="registrations in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-profile", action="store_true", help="do not record per-kernel HIP events")
    ap.add_argument("--workload", default="cfg2", choices=["cfg2", "cfg4", "cfg5"],
                    help="cfg2 (default, the metric's config): 64 scans 64x1800 vs one shared 200k submap, 10 iters; "
                         "cfg4: loop-closure style, every item has its OWN 200k target (index built per item per step); "
                         "cfg5: stress, 8 scans 128x2048 vs one shared 1M submap, 30 iters")
    args = ap.parse_args()

    global H, W, M_SUBMAP, ITERS
    if args.workload == "cfg5":
        H, W, M_SUBMAP, ITERS = 128, 2048, 1_000_000, 30
        if args.batch == BATCH:
            args.batch = 8
    own_targets = args.workload == "cfg4"
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # LISREG_BENCH_FORCE_DIST=1: take the multi-rank code path (process group, RCCL all-gather of the result blocks,
    # MAX-reduced timing) even with one rank — lets the path the driver runs at N = 2/4/8 be exercised on a 1-GPU box.
    use_dist = world > 1 or bool(os.environ.get("LISREG_BENCH_FORCE_DIST"))
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    n_gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    ctx = lisreg.Context(local_rank)
    stream = torch.cuda.Stream(device=dev)
    ctx.set_stream(stream.cuda_stream)
    ctx.set_option("rebuild_targets_each_run", 1)
    if os.environ.get("LISREG_COUNT"):
        ctx.set_option("count_searches", 1)

    # ---- synthetic inputs, generated straight into HBM ----------------------------------------------------------
    tc_dev, ts_dev, tc_host, ts_host = synth_torch.submap_device(M_SUBMAP, dev)
    scans, T_true, T_init = [], [], []
    for i in range(args.batch):
        seed = 1000 + rank * args.batch + i
        c, s, tt = synth_torch.make_scan_device(H, W, seed, dev)
        if os.environ.get("LISREG_BENCH_SHUFFLE"):      # robustness probe: destroy the scan order of the sources
            c = c[torch.randperm(c.shape[0], device=dev)].contiguous(); s = s[torch.randperm(s.shape[0], device=dev)].contiguous()
        scans.append((c, s)); T_true.append(tt)
        T_init.append(synth.perturb_pose(tt, np.random.default_rng(seed + 7919)))
    torch.cuda.synchronize()
    T_true = np.array(T_true, np.float32); T_init = np.array(T_init, np.float32)
    n_src = sum(c.shape[0] + s.shape[0] for c, s in scans)

    params = lisreg.default_params(lisreg.VARIANT_ODOM)
    params.fixed_iters = ITERS
    ctx.set_target_device(tc_dev.data_ptr(), tc_dev.shape[0], ts_dev.data_ptr(), ts_dev.shape[0])
    items = [dict(corner_ptr=c.data_ptr(), n_corner=c.shape[0], surf_ptr=s.data_ptr(), n_surf=s.shape[0]) for c, s in scans]
    own = []
    if own_targets:                      # every candidate pair gets its own copy of the submap in its own slot
        for i in range(args.batch):
            a, b = tc_dev.clone(), ts_dev.clone()
            own.append((a, b))
            ctx.set_target_device(a.data_ptr(), a.shape[0], b.data_ptr(), b.shape[0], slot=i)
            items[i]["target"] = i
        torch.cuda.synchronize()
    ctx.batch_prepare_device(items, T_init, params)

    gathered = torch.empty((world, args.batch, lisreg.RESULT_SIZE), dtype=torch.float32, device=dev)

    class _DevArray:       # zero-copy torch view of the library's device result block
        def __init__(self, ptr, shape):
            self.__cuda_array_interface__ = dict(shape=shape, typestr="<f4", data=(ptr, False), version=2)

    local_view = torch.as_tensor(_DevArray(ctx.result_device_ptr, (args.batch, lisreg.RESULT_SIZE)), device=dev)

    def step():
        ctx.batch_run()
        if use_dist:           # RCCL all-gather of the 64 x 12-float result blocks (poses + stats) over xGMI
            import torch.distributed as dist
            with torch.cuda.stream(stream):
                dist.all_gather_into_tensor(gathered.view(-1), local_view.view(-1))

    def barrier():
        if use_dist:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    if not args.no_profile:
        ctx.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if use_dist:
        import torch.distributed as dist
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    T_gpu, st_gpu = ctx.batch_fetch()          # also collects the event timings
    timing = ctx.timing() if not args.no_profile else None
    ctx.set_profiling(False)

    regs = n_gpus * args.batch * args.steps
    value = regs / elapsed
    ms_per_step = 1e3 * elapsed / args.steps

    # ---- roofline of the dominant kernel (k_assoc) ---------------------------------------------------------------
    roof = None
    if timing and timing["assoc_launches"] > 0:
        avg_ms = timing["assoc_ms"] / timing["assoc_launches"]
        alg_bytes = ALG_BYTES_PER_POINT_ITER * n_src                     # one launch = one GN iteration of the batch
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")           # PMC-derived HBM bytes/launch
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("k_assoc_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        roof = dict(bound="hbm", kernel="k_assoc_walk", achieved=round(achieved, 2), peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=round(achieved / HBM_PEAK_GBS, 5), traffic=traffic,
                    avg_launch_ms=round(avg_ms, 4), launches=timing["assoc_launches"],
                    algorithmic_bytes_per_launch=alg_bytes,
                    time_share=dict(assoc_ms=round(timing["assoc_ms"], 3), solve_ms=round(timing["solve_ms"], 3),
                                    index_ms=round(timing["index_ms"], 3), wall_ms=round(1e3 * elapsed, 3)))

    # ---- accuracy: vs ground truth for all items, vs the CPU oracle on the sampled items --------------------------
    err_truth = np.abs(T_gpu.astype(np.float64) - T_true.astype(np.float64))
    err_truth[:, :3] = np.abs((err_truth[:, :3] + np.pi) % (2 * np.pi) - np.pi)

    cpu = None
    parity = None
    if rank == 0 and n_gpus == 1 and args.cpu_regs > 0:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle_ctypes as oc                                       # checker + timed CPU baseline only
        oc.build()
        p_o = oc.default_params(1); p_o.fixed_iters = ITERS
        k = min(args.cpu_regs, args.batch)
        host_scans = [(synth_torch.records_to_pcl(scans[i][0]), synth_torch.records_to_pcl(scans[i][1])) for i in range(k)]
        tcpu0 = time.perf_counter()
        T_cpu = []
        for i in range(k):
            To, so, _ = oc.align(tc_host, ts_host, host_scans[i][0], host_scans[i][1], T_init[i], p_o, n_threads=1,
                                 use_kdtree=True, max_trace=1)
            T_cpu.append(To)
        tcpu = time.perf_counter() - tcpu0
        T_cpu = np.array(T_cpu)
        d = np.abs(T_gpu[:k].astype(np.float64) - T_cpu.astype(np.float64))
        parity = dict(items=k, max_rot_err_rad=float(d[:, :3].max()), max_trans_err_m=float(d[:, 3:].max()))
        cpu = dict(value=round(k / tcpu, 4), unit="registrations/s", cores=1, kind="port",
                   sample=f"{k} of the {args.batch} scans of this batch ({H}x{W} vs {M_SUBMAP // 1000}k submap, {ITERS} GN iters, "
                          f"kd-tree leaf 15, two tree builds per registration), 1 thread of {os.cpu_count()} host cores",
                   seconds=round(tcpu, 2))

    if os.environ.get("LISREG_COUNT"):
        cnt = ctx.counters()
        print("searched fraction per GN iteration:", [round(float(a) / max(float(b), 1), 4) for a, b in cnt[:ITERS]], file=sys.stderr)
    if rank == 0:
        out = {
            "metric": "scan-to-submap registrations/sec (64x1800 pts, 200k submap)",
            "value": round(value, 2), "unit": "registrations/s", "n_gpus": n_gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": {"cfg2": "configs[1]: batch=64 synthetic 64x1800 scans vs one shared 200k-pt submap per GPU, "
                                            "semantic mask off, 10 fixed GN iterations, target index build inside the step",
                                    "cfg4": "configs[3]-style: independent registrations, each against its OWN 200k-pt target "
                                            "(index built per item inside the step), 10 fixed GN iterations",
                                    "cfg5": "configs[4]: 128x2048 scans vs one shared 1M-pt submap, 30 fixed GN iterations"}[args.workload],
                       "batch_per_gpu": args.batch, "scan": [H, W], "submap_points": M_SUBMAP, "gn_iters": ITERS,
                       "source_points_per_batch": int(n_src), "parallelism": f"independent batches x{n_gpus} + RCCL all-gather of results"},
            "roofline": roof, "cpu_baseline": cpu,
            "accuracy": {"max_rot_err_vs_truth_rad": float(err_truth[:, :3].max()),
                         "max_trans_err_vs_truth_m": float(err_truth[:, 3:].max()),
                         "vs_cpu_oracle": parity,
                         "all_status_ok": bool(all(s["status"] == 0 for s in st_gpu))},
        }
    else:
        out = None
    if use_dist:
        import torch.distributed as dist
        # the gathered block of every rank must hold every rank's poses (rank r's slice == what rank r computed)
        g = gathered.cpu().numpy()
        assert np.array_equal(g[rank, :, :6], T_gpu), "all-gathered result block differs from the local results"
        dist.destroy_process_group()
    if out is not None:                     # the ONE JSON line, after every library has said what it had to say
        sys.stderr.flush()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)  # C-level buffers (RCCL prints its version banner through stdio)
        except Exception:
            pass
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
