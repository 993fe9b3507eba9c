#!/usr/bin/env python3
"""bench.py — scan-to-submap registrations/sec on MI355X (BASELINE.json metric).

Default workload (BASELINE.json configs[1]): per GPU a batch of 64 synthetic 64x1800 scans (every
valid pixel a feature: ~4.4 k edge + ~110.8 k planar points) against one shared 200 k-point submap, semantic mask
off, fixed 10 Gauss-Newton iterations.  A "step" = one pass of the hot path over that batch with the inputs
already resident in HBM: target index build (the reference rebuilds both kd-trees per registration,
odomEstimationNode.cpp:602-603; here once per batch because the submap is shared), 10 x {correspondence +
normal-equation kernel, solve kernel}, finalize.

  python bench.py --gpus N --steps K --warmup W [--workload cfg1|cfg2|cfg3|cfg4|cfg5|odom]

Multi-GPU: one process per GPU, independent batches per rank (weak scaling), one RCCL all-gather of the result
blocks per step.  `--gpus N` with N > 1 and no WORLD_SIZE in the environment re-launches this script under
`python -m torch.distributed.run --nproc-per-node N` (rendezvous on 127.0.0.1); under a launcher, WORLD_SIZE must
equal N and N devices must be visible, otherwise the script exits non-zero — it never reports a 1-GPU figure as N.

Prints ONE JSON line (rank 0).  `value` = registrations of all ranks / time of K steps (device-resident inputs).
If K steps take less than --min-seconds (default 2 s) the K-step loop is repeated R times back to back inside the
same timed bracket so that samplers see a busy GPU; `ms_per_step` is then the mean over R x K steps and
`timed_region_s` / `step_loop_repeats` say so.
`roofline.achieved` = algorithmic bytes per launch of the dominant kernel (96 B x point-iterations in the launch,
BASELINE.md §3) / its average duration measured with HIP events on the library's stream inside the timed region.
`cpu_baseline` = the CPU oracle (a port of the reference path, kd-tree build included) timed on this host on a
bounded sample of the same scans at 1 thread (headline), 2 threads (the reference's numberOfCores,
config/params.yaml:127) and all host cores.  `pcie_inclusive` = the same batch handed over as pinned HOST clouds in
the reference's 32-byte PCL layout through lisreg_align_batch (H2D of the sources, device-side packing, the
registration, D2H of poses and stats all inside the timed loop) — reported next to `value`, never as `value`.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "lis-slam_amd"))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
ALG_BYTES_PER_POINT_ITER = 96  # 16 B source read + 5 x 16 B neighbour gather

WORKLOADS = {
    # name: (H, W, submap points, batch per GPU, GN iterations, own target per item, description)
    "cfg1": (64, 1800, 50_000, 1, 10, False,
             "configs[0] synthetic stand-in: ONE 64x1800 scan vs a 50k-pt submap, 10 fixed GN iterations (latency-shaped)"),
    "cfg2": (64, 1800, 200_000, 64, 10, False,
             "configs[1]: batch=64 synthetic 64x1800 scans vs one shared 200k-pt submap per GPU, semantic mask off, "
             "10 fixed GN iterations, target index build inside the step"),
    "cfg3": (64, 1800, 0, 1, 0, False,
             "configs[2] synthetic stand-in: sequential replay of a synthetic drive (previous pose as the guess, semantic "
             "split -> per-class voxel grid -> label-weighted registration against the sliding local map, early exit)"),
    "odom": (64, 1800, 0, 1, 0, False,
             "configs[0] as a sequence: scan-to-map odometry on raw unlabelled 64x1800 sweeps of a synthetic drive (range image + "
             "LOAM features -> voxel grids -> copy #1 registration against the <= 19 newest keyframes, early exit)"),
    "cfg4": (64, 1800, 200_000, 256, 10, True,
             "configs[3] shape on one GPU: 256 independent registrations, each against its OWN 200k-pt target "
             "(index built per item inside the step), 10 fixed GN iterations"),
    "cfg5": (128, 2048, 1_000_000, 8, 30, False,
             "configs[4]: 128x2048 scans vs one shared 1M-pt submap, 30 fixed GN iterations"),
    "cfg4_icp": (64, 1800, 200_000, 256, 30, True,
                 "configs[3] with the reference's own verification step: 256 loop-closure candidates per GPU through pcl::IterativeClosestPoint's "
                 "restatement (lisreg_icp_align_batch: each candidate its OWN 200k-pt target whose k = 1 index is built inside the step, "
                 "max correspondence distance 10 m, <= 30 iterations, per-candidate early exit, fitness score), poses gathered over RCCL"),
}


def relaunch_under_torchrun(n, argv):
    """--gpus N without a launcher: start N ranks ourselves (one per GPU) and hand back their exit code."""
    port = 29500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=0, help="registrations per GPU per step (0 = the workload's own)")
    ap.add_argument("--cpu-regs", type=int, default=12, help="registrations per CPU-baseline leg (0 = skip the CPU baseline)")
    ap.add_argument("--no-profile", action="store_true", help="do not record per-kernel HIP events")
    ap.add_argument("--no-pcie", action="store_true", help="skip the PCIe-inclusive leg")
    ap.add_argument("--min-seconds", type=float, default=2.0, help="repeat the K-step loop until the timed region is this long")
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(relaunch_under_torchrun(args.gpus, sys.argv[1:]))

    import numpy as np
    import torch

    import lisreg
    from lisreg import synth, synth_torch

    # The contract is ONE JSON line on stdout.  RCCL writes a five-line version banner to file descriptor 1 when its first communicator
    # comes up (N > 1 ranks), so everything below runs with fd 1 pointing at stderr; emit_json() writes the line to the real stdout.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit_json(obj):
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(obj) + "\n").encode())

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; refusing to "
                         f"report a {world}-rank figure as {args.gpus} GPUs")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    n_dev = torch.cuda.device_count()
    oversub = bool(os.environ.get("LISREG_BENCH_OVERSUBSCRIBE"))      # CI only: several ranks share one GPU
    if n_dev < args.gpus and not oversub:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but only {n_dev} HIP device(s) visible")
    dev_index = local_rank % n_dev
    # LISREG_BENCH_FORCE_DIST=1: take the multi-rank code path (process group, RCCL all-gather of the result blocks,
    # MAX-reduced timing) even with one rank.
    use_dist = world > 1 or bool(os.environ.get("LISREG_BENCH_FORCE_DIST"))
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # The pose gather goes through the LIBRARY's own RCCL entry points (lisreg_comm_init / lisreg_gather_results, what a C++ node would
    # call); torch.distributed is the launcher's control plane only (gloo: unique-id broadcast, barrier, MAX of the times).
    # LISREG_BENCH_GATHER=torch takes torch.distributed's nccl backend for the gather instead (the path of rounds 1-3).
    native_gather = use_dist and os.environ.get("LISREG_BENCH_GATHER", "native") != "torch" and not (oversub and n_dev < world)
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # several ranks on ONE device (CI) cannot form an RCCL communicator: gloo carries the gather there
        backend = "gloo" if (native_gather or (oversub and n_dev < world)) else "nccl"
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        assert dist.get_world_size() == args.gpus
    n_gpus = world
    # self-check for the scaling record: what the process group says, and the device every rank is bound to
    rank_devices = None
    if use_dist:
        props = torch.cuda.get_device_properties(dev_index)
        mine = dict(rank=rank, local_rank=local_rank, device_index=dev_index, name=props.name,
                    pci_bus_id=getattr(props, "pci_bus_id", None), uuid=str(getattr(props, "uuid", "")))
        gathered_dev = [None] * world
        dist.all_gather_object(gathered_dev, mine)
        rank_devices = gathered_dev
        print(f"[bench rank {rank}/{dist.get_world_size()}] control plane {dist.get_backend()}, pose gather "
              f"{'lisreg_comm (native RCCL)' if native_gather else dist.get_backend()}, device cuda:{dev_index} ({props.name})", file=sys.stderr)

    def native_comm(ctx_):
        """lisreg_comm_init on every rank with rank 0's unique id (shipped over the control plane); returns the communicator size"""
        uid = [lisreg.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        ctx_.comm_init(rank, world, uid[0])
        n_ = ctx_.get_option("comm_nranks")
        print(f"[bench rank {rank}] lisreg_comm_init ok: comm_nranks {n_}, device cuda:{dev_index}", file=sys.stderr)
        return n_

    if args.workload == "cfg4_icp":
        out = bench_icp(args, lisreg, torch, np, synth, synth_torch, dev, dev_index, rank, world, use_dist, native_gather,
                        dist if use_dist else None, native_comm, rank_devices)
        if use_dist:
            dist.barrier(); dist.destroy_process_group()
        if out is not None:
            emit_json(out)
        return

    if args.workload in ("cfg3", "odom"):
        from lisreg import replay
        fn = replay.bench_sequence if args.workload == "cfg3" else replay.bench_odometry
        out = fn(dev_index, steps=args.steps, warmup=args.warmup) if rank == 0 else None
        if out is not None:
            frames, recs = out.pop("_frames"), out.pop("_recs")
            out["cpu_baseline"] = chain_cpu_baseline(args.workload, frames, recs, np) if args.cpu_regs > 0 else None
            out["pcl_reference"] = pcl_reference_note()
        if use_dist:
            dist.barrier(); dist.destroy_process_group()
        if out is not None:
            out.update(n_gpus=1, scaling="replicas only (a sequential drive does not shard)")
            emit_json(out)
        return

    H, W, M_SUBMAP, BATCH, ITERS, own_targets, wl_desc = WORKLOADS[args.workload]
    batch = args.batch if args.batch > 0 else BATCH

    ctx = lisreg.Context(dev_index)
    stream = torch.cuda.Stream(device=dev)
    ctx.set_stream(stream.cuda_stream)
    ctx.set_option("rebuild_targets_each_run", 1)
    for kv in filter(None, os.environ.get("LISREG_OPTS", "").split(",")):        # tuning experiments: LISREG_OPTS=name=value,...
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    if os.environ.get("LISREG_COUNT"):
        ctx.set_option("count_searches", 1)
        if os.environ.get("LISREG_COUNT") == "2":
            ctx.set_option("dump_neighbors", 1)

    # ---- synthetic inputs, generated straight into HBM ----------------------------------------------------------
    tc_dev, ts_dev, tc_host, ts_host = synth_torch.submap_device(M_SUBMAP, dev)
    scans, T_true, T_init = [], [], []
    n_distinct = min(batch, 64)                     # scans are reused beyond 64 (own-target batches differ in their targets)
    for i in range(n_distinct):
        seed = 1000 + rank * 64 + i
        c, s, tt = synth_torch.make_scan_device(H, W, seed, dev)
        if os.environ.get("LISREG_BENCH_SHUFFLE"):      # robustness probe: destroy the scan order of the sources
            c = c[torch.randperm(c.shape[0], device=dev)].contiguous(); s = s[torch.randperm(s.shape[0], device=dev)].contiguous()
        scans.append((c, s)); T_true.append(tt)
        if os.environ.get("LISREG_BENCH_PERTURB"):      # probe: "trans_m,rot_deg" of the initial guess (default 0.3 m, 2 deg)
            tr_, rd_ = (float(v) for v in os.environ["LISREG_BENCH_PERTURB"].split(","))
            T_init.append(synth.perturb_pose(tt, np.random.default_rng(seed + 7919), trans=tr_, rot_deg=rd_))
        else:
            T_init.append(synth.perturb_pose(tt, np.random.default_rng(seed + 7919)))
    for i in range(n_distinct, batch):
        scans.append(scans[i % n_distinct]); T_true.append(T_true[i % n_distinct]); T_init.append(T_init[i % n_distinct])
    torch.cuda.synchronize()
    T_true = np.array(T_true, np.float32); T_init = np.array(T_init, np.float32)
    n_src = sum(c.shape[0] + s.shape[0] for c, s in scans)

    params = lisreg.default_params(lisreg.VARIANT_ODOM)
    params.fixed_iters = ITERS
    ctx.set_target_device(tc_dev.data_ptr(), tc_dev.shape[0], ts_dev.data_ptr(), ts_dev.shape[0])
    items = [dict(corner_ptr=c.data_ptr(), n_corner=c.shape[0], surf_ptr=s.data_ptr(), n_surf=s.shape[0]) for c, s in scans]
    own, own_host = [], {}
    if own_targets:                      # every candidate pair gets its own submap (distinct seeds) in its own slot
        from lisreg import pack_device_records
        for i in range(batch):
            tci, tsi = synth.make_submap(M_SUBMAP, 42 + i + 1000 * rank)
            if i < 8:
                own_host[i] = (tci, tsi)
            a = torch.from_numpy(pack_device_records(tci)).to(dev); b = torch.from_numpy(pack_device_records(tsi)).to(dev)
            own.append((a, b))
            ctx.set_target_device(a.data_ptr(), a.shape[0], b.data_ptr(), b.shape[0], slot=i)
            items[i]["target"] = i
        torch.cuda.synchronize()
    ctx.batch_prepare_device(items, T_init, params)

    gathered = torch.empty((world, batch, lisreg.RESULT_SIZE), dtype=torch.float32, device=dev)
    comm_nranks = native_comm(ctx) if native_gather else None

    class _DevArray:       # zero-copy torch view of the library's device result block
        def __init__(self, ptr, shape):
            self.__cuda_array_interface__ = dict(shape=shape, typestr="<f4", data=(ptr, False), version=2)

    local_view = torch.as_tensor(_DevArray(ctx.result_device_ptr, (batch, lisreg.RESULT_SIZE)), device=dev)
    gloo_gather = use_dist and dist.get_backend() == "gloo" and not native_gather

    def step():
        ctx.batch_run()
        if native_gather:                  # ncclAllGather of the result blocks (poses + stats) on the context's stream, issued by the library
            ctx.gather_results(ctx.result_device_ptr, batch, gathered.data_ptr())
        elif use_dist and not gloo_gather: # RCCL all-gather through torch.distributed
            with torch.cuda.stream(stream):
                dist.all_gather_into_tensor(gathered.view(-1), local_view.view(-1))

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    # how many times the K-step loop must run to fill --min-seconds (decided from a short untimed probe; all ranks agree)
    t_probe = time.perf_counter()
    step(); barrier()
    per_step = max(time.perf_counter() - t_probe, 1e-5)
    repeats = max(1, int(np.ceil(args.min_seconds / (per_step * args.steps)))) if args.min_seconds > 0 else 1
    ctl_dev = "cpu" if (use_dist and dist.get_backend() == "gloo") else dev
    if use_dist:
        rt = torch.tensor([repeats], dtype=torch.int64, device=ctl_dev)
        dist.all_reduce(rt, op=dist.ReduceOp.MAX)
        repeats = int(rt.item())
    prof_repeats = min(repeats, max(1, 400 // max(args.steps, 1)))     # HIP events only in the first loops (bounded event count)
    barrier()
    t0 = time.perf_counter()
    for r in range(repeats):
        if not args.no_profile:
            if r == 0:
                ctx.set_profiling(True)
            elif r == prof_repeats:
                ctx.set_profiling_paused(True)
        for _ in range(args.steps):
            step()
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=ctl_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    T_gpu, st_gpu = ctx.batch_fetch()          # also collects the event timings
    timing = ctx.timing() if not args.no_profile else None
    ctx.set_profiling(False)
    if gloo_gather:                            # CI path: same gather, over gloo, outside the timed loop
        parts = [torch.empty((batch, lisreg.RESULT_SIZE)) for _ in range(world)]
        dist.all_gather(parts, local_view.cpu())
        gathered = torch.stack(parts)

    total_steps = repeats * args.steps
    regs = n_gpus * batch * total_steps
    value = regs / elapsed
    ms_per_step = 1e3 * elapsed / total_steps

    # ---- roofline of the dominant kernel (k_assoc_walk) -----------------------------------------------------------
    roof = None
    if timing and timing["assoc_launches"] > 0:
        avg_ms = timing["assoc_ms"] / timing["assoc_launches"]
        prof_steps_ = min(repeats, prof_repeats) * args.steps
        # one launch = one GN iteration of the batch — or of HALF the batch when the run is interleaved (option "interleave": the two halves
        # iterate on two streams, each half's solves underneath the other half's correspondence launch): bytes per launch follow the launch count
        alg_bytes = int(ALG_BYTES_PER_POINT_ITER * n_src * (prof_steps_ * ITERS) / timing["assoc_launches"])
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")           # PMC-derived HBM bytes/launch
        if os.path.exists(tpath) and args.workload == "cfg2":
            try:
                traffic = json.load(open(tpath)).get("k_assoc_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        prof_steps = min(repeats, prof_repeats) * args.steps
        # BASELINE.md section 3 / SURVEY.md section 8(d): the WHOLE step against the roofline — A_reg = 96 I N_s + 40 M / B_t bytes per
        # registration (B_t = registrations sharing a target) x registrations/s of ONE GPU / 8 TB/s.  Everything of a step counts against it
        # (index build, solves, launch gaps), so it is below the dominant kernel's own fraction `frac`.
        n_tgt_pts = ctx.get_option("index_target_points")
        a_reg = (ALG_BYTES_PER_POINT_ITER * ITERS * n_src + 40.0 * n_tgt_pts) / batch
        step_achieved = a_reg * (batch / (ms_per_step * 1e-3)) / 1e9
        fe_kib, grid_kib = ctx.get_option("index_kib_front_end"), ctx.get_option("index_kib_grid")
        fe_built_kib = ctx.get_option("index_kib_front_end_built")          # cell rows the last run really built (row_reach): <= the allocation above
        roof = dict(bound="hbm", kernel="k_assoc_walk", achieved=round(achieved, 2), peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=round(achieved / HBM_PEAK_GBS, 5), traffic=traffic,
                    step_frac=round(step_achieved / HBM_PEAK_GBS, 5), step_achieved=round(step_achieved, 2),
                    algorithmic_bytes_per_registration=int(a_reg),
                    step_frac_note="A_reg x registrations/s per GPU / 8 TB/s (BASELINE.md section 3): the whole step incl. index build and solves",
                    index_bytes_per_target_point=round(1024.0 * (fe_kib + grid_kib) / max(n_tgt_pts, 1), 1),
                    index_bytes_per_target_point_parts=dict(grid=round(1024.0 * grid_kib / max(n_tgt_pts, 1), 1),
                                                            front_end=round(1024.0 * fe_kib / max(n_tgt_pts, 1), 1),
                                                            front_end_built_by_the_last_run=round(1024.0 * fe_built_kib / max(n_tgt_pts, 1), 1)),
                    avg_launch_ms=round(avg_ms, 4), launches=timing["assoc_launches"],
                    algorithmic_bytes_per_launch=alg_bytes, search_front_end=ctx.front_end(),
                    launches_per_iteration=round(timing["assoc_launches"] / (prof_steps_ * ITERS), 3), interleaved=bool(ctx.get_option("interleaved_now")),
                    row_reach=dict(on=bool(ctx.get_option("row_reach_now")), query_iterations_in_cells_without_rows=ctx.get_option("row_reach_misses"),
                                   note="cell rows built only for the cells the batch's queries come within a metre of under their initial poses (marks made by lisreg_batch_prepare); a query elsewhere takes the cell walk: same results"),
                    per_step_ms=dict(assoc=round(timing["assoc_ms"] / prof_steps, 4), solve=round(timing["solve_ms"] / prof_steps, 4),
                                     index=round(timing["index_ms"] / prof_steps, 4), wall=round(ms_per_step, 4)))

    # ---- accuracy: vs ground truth for all items, vs the CPU oracle on the sampled items --------------------------
    err_truth = np.abs(T_gpu.astype(np.float64) - T_true.astype(np.float64))
    err_truth[:, :3] = np.abs((err_truth[:, :3] + np.pi) % (2 * np.pi) - np.pi)

    cpu = None
    parity = None
    # (rank 0 only; at N > 1 the other ranks wait at the barrier below — the leg is ~10-25 s of host work, after the timed region)
    if rank == 0 and args.cpu_regs > 0:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle_ctypes as oc                                       # checker + timed CPU baseline only
        oc.build()
        p_o = oc.default_params(1); p_o.fixed_iters = ITERS
        k = min(args.cpu_regs, batch)
        if args.workload == "cfg5":
            k = min(k, 3)
        host_scans = [(synth_torch.records_to_pcl(scans[i][0]), synth_torch.records_to_pcl(scans[i][1])) for i in range(k)]
        ncores = host_cores()
        legs = {}
        T_cpu = None
        for threads in sorted({1, 2, ncores}):
            tcpu0 = time.perf_counter()
            Ts = []
            for i in range(k if threads <= 2 else min(k, 6)):
                tci, tsi = own_host.get(i, (tc_host, ts_host)) if own_targets else (tc_host, ts_host)
                if own_targets and i not in own_host:
                    break
                To, so, _ = oc.align(tci, tsi, host_scans[i][0], host_scans[i][1], T_init[i], p_o, n_threads=threads,
                                     use_kdtree=True, max_trace=1)
                Ts.append(To)
            tcpu = time.perf_counter() - tcpu0
            legs[threads] = dict(value=round(len(Ts) / tcpu, 4), seconds=round(tcpu, 2), registrations=len(Ts))
            if threads == 1:
                T_cpu = np.array(Ts)
        kk = len(T_cpu)
        d = np.abs(T_gpu[:kk].astype(np.float64) - T_cpu.astype(np.float64))
        parity = dict(items=kk, max_rot_err_rad=float(d[:, :3].max()), max_trans_err_m=float(d[:, 3:].max()))
        if not own_targets and n_gpus == 1:
            # the same registrations through the exact-arithmetic build (lisreg_assoc.hip compiled with the reference's arithmetic): its
            # poses are expected to BE the oracle's (tests/test_exact.py); reported, not timed
            cx = lisreg.Context(dev_index)
            cx.set_option("exact_arithmetic", 1)
            cx.set_target(tc_host, ts_host)
            Tx, sx = cx.align_batch([dict(src_corner=host_scans[i][0], src_surf=host_scans[i][1]) for i in range(kk)],
                                    np.ascontiguousarray(T_init[:kk], np.float32), params)
            cx.close()
            parity["exact_build_poses_bit_identical"] = int(sum(np.array_equal(np.asarray(Tx[i], np.float32), np.asarray(T_cpu[i], np.float32))
                                                                for i in range(kk)))
            parity["exact_build_max_diff"] = float(np.abs(np.asarray(Tx, np.float64) - T_cpu.astype(np.float64)).max())
        cpu = dict(value=legs[1]["value"], unit="registrations/s", cores=1, kind="port", pcl_reference=pcl_reference_note(),
                   sample=f"{kk} of the {batch} registrations of this batch ({H}x{W} vs {M_SUBMAP // 1000}k submap, {ITERS} GN iters, "
                          f"kd-tree leaf 15, two tree builds per registration) per leg; OpenMP over the feature points like the "
                          f"reference's numberOfCores loops, kd-tree build single-threaded as in PCL; {ncores} host cores usable "
                          f"(os.cpu_count() = {os.cpu_count()})",
                   seconds=legs[1]["seconds"],
                   by_threads={str(t): v for t, v in legs.items()})

    # ---- the exact-arithmetic build on the SAME device-resident batch (the parity anchor of tests/test_exact.py), timed ----------
    exact_leg = None
    if rank == 0 and n_gpus == 1 and not own_targets and not os.environ.get("LISREG_BENCH_NO_EXACT"):
        cx = lisreg.Context(dev_index)
        cx.set_stream(stream.cuda_stream)
        cx.set_option("rebuild_targets_each_run", 1)
        cx.set_option("exact_arithmetic", 1)
        cx.set_target_device(tc_dev.data_ptr(), tc_dev.shape[0], ts_dev.data_ptr(), ts_dev.shape[0])
        cx.batch_prepare_device(items, T_init, params)
        cx.batch_run(); torch.cuda.synchronize()
        xs = max(3, min(args.steps, 10))
        tx0 = time.perf_counter()
        for _ in range(xs):
            cx.batch_run()
        torch.cuda.synchronize()
        dtx = time.perf_counter() - tx0
        Tx_all, _ = cx.batch_fetch()
        cx.close()
        dx = np.abs(np.asarray(Tx_all, np.float64) - T_gpu.astype(np.float64))
        exact_leg = dict(value=round(batch * xs / dtx, 2), unit="registrations/s", ms_per_step=round(1e3 * dtx / xs, 3), steps=xs,
                         max_pose_diff_vs_production=float(dx.max()),
                         note="option exact_arithmetic = 1 (the reference's arithmetic operation for operation, canonical ties, the pose's "
                              "sines / cosines from the host's libm: one host round trip per GN iteration) on the same device-resident batch")

    # ---- two independent batches in flight: two contexts, two HIP streams, the same device-resident inputs ---------------------------
    # `value` above is ONE stream running step after step.  A step has stretches that leave most of the chip idle (the 6x6 solves: one
    # workgroup per registration; launch gaps) or half busy (the index / row build: latency- and LDS-bound); a second, independent batch on
    # another stream fills them.  Reported next to `value`, never as `value`; poses must equal the single-stream ones.
    overlap_leg = None
    if rank == 0 and n_gpus == 1 and not own_targets and not os.environ.get("LISREG_BENCH_NO_OVERLAP"):
        lanes = []
        for _ in range(2):
            cx = lisreg.Context(dev_index)
            sx = torch.cuda.Stream(device=dev)
            cx.set_stream(sx.cuda_stream)
            cx.set_option("rebuild_targets_each_run", 1)
            cx.set_target_device(tc_dev.data_ptr(), tc_dev.shape[0], ts_dev.data_ptr(), ts_dev.shape[0])
            cx.batch_prepare_device(items, T_init, params)
            lanes.append((cx, sx))
        for cx, _ in lanes:
            cx.batch_run()
        torch.cuda.synchronize()
        osteps = 2 * max(5, min(args.steps, 20))
        to0 = time.perf_counter()
        for k in range(osteps):
            lanes[k & 1][0].batch_run()
        torch.cuda.synchronize()
        dto = time.perf_counter() - to0
        same = True
        for cx, _ in lanes:
            To, _ = cx.batch_fetch()
            same = same and bool(np.array_equal(np.asarray(To), T_gpu))
            cx.close()
        overlap_leg = dict(value=round(batch * osteps / dto, 2), unit="registrations/s", ms_per_step=round(1e3 * dto / osteps, 3), steps=osteps,
                           poses_equal_single_stream_run=same,
                           note="two contexts on two HIP streams, each running the same device-resident batch step after step (index build inside every "
                                "step); the streams' kernels share the chip, so per-kernel durations of this leg are not comparable with `roofline`")

    # The same overlap inside ONE context and ONE batch: option "interleave" = 2 cuts a fixed-iteration run in two at a registration boundary
    # and iterates the halves on two streams, each half's 6x6 solves and launch tails underneath the other half's correspondence launch
    # (bit-identical results).  Off by default — a half-launch's duration is then no longer its own, and `roofline` is made of launch
    # durations — and reported here, next to `value`, with the whole-step fraction it reaches.
    interleave_leg = None
    if rank == 0 and n_gpus == 1 and not own_targets and not os.environ.get("LISREG_BENCH_NO_OVERLAP"):
        cx = lisreg.Context(dev_index)
        sx = torch.cuda.Stream(device=dev)
        cx.set_stream(sx.cuda_stream)
        cx.set_option("rebuild_targets_each_run", 1)
        cx.set_option("interleave", 2)
        cx.set_target_device(tc_dev.data_ptr(), tc_dev.shape[0], ts_dev.data_ptr(), ts_dev.shape[0])
        cx.batch_prepare_device(items, T_init, params)
        for _ in range(3):
            cx.batch_run()
        torch.cuda.synchronize()
        isteps = 2 * max(5, min(args.steps, 20))
        ti0 = time.perf_counter()
        for _ in range(isteps):
            cx.batch_run()
        torch.cuda.synchronize()
        dti = time.perf_counter() - ti0
        engaged = bool(cx.get_option("interleaved_now"))
        Ti, _ = cx.batch_fetch()
        cx.close()
        iv = batch * isteps / dti
        interleave_leg = dict(value=round(iv, 2), unit="registrations/s", ms_per_step=round(1e3 * dti / isteps, 3), steps=isteps, engaged=engaged,
                              poses_equal_single_stream_run=bool(np.array_equal(np.asarray(Ti), T_gpu)),
                              step_frac=round(float(roof["algorithmic_bytes_per_registration"]) * iv / (HBM_PEAK_GBS * 1e9), 5) if roof else None,
                              note="option interleave = 2 (two halves of the batch on two streams, free-running); ONE context, the same batch and index "
                                   "build per step as `value`; not the default because per-launch durations stop being a launch's own")

    # ---- PCIe-inclusive leg: pinned host clouds in the reference's 32-byte layout through lisreg_align_batch ------
    # At N > 1 EVERY rank runs it at the same time (between two barriers): the ranks' feeder threads and uploads then compete for the host's
    # cores, memory and PCIe root ports — the contention SURVEY.md section 8(e) names as the multi-GPU limit — and rank 0 reports the
    # whole-job rate (sum over ranks / the slowest rank's time) next to every rank's own figure.
    pcie = None
    if not args.no_pcie and not own_targets:
        if use_dist:
            dist.barrier()
        pcie = pcie_inclusive_leg(lisreg, torch, np, dev_index, stream, scans, tc_dev, ts_dev, T_init, params, T_gpu,
                                  steps=max(3, min(args.steps, 20)), feeder_threads=max(2, min(8, host_cores() // max(n_gpus, 1))) if n_gpus > 1 else 0)
        if use_dist:
            allp = [None] * world
            dist.all_gather_object(allp, dict(rank=rank, value=pcie["value"], ms_per_step=pcie["ms_per_step"], stage_ms=pcie["stage_ms"],
                                              feeder=pcie["feeder"], in_series=pcie["in_series"]["value"]))
            if rank == 0:
                slowest = max(p_["ms_per_step"] for p_ in allp)
                pcie["all_ranks_concurrently"] = dict(
                    value=round(n_gpus * batch / (slowest * 1e-3), 2), unit="registrations/s",
                    note="every rank staged, ran and fetched its own batch at the same time; value = registrations of all ranks per step / the slowest rank's step time",
                    per_rank=allp)
            else:
                pcie = None

    if os.environ.get("LISREG_COUNT"):
        cnt = ctx.counters()
        print("searched fraction per GN iteration:", [round(float(a) / max(float(b), 1), 4) for a, b in cnt[:ITERS]], file=sys.stderr)
        print("wavefronts with a walking lane per GN iteration:", [round(float(a) / max(float(b), 1), 4) for a, b in ctx.wave_counters()[:ITERS]], file=sys.stderr)
        if os.environ.get("LISREG_COUNT") == "3":       # experiment builds put their own tallies here
            raw = ctx.raw_counters()
            print("raw counters 32..63:", raw[32:64], file=sys.stderr)
            print("raw counters 96..127:", raw[96:128], file=sys.stderr)
        if os.environ.get("LISREG_COUNT") == "2":
            raw = ctx.raw_counters()[96:96 + ITERS]
            print("queries with an unchanged ordered neighbour set / wavefronts where all are unchanged:", [(int(r >> 32), int(r & 0xffffffff)) for r in raw], file=sys.stderr)
    if rank == 0:
        out = {
            "metric": "scan-to-submap registrations/sec (64x1800 pts, 200k submap)",
            "value": round(value, 2), "unit": "registrations/s", "n_gpus": n_gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "step_loop_repeats": repeats, "timed_region_s": round(elapsed, 3),
            "config": {"workload": wl_desc,
                       "batch_per_gpu": batch, "scan": [H, W], "submap_points": M_SUBMAP, "gn_iters": ITERS,
                       "source_points_per_batch": int(n_src), "parallelism": f"independent batches x{n_gpus} + RCCL all-gather of results",
                       "process_group": None if not use_dist else {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                                                                   "pose_gather": "lisreg_comm_init / lisreg_gather_results (the library's own RCCL all-gather)" if native_gather else "torch.distributed " + dist.get_backend(),
                                                                   "comm_nranks": comm_nranks, "ranks": rank_devices}},
            "roofline": roof, "cpu_baseline": cpu, "pcl_reference": pcl_reference_note(), "pcie_inclusive": pcie, "two_batches_in_flight": overlap_leg, "interleaved_halves": interleave_leg, "exact_build": exact_leg,
            "accuracy": {"max_rot_err_vs_truth_rad": float(err_truth[:, :3].max()),
                         "max_trans_err_vs_truth_m": float(err_truth[:, 3:].max()),
                         "vs_cpu_oracle": parity,
                         "all_status_ok": bool(all(s["status"] == 0 for s in st_gpu))},
        }
    else:
        out = None
    if use_dist:
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
        emit_json(out)


ICP_ALG_BYTES_PER_POINT_ITER = 48   # per source point and ICP iteration: 16 B read + 16 B written back (transformPointCloud of the working
                                    # copy, icp.hpp applies it in place every iteration) + ONE 16 B neighbour gather (k = 1); search traffic not counted


def bench_icp(args, lisreg, torch, np, synth, synth_torch, dev, dev_index, rank, world, use_dist, native_gather, dist, native_comm,
              rank_devices):
    """--workload cfg4_icp: BASELINE configs[3] with the reference's own verification step (detectLoopClosureForSubMap,
    subMapOptmizationNode.cpp:2776-2840).  A step = for every candidate setInputTarget (k = 1 index of ITS 200k-point target, built inside
    the step like the reference's kd-tree), then ONE lisreg_icp_align_batch over all candidates (chained prev_mse = the static ICP object),
    then the pose gather (final transforms, n x 12 floats) across ranks."""
    H, W, M_SUBMAP, BATCH, ITERS, _, wl_desc = WORKLOADS["cfg4_icp"]
    batch = args.batch if args.batch > 0 else BATCH
    ctx = lisreg.Context(dev_index)
    stream = torch.cuda.Stream(device=dev)
    ctx.set_stream(stream.cuda_stream)
    from lisreg import pack_device_records
    n_distinct = min(batch, 64)
    srcs, T_fix, host_src = [], [], {}
    for i in range(n_distinct):
        seed = 1000 + rank * 64 + i
        c, s_, tt = synth_torch.make_scan_device(H, W, seed, dev)
        rec = torch.cat([c, s_]).contiguous()
        # the reference hands ICP the key frame ALREADY moved by its initial guess (transformPointCloud(cureKeyframeCloud, key2PreSubMapTrans),
        # :2820) and aligns from identity: same here, with the benchmark's initial error (0.3 m, 2 degrees)
        T_bad = synth.perturb_pose(tt, np.random.default_rng(seed + 7919))
        Mb = torch.tensor(synth.pose_matrix(T_bad), dtype=torch.float32, device=dev)
        moved = rec.clone()
        moved[:, :3] = rec[:, :3] @ Mb[:3, :3].T + Mb[:3, 3]
        srcs.append(moved.contiguous())
        T_fix.append(synth.pose_matrix(tt) @ np.linalg.inv(synth.pose_matrix(T_bad)))
    tgts, host_tgt = [], {}
    for i in range(batch):
        tci, tsi = synth.make_submap(M_SUBMAP, 42 + i + 1000 * rank)
        both = np.concatenate([pack_device_records(tci), pack_device_records(tsi)])
        if i < 4:
            host_tgt[i] = synth.concat_clouds([tci, tsi])
        tgts.append(torch.from_numpy(both).to(dev))
    torch.cuda.synchronize()
    items = [(i, (srcs[i % n_distinct].data_ptr(), srcs[i % n_distinct].shape[0]), None) for i in range(batch)]
    n_src = sum(srcs[i % n_distinct].shape[0] for i in range(batch))
    prm = lisreg.icp_default_params(0)
    prm.max_iters = ITERS
    poses_dev = torch.zeros((batch, 12), dtype=torch.float32, device=dev)
    gathered = torch.zeros((world, batch, 12), dtype=torch.float32, device=dev)
    comm_nranks = native_comm(ctx) if native_gather else None
    last = {}

    def step():
        if os.environ.get("LISREG_BENCH_ICP_SINGLE_SETS"):
            for i in range(batch):
                ctx.map_index_set_device(i, tgts[i].data_ptr(), tgts[i].shape[0])    # setInputTarget (:2793), one call per candidate
        else:                                                                        # ... or all candidates' targets in one call
            ctx.map_index_set_batch(list(range(batch)), [(tgts[i].data_ptr(), tgts[i].shape[0]) for i in range(batch)])
        res = ctx.icp_align_batch(items, prm, chain_prev_mse=True)
        last["res"] = res
        if use_dist:
            P = np.stack([r["T"][:3].reshape(12) for r in res]).astype(np.float32)
            poses_dev.copy_(torch.from_numpy(P), non_blocking=False)
            if native_gather:
                ctx.gather_results(poses_dev.data_ptr(), batch, gathered.data_ptr())
            elif dist.get_backend() == "nccl":
                dist.all_gather_into_tensor(gathered.view(-1), poses_dev.view(-1))

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    warm = max(1, min(args.warmup, 2))
    for _ in range(warm):
        step()
    barrier()
    steps = max(1, min(args.steps, 10))
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tt_ = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else dev)
        dist.all_reduce(tt_, op=dist.ReduceOp.MAX)
        elapsed = float(tt_.item())
    res = last["res"]
    # one more call with HIP events around every k_icp_assoc launch (and chain off, so that the events of ONE run are read)
    ctx.set_profiling(True)
    res_p = ctx.icp_align_batch(items, prm, chain_prev_mse=False)
    timing = ctx.timing()
    ctx.set_profiling(False)
    it_sum = int(sum(r["iters"] for r in res_p))
    launches = int(timing["assoc_launches"])
    # a launch works on the candidates still iterating: point-iterations actually done = sum over candidates of iterations x points
    pt_iters = int(sum(r["iters"] * items[k][1][1] for k, r in enumerate(res_p)))
    avg_ms = timing["assoc_ms"] / max(launches, 1)
    alg_bytes_launch = ICP_ALG_BYTES_PER_POINT_ITER * pt_iters / max(launches, 1)
    achieved = alg_bytes_launch / (avg_ms * 1e-3) / 1e9
    roof = dict(bound="hbm", kernel="k_icp_assoc", achieved=round(achieved, 2), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 5),
                traffic=None, avg_launch_ms=round(avg_ms, 4), launches=launches,
                algorithmic_bytes_per_launch=int(alg_bytes_launch),
                note=f"{ICP_ALG_BYTES_PER_POINT_ITER} B per source point and ICP iteration (16 B read + 16 B working copy written + one 16 B neighbour gather) x the "
                     f"point-iterations of the candidates still running ({pt_iters} over {launches} launches, {it_sum} candidate-iterations); "
                     "k = 1 search traffic (grid cells, candidates) is overhead and not counted")
    # accuracy: the correction ICP should find moves the mis-placed key frame onto the map
    err_t = max(float(np.abs(res[k]["T"][:3, 3] - T_fix[k % n_distinct][:3, 3]).max()) for k in range(batch))
    n_conv = int(sum(1 for r in res if r["converged"]))
    cpu = parity = None
    if rank == 0 and args.cpu_regs > 0:            # (at N > 1 the other ranks wait at the caller's barrier)
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle_ctypes as oc
        oc.build()
        po = oc.icp_default_params(0); po.max_iters = ITERS
        kk = min(2, len(host_tgt))
        tc0 = time.perf_counter()
        ros = []
        for k in range(kk):
            src_h = synth_torch.records_to_pcl(srcs[k % n_distinct])
            ro_ = oc.icp_align(host_tgt[k], src_h, po)
            ros.append(ro_); po.prev_mse = ro_["prev_mse"]
        tcpu = time.perf_counter() - tc0
        cpu = dict(value=round(kk / tcpu, 4), unit="candidates/s", cores=1, kind="port", seconds=round(tcpu, 2), pcl_reference=pcl_reference_note(),
                   sample=f"{kk} of the {batch} candidates ({H}x{W} key frame vs its 200k-point target, <= {ITERS} ICP iterations, kd-tree build "
                          "included) through the CPU restatement of pcl::IterativeClosestPoint, 1 thread (PCL's ICP is single-threaded)")
        dT = max(float(np.abs(res[k]["T"].astype(np.float64) - ros[k]["T"].astype(np.float64)).max()) for k in range(kk))
        parity = dict(items=kk, max_transform_entry_diff=dT, iterations=[(int(res[k]["iters"]), int(ros[k]["iters"])) for k in range(kk)],
                      states=[(int(res[k]["state"]), int(ros[k]["state"])) for k in range(kk)],
                      fitness=[(float(res[k]["fitness"]), float(ros[k]["fitness"])) for k in range(kk)])
    if use_dist and (native_gather or dist.get_backend() == "nccl"):
        g = gathered.cpu().numpy()
        mine = np.stack([r["T"][:3].reshape(12) for r in res]).astype(np.float32)
        assert np.array_equal(g[rank], mine), "gathered pose block differs from the local results"
    ctx.close()
    if rank != 0:
        return None
    return {"metric": "loop-closure candidate registrations/sec (ICP verification, 64x1800 key frame vs own 200k submap)",
            "value": round(world * batch * steps / elapsed, 2), "unit": "candidates/s", "n_gpus": world, "steps": steps, "warmup": warm,
            "ms_per_step": round(1e3 * elapsed / steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": wl_desc, "batch_per_gpu": batch, "scan": [H, W], "submap_points": M_SUBMAP, "max_icp_iters": ITERS,
                       "source_points_per_batch": int(n_src),
                       "process_group": None if not use_dist else {"backend": dist.get_backend(), "world_size": world, "comm_nranks": comm_nranks,
                                                                   "pose_gather": "lisreg_comm (native RCCL)" if native_gather else dist.get_backend(),
                                                                   "ranks": rank_devices}},
            "roofline": roof, "cpu_baseline": cpu,
            "accuracy": {"converged": n_conv, "candidates": batch, "max_translation_err_vs_truth_m": err_t, "vs_cpu_oracle": parity,
                         "iterations_min_max": [int(min(r["iters"] for r in res)), int(max(r["iters"] for r in res))]}}


def pcl_reference_note():
    """'if find_package(PCL QUIET) succeeds on the box, additionally time the verbatim KdTreeFLANN path; otherwise say
    that it was unavailable'.  Looked for here the way CMake's find_package would find it (a PCLConfig.cmake / pcl_common pkg-config file /
    the kdtree_flann header) — the image carries no PCL, FLANN, Eigen or OpenCV, so the verbatim path cannot be built and the timed CPU path
    is the oracle port (`kind: port`)."""
    import glob
    hits = []
    for pat in ("/usr/lib/*/cmake/pcl/PCLConfig.cmake", "/usr/share/pcl*/PCLConfig.cmake", "/usr/local/share/pcl*/PCLConfig.cmake",
                "/usr/lib/*/pkgconfig/pcl_common*.pc", "/usr/include/pcl*/pcl/kdtree/kdtree_flann.h", "/usr/local/include/pcl*/pcl/kdtree/kdtree_flann.h",
                "/opt/ros/*/include/pcl*/pcl/kdtree/kdtree_flann.h"):
        hits += glob.glob(pat)
    if hits:
        return "found (" + hits[0] + ") but not wired: the verbatim leg needs Eigen, FLANN and OpenCV as well"
    return "unavailable (find_package(PCL) would fail: no PCLConfig.cmake, pcl_common.pc or kdtree_flann.h on this box)"


def chain_cpu_baseline(workload, frames, recs, np):
    """cfg3 / odom: the CPU restatement's own frame loop (oracle/replay_oracle.py) on the first frames of the same drive — at ONE thread (faithful:
    -fopenmp never reaches the reference's compiler, CMakeLists.txt:5-6, and params.yaml:127 asks for numberOfCores 2) and at min(16, usable
    cores); never at the affinity mask's size (256 on the pool's hosts, which give the process 16 CPUs: oversubscribed, slower than 1 thread)."""
    try:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle_ctypes as oc
        import replay_oracle as ro
        kk = min(len(frames), 8)
        legs, dmax = {}, None
        for nthreads in sorted({1, max(1, min(16, host_cores()))}):
            tc0 = time.perf_counter()
            if workload == "cfg3":
                ref = ro.replay(list(frames[:kk]), n_threads=nthreads)
            else:
                ref = ro.replay_odom(list(frames[:kk]), oc.default_feature_params(), n_threads=nthreads)
            tcpu = time.perf_counter() - tc0
            legs[str(nthreads)] = dict(value=round(kk / tcpu, 3), seconds=round(tcpu, 2), frames=kk)
            if nthreads == 1:
                dmax = max(float(np.abs(np.asarray(a["T"], np.float64) - np.asarray(b["T"], np.float64)).max()) for a, b in zip(recs[:kk], ref))
        what = ("semantic split, per-class voxel grids, sliding local map, copy #2 registration with its two kd-tree builds per frame" if workload == "cfg3"
                else "range image + LOAM features, key-frame target with its voxel grids, copy #1 registration, key-frame gate")
        return dict(value=legs["1"]["value"], unit="frames/s", cores=1, kind="port", seconds=legs["1"]["seconds"], by_threads=legs,
                    pcl_reference=pcl_reference_note(),
                    sample=f"the first {kk} frames of the same drive through the CPU restatement's frame loop ({what}); headline = 1 thread, "
                           f"by_threads also has min(16, usable cores) OpenMP threads over the feature points ({host_cores()} cores usable here)",
                    max_pose_diff_vs_hip_chain=dmax)
    except Exception as e:                                      # the baseline is a report, never a reason to lose the line
        return dict(error=repr(e))


def host_cores():
    """cores this process may actually run on: scheduler affinity, capped by the cgroup CPU quota if there is one."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, int(q / per + 0.5)))
        except Exception:
            pass
    return max(1, n)


def pcie_inclusive_leg(lisreg, torch, np, dev_index, stream, scans, tc_dev, ts_dev, T_init, params, T_ref, steps, feeder_threads=0):
    """The same batch with the sources living in pinned HOST memory as PCL PointXYZI structs (32 B per point): every timed
    step uploads them (hipMemcpyAsync on the context's stream), packs them on the device, runs the registration and
    copies poses + stats back — SURVEY.md §8(d)'s "incl. H2D of sources and D2H of poses".  Measured twice: one context (copy and
    compute in series) and two contexts on two host threads / two streams (the next batch's upload overlaps this batch's kernels;
    contexts are independent and thread-safe per context, like the reference's concurrently running node objects)."""
    import ctypes as C
    import threading
    host, n_bytes = [], 0
    for c, s in scans:
        pair = []
        for rec in (c, s):
            h = torch.zeros((rec.shape[0], 8), dtype=torch.float32).pin_memory()        # x y z pad intensity label(u16) pad pad
            h[:, :3] = rec[:, :3].cpu()
            pair.append(h); n_bytes += h.numel() * 4
        host.append(pair)
    n = len(scans)
    arr = (lisreg.Item * n)()
    for i, (hc, hs) in enumerate(host):
        arr[i].src_corner = C.c_void_p(hc.data_ptr()) if hc.shape[0] else None; arr[i].n_corner = hc.shape[0]
        arr[i].src_surf = C.c_void_p(hs.data_ptr()) if hs.shape[0] else None; arr[i].n_surf = hs.shape[0]
        arr[i].stride_bytes = 32; arr[i].fmt = lisreg.FMT_XYZI; arr[i].target = 0

    class Lane:
        def __init__(self):
            self.ctx = lisreg.Context(dev_index)
            self.ctx.set_target_device(tc_dev.data_ptr(), tc_dev.shape[0], ts_dev.data_ptr(), ts_dev.shape[0])
            self.T = np.ascontiguousarray(T_init, np.float32).copy()
            self.st = (lisreg.Stats * n)()
            self.staged = (lisreg.Item * n)()
            self.Tin = np.ascontiguousarray(T_init, np.float32).copy()

        def one(self):
            self.T[:] = T_init
            rc = self.ctx._L.lisreg_align_batch(self.ctx._h, n, arr, C.byref(params), self.T.ctypes.data_as(C.POINTER(C.c_float)), self.st)
            if rc:
                raise RuntimeError(f"lisreg_align_batch failed: {rc}")

        # the pipelined form of the same work: stage (feeder threads pack + asynchronous upload), prepare + run, fetch
        def stage(self):
            rc = self.ctx._L.lisreg_stage_host_items(self.ctx._h, n, arr, self.staged)
            if rc:
                raise RuntimeError(f"lisreg_stage_host_items failed: {rc}")

        def launch(self):
            L = self.ctx._L
            rc = L.lisreg_batch_prepare(self.ctx._h, n, self.staged, C.byref(params), self.Tin.ctypes.data_as(C.POINTER(C.c_float)))
            rc = rc or L.lisreg_batch_run(self.ctx._h)
            if rc:
                raise RuntimeError(f"lisreg_batch_prepare / run failed: {rc}")

        def fetch(self):
            rc = self.ctx._L.lisreg_batch_fetch(self.ctx._h, self.T.ctypes.data_as(C.POINTER(C.c_float)), self.st)
            if rc:
                raise RuntimeError(f"lisreg_batch_fetch failed: {rc}")

    a = Lane()
    a.ctx.set_option("rebuild_targets_each_run", 1)
    if feeder_threads > 0:                    # several ranks on one host share its cores: each takes its share
        a.ctx.set_option("feeder_threads", feeder_threads)
    for kv in filter(None, os.environ.get("LISREG_OPTS", "").split(",")):        # tuning experiments, as for the main context
        k_, v_ = kv.split("=")
        a.ctx.set_option(k_, int(v_))
    a.one(); a.one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        a.one()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    same = bool(np.array_equal(a.T, T_ref))
    # ONE context, pipelined: batch k + 1 is packed by the feeder threads and uploaded on the copy stream while batch k runs
    a.stage(); a.launch(); a.stage(); a.fetch(); a.launch(); a.fetch()            # warm both staging buffers
    torch.cuda.synchronize()
    ksteps = 3 * steps
    runs = []
    phase = dict(stage=0.0, fetch=0.0, prepare=0.0, run=0.0)      # host time of the calling thread per phase, last run of the three
    for rep in range(3):                      # three back-to-back runs of the pipelined loop: the host side is the variable part
        for k_ in phase: phase[k_] = 0.0
        t0 = time.perf_counter()
        a.stage(); a.launch()
        for _ in range(ksteps - 1):
            ta = time.perf_counter()
            a.stage()                 # k + 1: host packing + H2D, underneath the kernels of k
            tb = time.perf_counter()
            a.fetch()                 # k: D2H of poses and stats
            tc = time.perf_counter()
            L_ = a.ctx._L
            rc_ = L_.lisreg_batch_prepare(a.ctx._h, n, a.staged, C.byref(params), a.Tin.ctypes.data_as(C.POINTER(C.c_float)))
            td = time.perf_counter()
            rc_ = rc_ or L_.lisreg_batch_run(a.ctx._h)       # k + 1
            te = time.perf_counter()
            if rc_:
                raise RuntimeError(f"lisreg_batch_prepare / run failed: {rc_}")
            phase["stage"] += tb - ta; phase["fetch"] += tc - tb; phase["prepare"] += td - tc; phase["run"] += te - td
        a.fetch()
        torch.cuda.synchronize()
        runs.append(time.perf_counter() - t0)
    dtp = sorted(runs)[1]
    samep = bool(np.array_equal(a.T, T_ref))
    by_engine, n_chunks = a.ctx.get_option("feeder_chunks_by_copy_engine"), a.ctx.get_option("feeder_chunks")
    a_threads, a_node, a_cpus = a.ctx.get_option("feeder_threads"), a.ctx.get_option("feeder_numa_node"), a.ctx.get_option("feeder_numa_cpus")
    # the upload alone (stage + wait), for the achieved link rate
    t0 = time.perf_counter()
    for _ in range(steps):
        a.stage()
        torch.cuda.synchronize()          # device-wide: includes the context's copy stream
    dtu = (time.perf_counter() - t0) / steps
    a.ctx.close()
    return dict(value=round(n * ksteps / dtp, 2), unit="registrations/s", ms_per_step=round(1e3 * dtp / ksteps, 3), steps=ksteps,
                runs=[round(n * ksteps / t, 2) for t in runs], spread="value = the median of three runs of the pipelined loop; `runs` lists all three",
                feeder=dict(threads=a_threads, numa_node=a_node, cpus_bound=a_cpus,
                            note="packing threads bound to the CPUs of the device's NUMA node that this process owns (0 = no such CPU / no sysfs entry: unbound); "
                                 "the pinned staging buffers come from hipHostMalloc, i.e. from that node"),
                h2d_struct_bytes_per_step=int(n_bytes), h2d_link_bytes_per_step=int(n_bytes // 2), d2h_bytes_per_step=int(n * 12 * 4),
                note="pinned host PCL structs (32 B/pt) in, poses and stats out, every step, ONE context: lisreg_stage_host_items (feeder threads "
                     "pack the structs to 16-byte records in pinned staging, chunks uploaded on a copy stream as they complete, the copy engine working the other end of the batch) for batch k+1 "
                     "runs underneath the kernels of batch k; then fetch(k), prepare + run(k+1).  The step includes the target index build and "
                     "10 GN iterations like `value` of the main line.  `in_series`: the synchronous lisreg_align_batch (stage, run, fetch back to back)",
                poses_equal_device_resident_run=samep,
                chunks_taken_by_copy_engine=f"{by_engine} of {n_chunks} in the last pipelined step (structs that cross as they are — 32 B per point — and are "
                                            "packed on the device: the copy engine takes chunks from the far end of the batch whenever it is idle and the next packed chunk is not ready)",
                stage_ms=round(1e3 * dtu, 3),
                host_ms_per_step_by_phase={k_: round(1e3 * v_ / max(ksteps - 1, 1), 3) for k_, v_ in phase.items()},
                host_ms_note="host time of the calling thread inside each call of the pipelined loop (last of the three runs): stage = feeder threads pack + copies are queued (blocks until the batch is packed), fetch = waits for the running batch, prepare = tables + marks, run = enqueue",
                link_rate_GBps=round((n_bytes // 2) / dtu * 1e-9, 2),
                link_rate_note="lisreg_stage_host_items alone, to completion: host packing of the structs (feeder threads) with the H2D copies of the 16-byte records following chunk by chunk; link bytes / that time — a lower bound of the achieved H2D rate",
                in_series=dict(value=round(n * steps / dt, 2), ms_per_step=round(1e3 * dt / steps, 3), steps=steps,
                               poses_equal_device_resident_run=same))


if __name__ == "__main__":
    main()
