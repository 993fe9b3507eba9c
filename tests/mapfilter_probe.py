"""Ad-hoc measurement (not a test) for SURVEY.md §8 f-3: map index build, k = 1 search, dynamic-point filter and box crop on
device-resident clouds, timed with HIP events on the library's stream, next to the CPU restatement (1 thread, kd-tree)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("lis-slam_amd", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import lisreg, oracle_ctypes as oc
from lisreg import synth
oc.build()
ctx = lisreg.Context(0)
hip = C.CDLL("libamdhip64.so")
def ev():
    e = C.c_void_p(); hip.hipEventCreate(C.byref(e)); return e
def timed(fn, reps=10):
    fn(); s, e = ev(), ev(); st = C.c_void_p(ctx.stream)
    hip.hipEventRecord(s, st)
    for _ in range(reps): fn()
    hip.hipEventRecord(e, st); hip.hipEventSynchronize(e)
    ms = C.c_float(); hip.hipEventElapsedTime(C.byref(ms), s, e); return ms.value / reps
for n_map, (h, w) in ((200000, (64, 1800)), (1000000, (64, 1800))):
    mc, ms_ = synth.make_submap(n_map, 42, labelled=True)
    def cat(a, b):
        o = np.zeros(len(a) + len(b), a.dtype); o[: len(a)], o[len(a):] = a, b; return o
    m = cat(mc, ms_)
    sc = synth.make_scan(h, w, 1000, labelled=True)
    q = cat(sc["corner"], sc["surf"])
    M = synth.pose_matrix(sc["T_true"])
    wq = synth.pcl_xyz(q).astype(np.float64) @ M[:3, :3].T + M[:3, 3]
    wq[:, :2] += np.random.default_rng(1).normal(0, 0.3, (len(q), 2))
    q["x"], q["y"], q["z"] = wq[:, 0], wq[:, 1], wq[:, 2]
    rm, rq = lisreg.pack_device_records(m), lisreg.pack_device_records(q)
    dm, dq, dout = lisreg.DeviceArray(rm), lisreg.DeviceArray(rq), lisreg.DeviceArray(np.zeros_like(rq))
    didx, dd2 = lisreg.DeviceArray(np.zeros(len(q), np.int32)), lisreg.DeviceArray(np.zeros(len(q), np.float32))
    t_build = timed(lambda: ctx.map_index_set_device(3, dm.ptr, len(m)), 5)
    print(f"map {len(m)} pts, scan {len(q)} pts")
    print(f"  map_index_set (device cloud): {t_build:.3f} ms")
    for cap in (1.0, 3.0, 1e18):
        t = timed(lambda: ctx.nearest_device(3, dq.ptr, len(q), cap, didx.ptr, dd2.ptr))
        print(f"  nearest cap {cap:g}: {t:.3f} ms ({len(q)/t/1e3:.1f} Mq/s)")
    res = {}
    def filt():
        res["n"] = ctx.dynamic_filter_device(3, dq.ptr, len(q), 30.0, 0.3, 1.0, 0.05, dout.ptr)
    t = timed(filt)
    t0 = time.perf_counter(); want, _ = oc.dynamic_filter(m, q, 30.0, 0.3, 1.0, 0.05); cpu = 1e3 * (time.perf_counter() - t0)
    print(f"  dynamic_filter: {t:.3f} ms  kept {res['n']} (oracle {len(want)})  CPU oracle incl. kd-tree build {cpu:.1f} ms  x{cpu/t:.0f}")
    b = oc.cloud_bounds(q)
    dmo = lisreg.DeviceArray(np.zeros_like(rm))
    def crop():
        res["c"] = ctx.bbx_filter_device(dm.ptr, len(m), b, False, dmo.ptr)
    t = timed(crop)
    t0 = time.perf_counter(); wc = oc.bbx_filter(m, b); cpu = 1e3 * (time.perf_counter() - t0)
    print(f"  bbx_filter of the map: {t:.3f} ms ({(16*len(m)+16*res['c']+8*len(m))/t/1e6:.1f} GB/s)  kept {res['c']} (oracle {len(wc)})  CPU {cpu:.1f} ms")

# ---- §8 f-4: ICP (loop-closure settings) on the same clouds; source = the scan pushed off by a pose error
import test_icp as ti                      # tests/ is this file's directory
for n_map in (200000, 1000000):
    tgt, src, _ = ti._case(61, n_map=n_map, trans=1.0, rot_deg=4.0, hw=(64, 1800))
    rt, rs = lisreg.pack_device_records(tgt), lisreg.pack_device_records(src)
    dt_, ds = lisreg.DeviceArray(rt), lisreg.DeviceArray(rs)
    ctx.map_index_set_device(4, dt_.ptr, len(rt))
    pg = lisreg.icp_default_params(0)
    res = {}
    def run():
        res["r"] = ctx.icp_align_device(4, ds.ptr, len(rs), pg)
    t = timed(run, 5)
    t0 = time.perf_counter(); ro = oc.icp_align(tgt, src, oc.icp_default_params(0)); cpu = 1e3 * (time.perf_counter() - t0)
    r = res["r"]
    print(f"ICP target {len(tgt)} source {len(src)}: GPU {t:.3f} ms for {r['iters']} iterations + fitness ({t/(r['iters']+1):.3f} ms per k=1 pass), "
          f"pose diff vs oracle {ti._pose_diff(r['T'], ro['T'])}, state {r['state']}, fitness {r['fitness']:.5f} | CPU oracle {cpu:.0f} ms ({ro['iters']} it, fitness {ro['fitness']:.5f})  x{cpu/t:.0f}")
