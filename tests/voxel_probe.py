"""Ad-hoc measurement (not a test) for SURVEY.md §8 f-1: device-resident voxel-grid down-sampling and cloud transform,
timed with HIP events on the library's stream, next to the CPU restatement (1 thread)."""
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
def timed(fn, reps=20):
    fn(); s, e = ev(), ev(); st = C.c_void_p(ctx.stream)
    hip.hipEventRecord(s, st)
    for _ in range(reps): fn()
    hip.hipEventRecord(e, st); hip.hipEventSynchronize(e)
    ms = C.c_float(); hip.hipEventElapsedTime(C.byref(ms), s, e); return ms.value / reps
cases = []
tc, ts = synth.make_submap(200000, 42, labelled=True)
cases.append(("200k submap surf, leaf 0.4", ts, 0.4))
sc = synth.make_scan(64, 1800, 1000, labelled=True)
cases.append(("64x1800 scan surf (110k), leaf 0.4", sc["surf"], 0.4))
cases.append(("64x1800 scan surf (110k), leaf 0.2", sc["surf"], 0.2))
big = synth.concat_clouds([synth.make_submap(200000, 100 + k, labelled=True)[1] for k in range(10)])
cases.append(("assembled local map 1.9M (10 keyframe-sized clouds), leaf 0.4", big, 0.4))
for name, cloud, leaf in cases:
    rec = lisreg.pack_device_records(cloud); n = len(cloud)
    din, dout = lisreg.DeviceArray(rec), lisreg.DeviceArray(np.zeros_like(rec))
    res = {}
    def run():
        res["r"] = ctx.voxel_downsample_device(din.ptr, n, leaf, dout.ptr, n)
    ms = timed(run, 10)
    t0 = time.perf_counter(); rc, do = oc.voxel_grid(cloud, leaf); cpu = 1e3 * (time.perf_counter() - t0)
    n_out = res["r"][1]
    alg = n * (16 + 16 + 16) + n_out * 16          # read + sort key/index traffic + gather + output
    print(f"{name}: n={n} -> {n_out} (oracle {len(do)})  GPU {ms:.3f} ms ({n/ms/1e3:.1f} Mpts/s, {alg/ms/1e6:.1f} GB/s of ~48 B/pt)  CPU oracle {cpu:.1f} ms  x{cpu/ms:.0f}")
    T = np.array([0.01, 0.02, 0.5, 1, 2, 3], np.float32)
    mt = timed(lambda: ctx.transform_cloud_device(din.ptr, n, T, dout.ptr), 10)
    print(f"    transform_cloud: {mt:.4f} ms ({32*n/mt/1e6:.1f} GB/s of 32 B/pt; includes one 48-B H2D + sync per call)")
