"""Ad-hoc measurement (not a test) for SURVEY.md §8 f-2: device-resident projection + feature extraction per sweep."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("lis-slam_amd", "oracle"): sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, lisreg, oracle_ctypes as oc
from lisreg import synth
oc.build()
ctx = lisreg.Context(0); hip = C.CDLL("libamdhip64.so")
def ev():
    e = C.c_void_p(); hip.hipEventCreate(C.byref(e)); return e
def timed(fn, reps=20):
    fn(); s, e = ev(), ev(); st = C.c_void_p(ctx.stream)
    hip.hipEventRecord(s, st)
    for _ in range(reps): fn()
    hip.hipEventRecord(e, st); hip.hipEventSynchronize(e)
    ms = C.c_float(); hip.hipEventElapsedTime(C.byref(ms), s, e); return ms.value / reps
for h, w, rate in ((64, 1800, 2), (64, 1800, 1), (128, 2048, 1)):
    c = synth.make_raw_scan(h, w, 8000 + h)
    rec = np.zeros((len(c), 4), np.float32); rec[:, 0], rec[:, 1], rec[:, 2] = c["x"], c["y"], c["z"]
    rec[:, 3] = c["ring"].astype(np.uint32).view(np.float32)
    din = lisreg.DeviceArray(rec); cap = h * w
    names = ("deskewed", "corner", "surface", "corner_sharp", "surface_sharp")
    outs = {k: lisreg.DeviceArray(np.zeros((cap, 4), np.float32)) for k in names}
    pg = lisreg.FeatureParams(h, w, rate, 0.0, 70.0, 1.0, 0.1); po = oc.FeatureParams(h, w, rate, 0.0, 70.0, 1.0, 0.1)
    res = {}
    def run(): res["c"] = ctx.extract_features_device(din.ptr, len(c), pg, {k: v.ptr for k, v in outs.items()}, cap)
    ms = timed(run, 20)
    t0 = time.perf_counter(); ro = oc.extract_features(c, po); cpu = 1e3 * (time.perf_counter() - t0)
    print(f"{h}x{w} rate {rate}: {len(c)} pts -> {res['c']}  GPU {ms:.3f} ms ({len(c)/ms/1e3:.0f} Mpts/s)  CPU oracle {cpu:.2f} ms  x{cpu/ms:.1f}")

# batched: 8 sweeps in one pass (grid = sweeps x rings)
h, w, rate, S = 64, 1800, 2, 8
sweeps = [synth.make_raw_scan(h, w, 8100 + k) for k in range(S)]
recs = []
for c in sweeps:
    rec = np.zeros((len(c), 4), np.float32); rec[:, 0], rec[:, 1], rec[:, 2] = c["x"], c["y"], c["z"]
    rec[:, 3] = c["ring"].astype(np.uint32).view(np.float32); recs.append(rec)
dins = [lisreg.DeviceArray(r) for r in recs]; cap = h * w
names = ("deskewed", "corner", "surface", "corner_sharp", "surface_sharp")
outs = [{k: lisreg.DeviceArray(np.zeros((cap, 4), np.float32)) for k in names} for _ in range(S)]
pg = lisreg.FeatureParams(h, w, rate, 0.0, 70.0, 1.0, 0.1)
def runb(): ctx.extract_features_batch_device([d.ptr for d in dins], [len(r) for r in recs], pg, [{k: v.ptr for k, v in o.items()} for o in outs], cap)
ms = timed(runb, 20)
t0 = time.perf_counter(); runb(); wall = 1e3 * (time.perf_counter() - t0)
print(f"batched {S} x {h}x{w} rate {rate}: GPU {ms:.3f} ms total ({ms / S:.3f} ms per sweep), host-inclusive call {wall:.3f} ms")
