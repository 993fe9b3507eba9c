import sys, time, os
import numpy as np
sys.path.insert(0, "/root/repo/lis-slam_amd"); sys.path.insert(0, "/root/repo/oracle")
import lisreg, oracle_ctypes as oc
from lisreg import replay
oc.build()
frames = [c for c, _ in replay.synthetic_drive(3)]
parts = oc.semantic_split(frames[1])
full = dict(dynamic=parts[0], ground=parts[1], building=parts[2], pole=parts[3], outlier=parts[4])
order = ("dynamic", "pole", "ground", "building", "outlier")
leaf = [0.2, 0.05, 0.6, 0.4, 0.6]
ctx = lisreg.Context(0)
recs = [lisreg.pack_device_records(full[k]) for k in order]
print("class sizes", [len(r) for r in recs])
ins = [lisreg.DeviceArray(r if len(r) else np.zeros((1,4),np.float32)) for r in recs]
outs = [lisreg.DeviceArray(np.zeros((max(len(r),1),4),np.float32)) for r in recs]
def singles():
    return [ctx.voxel_downsample_device(i.ptr, len(r), lf, o.ptr, max(len(r),1))[1] if len(r) else 0 for i, r, lf, o in zip(ins, recs, leaf, outs)]
def multi():
    return ctx.voxel_downsample_multi_device([i.ptr for i in ins], [len(r) for r in recs], leaf, [o.ptr for o in outs], [max(len(r),1) for r in recs])
import torch
for name, fn in (("singles", singles), ("multi", multi), ("singles", singles), ("multi", multi)):
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): r = fn()
    torch.cuda.synchronize()
    print(name, r, f"{(time.perf_counter()-t0)/50*1e3:.3f} ms")
