"""Ad-hoc measurement (not a test): host-side time per stage of the replay drivers (lisreg.replay), per frame."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lis-slam_amd"))
import lisreg
from lisreg import replay

def wrap(ctx, names, T):
    for nm in names:
        fn = getattr(ctx, nm)
        def w(*a, _fn=fn, _nm=nm, **k):
            t0 = time.perf_counter(); out = _fn(*a, **k); T[_nm] = T.get(_nm, 0.0) + time.perf_counter() - t0; return out
        setattr(ctx, nm, w)

n = 40
for title, make, frames in (
        ("submap loop, host clouds", lambda c: replay.Replayer(c, 2), [c for c, _ in replay.synthetic_drive(n)]),
        ("submap loop, device-resident", lambda c: replay.DeviceReplayer(c, 2), [c for c, _ in replay.synthetic_drive(n)]),
        ("odometry loop, host clouds", lambda c: replay.OdomReplayer(c), [c for c, _ in replay.synthetic_raw_drive(n)]),
        ("odometry loop, device-resident", lambda c: replay.DeviceOdomReplayer(c), [c for c, _ in replay.synthetic_raw_drive(n)])):
    ctx = lisreg.Context(0)
    T = {}
    wrap(ctx, ("semantic_split", "semantic_split_device", "voxel_downsample", "voxel_downsample_device", "voxel_downsample_multi_device", "localmap_extract", "localmap_insert",
               "localmap_insert_device", "align", "align_device", "extract_features", "extract_features_device", "keyframes_target",
               "keyframes_push", "keyframes_push_device"), T)
    r = make(ctx)
    for k, c in enumerate(frames):
        if k == 10:
            T.clear(); t0 = time.perf_counter()
        r.step(c)
    tot = time.perf_counter() - t0
    print(title, "— ms per frame:", {k: round(1e3 * v / (n - 10), 3) for k, v in T.items()}, "total", round(1e3 * tot / (n - 10), 3))
    ctx.close()
