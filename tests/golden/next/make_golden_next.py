"""Generates tests/golden/next/next_rows.npz: small inputs + expected outputs for the SURVEY.md §8 f rows, computed with
numpy / scipy only (the independent mirrors the tests also use live) — neither the C oracle nor the HIP library is involved.
Run from the repo root:  python tests/golden/next/make_golden_next.py"""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
for p in (os.path.join(ROOT, "lis-slam_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
from lisreg import synth
import lisreg_numpy as ln
import test_voxel, test_features, test_mapfilter, test_icp

f32 = np.float32
out = {}
def put_cloud(name, c):
    for f in c.dtype.names:
        out[f"{name}.{f}"] = np.ascontiguousarray(c[f])

# ---- f-1: voxel grid ---------------------------------------------------------------------------------------------------
cv = test_voxel._cloud(901, n_scan=(8, 300))
cen, lab = test_voxel._numpy_voxel(cv, 0.4)
put_cloud("vox_in", cv); out["vox_leaf"] = f32(0.4); out["vox_centroid"] = cen; out["vox_label"] = lab

# ---- f-2: features + de-skew -------------------------------------------------------------------------------------------
raw = synth.make_raw_scan(8, 240, 902)
P = dict(n_scan=8, horizon_scan=240, downsample_rate=1, min_range=0.0, max_range=70.0, edge_threshold=1.0, surf_threshold=0.1)
r = ln.extract_features(raw["x"], raw["y"], raw["z"], raw["ring"], P)
put_cloud("feat_in", raw)
for k, v in r.items():
    out[f"feat_{k}"] = np.asarray(v, np.int32)
t, rot = test_features._imu_tables(903)
out["imu_time"], out["imu_rot"] = t, rot
out["deskew_xyz"] = test_features._numpy_deskew(raw, np.asarray(r["deskewed"]), t, rot, 100.0)

# ---- f-3: map filters --------------------------------------------------------------------------------------------------
m, q = test_mapfilter._scene(904, n_map=6000)
q = q[:2500].copy()
_, d2 = test_mapfilter._d2_numpy(m, q)
r2 = q["x"] * q["x"] + q["y"] * q["y"]
keep = (r2 > f32(25.0) * f32(25.0)) | ((d2 > f32(0.05) * f32(0.05)) & (d2 < f32(0.3) * f32(0.3))) | (d2 > f32(1.0) * f32(1.0))
put_cloud("map", m); put_cloud("map_q", q)
out["nn_d2"] = d2; out["dyn_keep"] = keep
xyz = synth.pcl_xyz(q).astype(np.float64)
box = np.array([float(q["x"][17]), -30.0, -2.0, float(q["x"][17]) + 25.0, 30.0, 10.0])
out["box"] = box; out["box_inside"] = np.all((xyz > box[:3]) & (xyz < box[3:]), axis=1)
out["bounds"] = np.concatenate([xyz.min(0), xyz.max(0)])

# ---- f-4: ICP (PCL-style) and OptimizedICPGN ---------------------------------------------------------------------------
tgt, src, _ = test_icp._case(905, n_map=6000, trans=0.5, rot_deg=2.0, hw=(8, 240))
put_cloud("icp_tgt", tgt); put_cloud("icp_src", src)
F, iters, state, prev = test_icp._numpy_icp(tgt, src, 10.0, 30, 1e-4, 1e-4)
out["icp_T"], out["icp_iters"], out["icp_state"] = F, np.int32(iters), np.int32(state)
Tg, fit = test_icp._numpy_icp_gn(tgt, src, 12, 4.0, np.eye(4))
out["gn_T"], out["gn_fitness"] = Tg, np.float64(fit)
np.savez_compressed(os.path.join(HERE, "next_rows.npz"), **out)
print("wrote", os.path.join(HERE, "next_rows.npz"), {k: np.asarray(v).shape for k, v in out.items() if not "." in k})
