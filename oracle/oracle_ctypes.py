"""ctypes binding of oracle/liblisreg_oracle.so — TEST INFRASTRUCTURE ONLY (see lisreg_oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
TRACE_STRIDE = 56


class Params(C.Structure):
    _fields_ = [("max_iters", C.c_int), ("fixed_iters", C.c_int), ("knn_sq_thresh", C.c_float),
                ("conv_deg", C.c_float), ("conv_cm", C.c_float), ("min_corr", C.c_int),
                ("eig_thresh", C.c_float), ("edge_min", C.c_int), ("surf_min", C.c_int),
                ("line_ratio", C.c_float), ("plane_tol", C.c_float), ("accept_s", C.c_float),
                ("use_label_weight", C.c_int), ("label_score", C.c_float * 32),
                ("emulate_matp_shadow", C.c_int), ("skip_empty_target", C.c_int), ("use_imu_blend", C.c_int),
                ("imu_rpy_weight", C.c_float), ("rotation_tol", C.c_float), ("z_tol", C.c_float)]


class Imu(C.Structure):
    _fields_ = [("imu_available", C.c_int), ("imu_roll_init", C.c_float), ("imu_pitch_init", C.c_float)]


class IcpParams(C.Structure):
    _fields_ = [("max_corr_dist", C.c_double), ("max_iters", C.c_int), ("reserved", C.c_int),
                ("transformation_epsilon", C.c_double), ("euclidean_fitness_epsilon", C.c_double), ("prev_mse", C.c_double)]


class IcpResult(C.Structure):
    _fields_ = [("final_transform", C.c_float * 16), ("converged", C.c_int), ("iters", C.c_int), ("state", C.c_int),
                ("n_corr_last", C.c_int), ("fitness", C.c_double), ("prev_mse", C.c_double)]


class Deskew(C.Structure):
    _fields_ = [("enabled", C.c_int), ("imu_pointer_cur", C.c_int), ("imu_time", C.POINTER(C.c_double)),
                ("imu_rot_x", C.POINTER(C.c_double)), ("imu_rot_y", C.POINTER(C.c_double)), ("imu_rot_z", C.POINTER(C.c_double)),
                ("time_scan_cur", C.c_double), ("time_device", C.POINTER(C.c_float))]


def make_deskew(imu_time, rot_x, rot_y, rot_z, time_scan_cur, enabled=True, time_device_ptr=0):
    """lisreg_deskew from the integrated IMU tables (imuTime / imuRotX,Y,Z of imuDeskewInfo); keeps the arrays alive."""
    arrs = [np.ascontiguousarray(a, np.float64) for a in (imu_time, rot_x, rot_y, rot_z)]
    d = Deskew()
    d.enabled = 1 if enabled else 0
    d.imu_pointer_cur = len(arrs[0]) - 1
    d.imu_time, d.imu_rot_x, d.imu_rot_y, d.imu_rot_z = [a.ctypes.data_as(C.POINTER(C.c_double)) for a in arrs]
    d.time_scan_cur = float(time_scan_cur)
    d.time_device = C.cast(C.c_void_p(time_device_ptr), C.POINTER(C.c_float)) if time_device_ptr else None
    d._keep = arrs
    return d


class IcpGnResult(C.Structure):
    _fields_ = [("final_transform", C.c_float * 16), ("steps_applied", C.c_int), ("n_corr_last", C.c_int),
                ("fitness", C.c_float), ("reserved", C.c_int)]


class FeatureParams(C.Structure):
    _fields_ = [("n_scan", C.c_int), ("horizon_scan", C.c_int), ("downsample_rate", C.c_int), ("min_range", C.c_float),
                ("max_range", C.c_float), ("edge_threshold", C.c_float), ("surf_threshold", C.c_float)]


def default_feature_params() -> "FeatureParams":
    """config/params.yaml:68-74, 117-118 (KITTI HDL-64 settings)."""
    return FeatureParams(64, 1800, 2, 0.0, 70.0, 1.0, 0.1)


class Stats(C.Structure):
    _fields_ = [("iters", C.c_int), ("deltaR", C.c_float), ("deltaT", C.c_float), ("degenerate", C.c_int),
                ("n_corr_last", C.c_int), ("status", C.c_int)]


LABEL_SCORE = [1.0, 1.0, 0.6, 0.5, 0.8, 0.5, 0.5, 0.5, 0.5, 1.2, 1.2, 1.2, 0.5, 1.0, 0.8, 0.5, 1.3, 0.5, 1.5, 1.5]


def default_params(variant: int = 1) -> Params:
    """Literals of the three reference copies (independent of liblisreg's lisreg_default_params)."""
    p = Params()
    p.max_iters, p.knn_sq_thresh, p.conv_deg, p.conv_cm = {1: (15, 1.0, 0.005, 0.05), 2: (20, 2.0, 0.003, 0.03),
                                                          3: (30, 2.0, 0.002, 0.02)}[variant]
    p.fixed_iters = 0
    p.min_corr, p.eig_thresh, p.edge_min, p.surf_min = 50, 100.0, -1, 100
    p.line_ratio, p.plane_tol, p.accept_s = 3.0, 0.2, 0.1
    p.use_label_weight = 0 if variant == 1 else 1
    for i in range(32):
        p.label_score[i] = LABEL_SCORE[i] if i < 20 else 0.0      # unknown label: std::map::operator[] yields 0 -> w = 2
    p.emulate_matp_shadow = 1
    p.skip_empty_target = 1 if variant == 3 else 0
    p.use_imu_blend = 0 if variant == 3 else 1
    p.imu_rpy_weight, p.rotation_tol, p.z_tol = 0.1, 1000.0, 1000.0
    return p


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(_HERE, "liblisreg_oracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        vp, fp, ip = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int)
        L.orc_pose_to_matrix.argtypes = [fp, fp]
        L.orc_kdtree_build.restype = vp
        L.orc_kdtree_build.argtypes = [fp, C.c_int, C.c_int]
        L.orc_kdtree_free.argtypes = [vp]
        L.orc_kdtree_knn.argtypes = [vp, fp, C.c_int, ip, fp]
        L.orc_bruteforce_knn.argtypes = [fp, C.c_int, fp, C.c_int, ip, fp]
        L.orc_eigen_sym.argtypes = [fp, C.c_int, fp, fp]
        L.orc_lstsq5x3.argtypes = [fp, fp, fp]
        L.orc_solve6.argtypes = [fp, fp, fp]
        L.orc_inv6.argtypes = [fp, fp]
        L.orc_corner_coeff.argtypes = [fp, fp, C.c_float, C.POINTER(Params), fp]
        L.orc_surf_coeff.argtypes = [fp, fp, C.c_float, C.POINTER(Params), fp]
        L.orc_jacobian_row.argtypes = [fp, fp, fp, fp, fp]
        L.orc_transform_update.argtypes = [C.POINTER(Params), C.POINTER(Imu), fp]
        L.orc_align.argtypes = [vp, C.c_int, vp, C.c_int, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int,
                                C.POINTER(Params), C.POINTER(Imu), fp, ip, C.POINTER(Stats), fp, C.c_int,
                                C.c_int, C.c_int]
        L.orc_stage_coeffs.argtypes = [C.c_int, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.POINTER(Params), fp,
                                       C.POINTER(C.c_ubyte), fp]
        L.orc_voxel_grid.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_float, vp, ip]
        L.orc_transform_cloud.argtypes = [vp, C.c_int, C.c_int, C.c_int, fp, vp]
        L.orc_transform_cloud.restype = None
        L.orc_extract_features.argtypes = [vp, C.c_int, C.c_int, C.POINTER(FeatureParams), ip, ip, ip, ip, ip, ip]
        L.orc_extract_features.restype = None
        L.orc_semantic_classes.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_uint), C.POINTER(C.c_ubyte)]
        L.orc_semantic_classes.restype = None
        L.orc_deskew_points.argtypes = [vp, C.c_int, C.POINTER(Deskew), ip, C.c_int, fp]
        L.orc_deskew_points.restype = None
        dp = C.POINTER(C.c_double)
        L.orc_cloud_bounds.argtypes = [vp, C.c_int, C.c_int, dp]
        L.orc_cloud_bounds.restype = None
        L.orc_bbx_filter.argtypes = [vp, C.c_int, C.c_int, dp, C.c_int, ip, ip]
        L.orc_bbx_filter.restype = None
        L.orc_dynamic_filter.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, ip, ip]
        L.orc_nearest.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, C.c_float, ip, fp]
        L.orc_nearest.restype = None
        L.orc_umeyama.argtypes = [fp, fp, C.c_int, C.c_int, fp]
        L.orc_umeyama.restype = None
        L.orc_icp_align.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, C.POINTER(IcpParams), fp, C.c_int, C.POINTER(IcpResult)]
        L.orc_icp_align.restype = None
        L.orc_icp_gn.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, C.c_uint, C.c_float, fp, C.c_int, C.POINTER(IcpGnResult)]
        L.orc_icp_gn.restype = None
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _vp(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None and len(a) else None


def align(tgt_corner, tgt_surf, src_corner, src_surf, T_init, params: Params, imu: Imu | None = None,
          degenerate_in: int = 0, fmt: int = 1, n_threads: int = 1, use_kdtree: bool = True, max_trace: int = 64):
    """Runs orc_align on PCL-struct arrays (synth.PCL_DTYPE).  Returns (T, stats dict, trace[n,56])."""
    L = lib()
    T = np.array(T_init, np.float32).copy()
    deg = C.c_int(degenerate_in)
    st = Stats()
    trace = np.zeros((max_trace, TRACE_STRIDE), np.float32)
    arrs = [np.ascontiguousarray(a) for a in (tgt_corner, tgt_surf, src_corner, src_surf)]
    stride = arrs[0].dtype.itemsize
    L.orc_align(_vp(arrs[0]), len(arrs[0]), _vp(arrs[1]), len(arrs[1]), _vp(arrs[2]), len(arrs[2]),
                _vp(arrs[3]), len(arrs[3]), stride, fmt, C.byref(params), C.byref(imu) if imu else None,
                _fp(T), C.byref(deg), C.byref(st), _fp(trace), max_trace, n_threads, 1 if use_kdtree else 0)
    bound = params.fixed_iters if params.fixed_iters > 0 else params.max_iters
    n_rec = min(max_trace, min(st.iters + 1, bound))
    stats = dict(iters=st.iters, deltaR=st.deltaR, deltaT=st.deltaT, degenerate=st.degenerate,
                 n_corr_last=st.n_corr_last, status=st.status)
    return T, stats, trace[:n_rec]


def stage_coeffs(kind, tgt, src, T, params: Params, fmt: int = 1):
    L = lib()
    tgt = np.ascontiguousarray(tgt); src = np.ascontiguousarray(src)
    flags = np.zeros(len(src), np.uint8)
    coeffs = np.zeros((len(src), 4), np.float32)
    Tf = np.array(T, np.float32)
    L.orc_stage_coeffs(kind, _vp(tgt), len(tgt), _vp(src), len(src), tgt.dtype.itemsize, fmt, C.byref(params),
                       _fp(Tf), flags.ctypes.data_as(C.POINTER(C.c_ubyte)), _fp(coeffs))
    return flags.astype(bool), coeffs


def voxel_grid(cloud, leaf: float, fmt: int = 1):
    """orc_voxel_grid on a PCL-struct array.  Returns (status, downsampled array)."""
    L = lib()
    cloud = np.ascontiguousarray(cloud)
    out = np.zeros_like(cloud)
    n_out = C.c_int(0)
    rc = L.orc_voxel_grid(_vp(cloud), len(cloud), cloud.dtype.itemsize, fmt, leaf, out.ctypes.data_as(C.c_void_p), C.byref(n_out))
    return rc, out[: n_out.value]


def transform_cloud(cloud, T, fmt: int = 1):
    L = lib()
    cloud = np.ascontiguousarray(cloud)
    out = np.zeros_like(cloud)
    Tf = np.array(T, np.float32)
    L.orc_transform_cloud(_vp(cloud), len(cloud), cloud.dtype.itemsize, fmt, _fp(Tf), out.ctypes.data_as(C.c_void_p))
    return out


def extract_features(cloud, params: "FeatureParams"):
    """orc_extract_features on a PointXYZIRT struct array.  Returns dict of index arrays into `cloud`."""
    L = lib()
    cloud = np.ascontiguousarray(cloud)
    hw = params.n_scan * params.horizon_scan
    bufs = [np.zeros(hw + 16, np.int32) for _ in range(5)]
    counts = np.zeros(5, np.int32)
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
    L.orc_extract_features(_vp(cloud), len(cloud), cloud.dtype.itemsize, C.byref(params), *[ip(b) for b in bufs], ip(counts))
    names = ["deskewed", "corner", "surface", "corner_sharp", "surface_sharp"]
    return {k: bufs[i][: counts[i]].copy() for i, k in enumerate(names)}


# config/label.yaml:177-196 `using_label` (label 0 has no entry -> 0 -> outlier)
USING_LABEL = [0, 10, 10, 10, 10, 10, 10, 10, 10, 40, 40, 40, 70, 50, 50, 70, 81, 70, 81, 81] + [0] * 12


def semantic_split(cloud, using_label=None):
    """Returns the five clouds (dynamic, ground, building, pole, outlier) as the stable partition by orc_semantic_classes."""
    L = lib()
    cloud = np.ascontiguousarray(cloud)
    m = (C.c_uint * 32)(*(using_label or USING_LABEL))
    cls = np.zeros(len(cloud), np.uint8)
    L.orc_semantic_classes(_vp(cloud), len(cloud), cloud.dtype.itemsize, m, cls.ctypes.data_as(C.POINTER(C.c_ubyte)))
    return [cloud[cls == k] for k in range(5)]


FLT_MAX = 3.4028234663852886e38


def cloud_bounds(cloud):
    cloud = np.ascontiguousarray(cloud)
    b = (C.c_double * 6)()
    lib().orc_cloud_bounds(_vp(cloud), len(cloud), cloud.dtype.itemsize, b)
    return np.array(list(b))


def bbx_filter(cloud, bounds, delete_box=False):
    cloud = np.ascontiguousarray(cloud)
    keep = np.zeros(max(len(cloud), 1), np.int32)
    m = C.c_int(0)
    b = (C.c_double * 6)(*[float(v) for v in bounds])
    lib().orc_bbx_filter(_vp(cloud), len(cloud), cloud.dtype.itemsize, b, 1 if delete_box else 0,
                         keep.ctypes.data_as(C.POINTER(C.c_int)), C.byref(m))
    return cloud[keep[: m.value]]


def dynamic_filter(map_cloud, cloud, center_radius, dist_thre_min=FLT_MAX, dist_thre_max=FLT_MAX, near_dist_thre=0.0):
    """(filtered cloud, applied).  map_cloud and cloud must share a dtype (one stride)."""
    cloud = np.ascontiguousarray(cloud)
    map_cloud = np.ascontiguousarray(map_cloud)
    assert map_cloud.dtype == cloud.dtype
    keep = np.zeros(max(len(cloud), 1), np.int32)
    m = C.c_int(0)
    rc = lib().orc_dynamic_filter(_vp(map_cloud), len(map_cloud), _vp(cloud), len(cloud), cloud.dtype.itemsize, center_radius,
                                  dist_thre_min, dist_thre_max, near_dist_thre, keep.ctypes.data_as(C.POINTER(C.c_int)),
                                  C.byref(m))
    return cloud[keep[: m.value]], bool(rc)


def nearest(map_cloud, query, max_dist=1e18):
    map_cloud = np.ascontiguousarray(map_cloud)
    query = np.ascontiguousarray(query)
    assert map_cloud.dtype == query.dtype
    idx = np.zeros(max(len(query), 1), np.int32)
    sqd = np.zeros(max(len(query), 1), np.float32)
    lib().orc_nearest(_vp(map_cloud), len(map_cloud), _vp(query), len(query), query.dtype.itemsize, max_dist,
                      idx.ctypes.data_as(C.POINTER(C.c_int)), _fp(sqd))
    return idx[: len(query)], sqd[: len(query)]


def icp_default_params(kind: int = 0) -> "IcpParams":
    """subMapOptmizationNode.cpp:2765-2769 (kind 0) / :1445-1449 (kind 1)"""
    p = IcpParams()
    p.max_corr_dist, p.max_iters, p.transformation_epsilon, p.euclidean_fitness_epsilon = \
        ((10.0, 30, 1e-4, 1e-4) if kind == 0 else (0.2, 50, 1e-5, 1e-5))
    p.prev_mse = np.finfo(np.float64).max
    return p


def umeyama(src_xyz, dst_xyz, float_sums=False):
    src = np.ascontiguousarray(src_xyz, np.float32); dst = np.ascontiguousarray(dst_xyz, np.float32)
    T = np.zeros(16, np.float32)
    lib().orc_umeyama(_fp(src), _fp(dst), len(src), 1 if float_sums else 0, _fp(T))
    return T.reshape(4, 4)


def icp_align(target, source, params: "IcpParams", guess=None, float_sums=False):
    """float_sums=True: Eigen-like sequential float means (order-dependent); False: double means (the GPU parity target)."""
    target = np.ascontiguousarray(target); source = np.ascontiguousarray(source)
    assert target.dtype == source.dtype
    res = IcpResult()
    g = None if guess is None else _fp(np.ascontiguousarray(guess, np.float32).ravel())
    lib().orc_icp_align(_vp(target), len(target), _vp(source), len(source), source.dtype.itemsize, C.byref(params), g, 1 if float_sums else 0,
                        C.byref(res))
    return dict(T=np.array(list(res.final_transform), np.float32).reshape(4, 4), converged=bool(res.converged), iters=res.iters,
                state=res.state, n_corr_last=res.n_corr_last, fitness=res.fitness, prev_mse=res.prev_mse)


def icp_gn_match(target, source, max_iterations, max_correspond_distance, predict_pose, float_sums=False):
    """OptimizedICPGN::Match + GetFitnessScore (registration.cpp:19-115)"""
    target = np.ascontiguousarray(target); source = np.ascontiguousarray(source)
    assert target.dtype == source.dtype
    res = IcpGnResult()
    g = np.ascontiguousarray(predict_pose, np.float32).ravel()
    lib().orc_icp_gn(_vp(target), len(target), _vp(source), len(source), source.dtype.itemsize, max_iterations,
                     max_correspond_distance, _fp(g), 1 if float_sums else 0, C.byref(res))
    return dict(T=np.array(list(res.final_transform), np.float32).reshape(4, 4), steps_applied=res.steps_applied,
                n_corr_last=res.n_corr_last, fitness=res.fitness)


def deskew_points(cloud, deskew: "Deskew", idx):
    """orc_deskew_points: de-skewed xyz [m,3] of the pixel-owning points idx (input indices) of a PointXYZIRT array"""
    cloud = np.ascontiguousarray(cloud)
    idx = np.ascontiguousarray(idx, np.int32)
    out = np.zeros((max(len(idx), 1), 3), np.float32)
    lib().orc_deskew_points(_vp(cloud), cloud.dtype.itemsize, C.byref(deskew), idx.ctypes.data_as(C.POINTER(C.c_int)), len(idx), _fp(out))
    return out[: len(idx)]
