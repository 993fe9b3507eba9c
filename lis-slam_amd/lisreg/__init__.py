"""lisreg — thin ctypes binding of liblisreg.so (include/lisreg.h), used by tests/, bench.py and smoke().

The product is the C-ABI shared library built from lis-slam_amd/csrc (hand-written HIP for gfx950); this module
only marshals numpy arrays of PCL point structs (the reference's host layout, src/include/common.h:9,25-35)
and device pointers across that boundary.  It has NO compute of its own and NO CPU fallback: if the shared
library is missing or no HIP device is visible, construction fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(_PKG, "lib", "liblisreg.so")
CSRC = os.path.join(_PKG, "csrc")

OK, NOT_ENOUGH_FEATURES, TOO_FEW_CORRESPONDENCES, LEAF_TOO_SMALL = 0, 1, 2, 3
ERR_ARG, ERR_HIP, ERR_NO_TARGET, ERR_NOMEM, ERR_COMM = -1, -2, -3, -4, -5
FMT_XYZI, FMT_XYZIL, FMT_DEVICE, FMT_XYZIRT, FMT_DEVICE_XYZI = 0, 1, 2, 3, 4
VARIANT_ODOM, VARIANT_KEYFRAME, VARIANT_SUBMAP = 1, 2, 3
TRACE_STRIDE = 56
RESULT_SIZE = 12

# every symbol include/lisreg.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "lisreg_device_count", "lisreg_create", "lisreg_destroy", "lisreg_last_error", "lisreg_set_stream",
    "lisreg_get_stream", "lisreg_default_params", "lisreg_set_target", "lisreg_set_target_slot",
    "lisreg_target_from_classes", "lisreg_align", "lisreg_align_batch", "lisreg_batch_prepare", "lisreg_batch_run",
    "lisreg_batch_fetch", "lisreg_stage_host_items", "lisreg_upload_cloud", "lisreg_concat_device", "lisreg_batch_result_device", "lisreg_set_option", "lisreg_get_option", "lisreg_get_counters", "lisreg_get_neighbors", "lisreg_test_fit_models", "lisreg_get_target_index", "lisreg_get_target_graph", "lisreg_get_target_cell_rows", "lisreg_keyframes_reset", "lisreg_keyframes_push", "lisreg_keyframes_target", "lisreg_get_trace",
    "lisreg_set_profiling", "lisreg_get_timing", "lisreg_pose_to_matrix", "lisreg_transform_update",
    "lisreg_comm_unique_id", "lisreg_comm_init", "lisreg_gather_results", "lisreg_comm_destroy",
    "lisreg_voxel_downsample", "lisreg_voxel_downsample_multi", "lisreg_transform_cloud",
    "lisreg_extract_features", "lisreg_extract_features_deskew", "lisreg_extract_features_batch", "lisreg_default_feature_params", "lisreg_semantic_split",
    "lisreg_map_index_set", "lisreg_map_index_set_batch", "lisreg_nearest", "lisreg_dynamic_filter", "lisreg_bbx_filter", "lisreg_cloud_bounds",
    "lisreg_localmap_default_params", "lisreg_localmap_reset", "lisreg_localmap_insert", "lisreg_localmap_extract",
    "lisreg_localmap_get", "lisreg_predict_pose", "lisreg_guess_state_init", "lisreg_update_initial_guess", "lisreg_submap_insert", "lisreg_submap_extract", "lisreg_submap_crop_boxes",
    "lisreg_icp_default_params", "lisreg_icp_align", "lisreg_icp_align_batch", "lisreg_icp_gn_match",
]


ICP_NOT_CONVERGED, ICP_ITERATIONS, ICP_TRANSFORM, ICP_ABS_MSE, ICP_REL_MSE, ICP_NO_CORRESPONDENCES = range(6)


class IcpParams(C.Structure):
    _fields_ = [("max_corr_dist", C.c_double), ("max_iters", C.c_int), ("reserved", C.c_int),
                ("transformation_epsilon", C.c_double), ("euclidean_fitness_epsilon", C.c_double), ("prev_mse", C.c_double)]


class IcpResult(C.Structure):
    _fields_ = [("final_transform", C.c_float * 16), ("converged", C.c_int), ("iters", C.c_int), ("state", C.c_int),
                ("n_corr_last", C.c_int), ("fitness", C.c_double), ("prev_mse", C.c_double)]

    def as_dict(self):
        return dict(T=np.array(list(self.final_transform), np.float32).reshape(4, 4), converged=bool(self.converged),
                    iters=self.iters, state=self.state, n_corr_last=self.n_corr_last, fitness=self.fitness, prev_mse=self.prev_mse)


class GuessInput(C.Structure):
    _fields_ = [("odom_available", C.c_int), ("imu_available", C.c_int),
                ("imu_roll_init", C.c_float), ("imu_pitch_init", C.c_float), ("imu_yaw_init", C.c_float),
                ("initial_guess_x", C.c_float), ("initial_guess_y", C.c_float), ("initial_guess_z", C.c_float),
                ("initial_guess_roll", C.c_float), ("initial_guess_pitch", C.c_float), ("initial_guess_yaw", C.c_float)]


class GuessState(C.Structure):
    _fields_ = [("first_trans_available", C.c_int), ("last_imu_pre_trans_available", C.c_int), ("first", C.c_int), ("reserved", C.c_int),
                ("last_imu_transformation", C.c_float * 12), ("last_imu_pre_transformation", C.c_float * 12),
                ("last_transform_tobe_mapped", C.c_float * 6)]


class IcpItem(C.Structure):
    _fields_ = [("source", C.c_void_p), ("n", C.c_int), ("slot", C.c_int), ("guess", C.POINTER(C.c_float))]


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


class Stats(C.Structure):
    _fields_ = [("iters", C.c_int), ("deltaR", C.c_float), ("deltaT", C.c_float), ("degenerate", C.c_int),
                ("n_corr_last", C.c_int), ("status", C.c_int)]


class Item(C.Structure):
    _fields_ = [("src_corner", C.c_void_p), ("n_corner", C.c_int), ("src_surf", C.c_void_p), ("n_surf", C.c_int),
                ("stride_bytes", C.c_int), ("fmt", C.c_int), ("target", C.c_int), ("degenerate_in", C.c_int),
                ("imu", Imu)]


class LocalMapParams(C.Structure):
    _fields_ = [("max_num_pts", C.c_int), ("dynamic_removal_on", C.c_int), ("dynamic_removal_center_radius", C.c_float),
                ("dynamic_dist_thre_min", C.c_float), ("dynamic_dist_thre_max", C.c_float), ("near_dist_thre", C.c_float),
                ("leaf", C.c_float * 5), ("crop_box", C.c_float * 6), ("crop_pad", C.c_float)]


class KeyframesInfo(C.Structure):
    _fields_ = [("n_keyframes", C.c_int), ("n_target_corner", C.c_int), ("n_target_surf", C.c_int)]


class LocalMapInfo(C.Structure):
    _fields_ = [("n", C.c_int * 5), ("feature_point_num", C.c_int), ("bound", C.c_double * 6), ("crop", C.c_double * 6),
                ("n_target_corner", C.c_int), ("n_target_surf", C.c_int)]


class SubmapInfo(C.Structure):
    _fields_ = [("n", C.c_int * 5), ("feature_point_num", C.c_int), ("local_bound", C.c_double * 6), ("bound", C.c_double * 6)]


class SubmapExtractOut(C.Structure):
    _fields_ = [("isect", C.c_double * 6), ("isect_local", C.c_double * 6), ("n_target_corner", C.c_int), ("n_target_surf", C.c_int),
                ("src_corner", C.c_void_p), ("n_src_corner", C.c_int), ("src_surf", C.c_void_p), ("n_src_surf", C.c_int)]


LOCALMAP_CLASSES = ("dynamic", "pole", "ground", "building", "outlier")      # class order of localMap_t (subMap.h:742-753)


class FeatureParams(C.Structure):
    _fields_ = [("n_scan", C.c_int), ("horizon_scan", C.c_int), ("downsample_rate", C.c_int), ("min_range", C.c_float),
                ("max_range", C.c_float), ("edge_threshold", C.c_float), ("surf_threshold", C.c_float)]


class FeatureOut(C.Structure):
    _fields_ = [(f, t) for name in ("deskewed", "corner", "surface", "corner_sharp", "surface_sharp")
                for f, t in ((name, C.c_void_p), ("cap_" + name, C.c_int), ("n_" + name, C.c_int))]


class SemanticOut(C.Structure):
    _fields_ = [("cloud", C.c_void_p * 5), ("cap", C.c_int * 5), ("n", C.c_int * 5)]


class LisregError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"lisreg error {code}: {msg}")
        self.code = code


def build(verbose: bool = False) -> str:
    """Compile liblisreg.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
    subprocess.check_call(["make", "-C", CSRC] + ([] if verbose else ["-s"]))
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(f"{LIB_PATH} not built — run `make -C {CSRC}` (there is no fallback path)")
        L = C.CDLL(LIB_PATH)
        vp, fp, ip = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int)
        L.lisreg_device_count.restype = C.c_int
        L.lisreg_create.argtypes = [C.c_int, C.POINTER(vp)]
        L.lisreg_destroy.argtypes = [vp]
        L.lisreg_destroy.restype = None
        L.lisreg_last_error.argtypes = [vp]
        L.lisreg_last_error.restype = C.c_char_p
        L.lisreg_set_stream.argtypes = [vp, vp]
        L.lisreg_get_stream.argtypes = [vp]
        L.lisreg_get_stream.restype = vp
        L.lisreg_default_params.argtypes = [C.c_int, C.POINTER(Params)]
        L.lisreg_set_target.argtypes = [vp, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int]
        L.lisreg_set_target_slot.argtypes = [vp, C.c_int, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int]
        L.lisreg_target_from_classes.argtypes = [vp, C.c_int, vp, C.c_int, vp, C.c_int, vp, C.c_int, vp, C.c_int,
                                                 C.c_int, C.c_int]
        L.lisreg_align.argtypes = [vp, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.POINTER(Params),
                                   C.POINTER(Imu), fp, C.POINTER(Stats)]
        L.lisreg_align_batch.argtypes = [vp, C.c_int, C.POINTER(Item), C.POINTER(Params), fp, C.POINTER(Stats)]
        L.lisreg_batch_prepare.argtypes = [vp, C.c_int, C.POINTER(Item), C.POINTER(Params), fp]
        L.lisreg_stage_host_items.argtypes = [vp, C.c_int, C.POINTER(Item), C.POINTER(Item)]
        L.lisreg_batch_run.argtypes = [vp]
        L.lisreg_batch_fetch.argtypes = [vp, fp, C.POINTER(Stats)]
        L.lisreg_batch_result_device.argtypes = [vp]
        L.lisreg_batch_result_device.restype = vp
        L.lisreg_set_option.argtypes = [vp, C.c_char_p, C.c_int]
        L.lisreg_upload_cloud.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp]
        L.lisreg_concat_device.argtypes = [vp, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int), vp, C.POINTER(C.c_int)]
        L.lisreg_get_option.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int)]
        L.lisreg_get_neighbors.argtypes = [vp, C.POINTER(C.c_int), C.c_int]
        if hasattr(L, "lisreg_test_fit_models"): L.lisreg_test_fit_models.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(Params), C.c_int, C.POINTER(C.c_float)]
        L.lisreg_get_target_index.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_float), C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.lisreg_get_target_graph.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int]
        L.lisreg_get_target_cell_rows.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int,
                                                  C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int]
        L.lisreg_get_counters.argtypes = [vp, C.POINTER(C.c_ulonglong), C.c_int]
        L.lisreg_get_trace.argtypes = [vp, fp, C.c_int]
        L.lisreg_set_profiling.argtypes = [vp, C.c_int]
        L.lisreg_get_timing.argtypes = [vp, C.POINTER(C.c_double)]
        L.lisreg_pose_to_matrix.argtypes = [fp, fp]
        L.lisreg_pose_to_matrix.restype = None
        L.lisreg_transform_update.argtypes = [C.POINTER(Params), C.POINTER(Imu), fp]
        L.lisreg_transform_update.restype = None
        L.lisreg_comm_unique_id.argtypes = [C.POINTER(C.c_ubyte)]
        L.lisreg_comm_init.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_ubyte)]
        L.lisreg_gather_results.argtypes = [vp, vp, C.c_int, vp]
        L.lisreg_comm_destroy.argtypes = [vp]
        L.lisreg_comm_destroy.restype = None
        L.lisreg_voxel_downsample_multi.argtypes = [vp, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_float), C.c_int,
                                                    C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.lisreg_voxel_downsample.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_float, vp, C.c_int, ip]
        L.lisreg_transform_cloud.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, fp, vp]
        L.lisreg_extract_features.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.POINTER(FeatureParams), C.POINTER(FeatureOut)]
        L.lisreg_default_feature_params.argtypes = [C.POINTER(FeatureParams)]
        L.lisreg_extract_features_batch.argtypes = [vp, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(FeatureParams),
                                                    C.POINTER(FeatureOut)]
        L.lisreg_extract_features_deskew.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.POINTER(FeatureParams), C.POINTER(Deskew),
                                                     C.POINTER(FeatureOut)]
        L.lisreg_semantic_split.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint32), C.POINTER(SemanticOut)]
        ip, dp = C.POINTER(C.c_int), C.POINTER(C.c_double)
        L.lisreg_map_index_set.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int]
        L.lisreg_map_index_set_batch.argtypes = [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int, C.c_int]
        L.lisreg_nearest.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_float, vp, vp]
        L.lisreg_dynamic_filter.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float,
                                            C.c_float, vp, ip]
        L.lisreg_bbx_filter.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, dp, C.c_int, vp, ip]
        L.lisreg_cloud_bounds.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, dp]
        L.lisreg_localmap_default_params.argtypes = [C.POINTER(LocalMapParams)]
        L.lisreg_localmap_reset.argtypes = [vp, C.c_int]
        L.lisreg_keyframes_reset.argtypes = [vp, C.c_int]
        L.lisreg_keyframes_push.argtypes = [vp, C.c_int, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_int, C.POINTER(KeyframesInfo)]
        L.lisreg_keyframes_target.argtypes = [vp, C.c_int, C.c_float, C.c_float, C.c_int, C.POINTER(KeyframesInfo)]
        L.lisreg_localmap_insert.argtypes = [vp, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int, C.c_int, fp,
                                             C.POINTER(LocalMapParams), C.POINTER(LocalMapInfo)]
        L.lisreg_localmap_extract.argtypes = [vp, C.c_int, fp, C.POINTER(LocalMapParams), C.c_int, C.POINTER(LocalMapInfo)]
        L.lisreg_localmap_get.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, ip]
        L.lisreg_submap_insert.argtypes = [vp, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int, C.c_int, fp, fp,
                                           C.POINTER(LocalMapParams), C.POINTER(SubmapInfo)]
        L.lisreg_submap_extract.argtypes = [vp, C.c_int, C.c_int, fp, fp, C.c_float, C.c_float, C.c_float, C.c_int, C.POINTER(SubmapExtractOut)]
        L.lisreg_submap_crop_boxes.argtypes = [C.POINTER(C.c_double), fp, C.POINTER(C.c_double), fp, C.c_float, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.lisreg_submap_crop_boxes.restype = None
        L.lisreg_predict_pose.argtypes = [fp, fp, fp]
        L.lisreg_predict_pose.restype = None
        L.lisreg_guess_state_init.argtypes = [C.POINTER(GuessState)]
        L.lisreg_guess_state_init.restype = None
        L.lisreg_update_initial_guess.argtypes = [C.c_int, C.c_int, C.POINTER(GuessInput), C.POINTER(GuessState), fp, fp]
        L.lisreg_update_initial_guess.restype = None
        L.lisreg_icp_default_params.argtypes = [C.c_int, C.POINTER(IcpParams)]
        L.lisreg_icp_gn_match.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_uint, C.c_float, fp, C.POINTER(IcpGnResult), vp]
        L.lisreg_icp_align.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.POINTER(IcpParams), fp, C.POINTER(IcpResult), vp]
        L.lisreg_icp_align_batch.argtypes = [vp, C.POINTER(IcpItem), C.c_int, C.c_int, C.c_int, C.POINTER(IcpParams), C.c_int, C.POINTER(IcpResult)]
        _lib = L
    return _lib


def localmap_default_params() -> LocalMapParams:
    p = LocalMapParams()
    if lib().lisreg_localmap_default_params(C.byref(p)):
        raise LisregError(ERR_ARG, "lisreg_localmap_default_params")
    return p


def predict_pose(T_last, T_cur) -> np.ndarray:
    a = np.ascontiguousarray(T_last, np.float32); b = np.ascontiguousarray(T_cur, np.float32)
    out = np.zeros(6, np.float32)
    f = lambda x: x.ctypes.data_as(C.POINTER(C.c_float))
    lib().lisreg_predict_pose(f(a), f(b), f(out))
    return out


class InitialGuess:
    """updateInitialGuess with its function-local statics (lisreg_update_initial_guess): variant 0 = odomEstimationNode.cpp:297-419,
    1 = subMapOptmizationNode.cpp:896-1032.  update(T, ...) returns (T_new, transPredictionMapped or None)."""

    def __init__(self, variant: int = 0, use_imu_heading_initialization: bool = False):
        self.variant, self.heading = variant, use_imu_heading_initialization
        self.state = GuessState()
        lib().lisreg_guess_state_init(C.byref(self.state))

    def update(self, T, odom_available=False, imu_available=False, imu_rpy=(0.0, 0.0, 0.0), initial_guess=(0.0,) * 6):
        """initial_guess = (x, y, z, roll, pitch, yaw) of cloudInfo.initialGuess*"""
        inp = GuessInput(1 if odom_available else 0, 1 if imu_available else 0, *[float(v) for v in imu_rpy], *[float(v) for v in initial_guess])
        T = np.array(T, np.float32)
        # "not assigned" is told by a bit pattern the library cannot produce (a NaN with a payload of its own), not by NaN-ness: a prediction
        # that really IS NaN (a NaN pose propagated through the increment) comes back as what it is
        sentinel = np.uint32(0x7FC0DEAD)
        pred = np.full(6, sentinel, np.uint32).view(np.float32)
        f = lambda x: x.ctypes.data_as(C.POINTER(C.c_float))
        lib().lisreg_update_initial_guess(self.variant, 1 if self.heading else 0, C.byref(inp), C.byref(self.state), f(T), f(pred))
        return T, (None if (pred.view(np.uint32) == sentinel).all() else pred)


def _info_dict(info: LocalMapInfo) -> dict:
    return dict(n=list(info.n), feature_point_num=info.feature_point_num, bound=np.array(list(info.bound)),
                crop=np.array(list(info.crop)), n_target_corner=info.n_target_corner, n_target_surf=info.n_target_surf)


def icp_default_params(kind: int = 0) -> IcpParams:
    p = IcpParams()
    if lib().lisreg_icp_default_params(kind, C.byref(p)):
        raise LisregError(ERR_ARG, "lisreg_icp_default_params")
    return p


def default_params(variant: int = VARIANT_ODOM) -> Params:
    p = Params()
    rc = lib().lisreg_default_params(variant, C.byref(p))
    if rc:
        raise LisregError(rc, "lisreg_default_params")
    return p


def pose_to_matrix(T) -> np.ndarray:
    T = np.ascontiguousarray(T, np.float32)
    M = np.zeros(12, np.float32)
    lib().lisreg_pose_to_matrix(T.ctypes.data_as(C.POINTER(C.c_float)), M.ctypes.data_as(C.POINTER(C.c_float)))
    return M.reshape(3, 4)


def transform_update(params: Params, imu: Imu | None, T) -> np.ndarray:
    T = np.array(T, np.float32)
    lib().lisreg_transform_update(C.byref(params), C.byref(imu) if imu is not None else None,
                                  T.ctypes.data_as(C.POINTER(C.c_float)))
    return T


def _vp(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None and len(a) else None


def _fmt_of(cloud) -> int:
    return FMT_XYZIL if (cloud.dtype.names and "label" in cloud.dtype.names) else FMT_XYZI


def stats_dict(s: Stats) -> dict:
    return dict(iters=s.iters, deltaR=s.deltaR, deltaT=s.deltaT, degenerate=s.degenerate,
                n_corr_last=s.n_corr_last, status=s.status)


class Context:
    """One lisreg_ctx (single-threaded; one per caller, like the three reference node classes)."""

    def __init__(self, device: int = 0):
        self._L = lib()
        h = C.c_void_p()
        rc = self._L.lisreg_create(device, C.byref(h))
        if rc:
            raise LisregError(rc, self._L.lisreg_last_error(None).decode())
        self._h = h
        self._keep = []

    def close(self):
        if self._h:
            self._L.lisreg_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, allow=(OK,)):
        if rc not in allow:
            raise LisregError(rc, self._L.lisreg_last_error(self._h).decode())
        return rc

    @property
    def stream(self) -> int:
        return self._L.lisreg_get_stream(self._h) or 0

    def set_stream(self, hip_stream: int | None):
        self._chk(self._L.lisreg_set_stream(self._h, C.c_void_p(hip_stream) if hip_stream else None))

    def set_option(self, name: str, value: int):
        self._chk(self._L.lisreg_set_option(self._h, name.encode(), int(value)))

    def get_option(self, name: str) -> int:
        v = C.c_int(0)
        self._chk(self._L.lisreg_get_option(self._h, name.encode(), C.byref(v)))
        return v.value

    def neighbors(self, n_elems: int) -> np.ndarray:
        """[6, n_elems]: rows 0-4 original target indices of every source point's neighbours in the last GN iteration run
        (-1: none), row 5 = 1 where the point contributed a correspondence."""
        out = np.full((6, n_elems), -1, np.int32)
        self._chk(self._L.lisreg_get_neighbors(self._h, out.ctypes.data_as(C.POINTER(C.c_int)), n_elems))
        return out

    def test_fit_models(self, kind: int, neighbours: np.ndarray, queries: np.ndarray, params, exact: bool = False) -> np.ndarray:
        """Test hook: the device functions of the residual models (kind 1 = surfOptimization's body, 0 = cornerOptimization's) on
        neighbours[n, 5, 3] / queries[n, 3]; returns float32[n, 10] as include/lisreg.h describes."""
        nb = np.ascontiguousarray(neighbours, np.float32).reshape(-1, 15)
        q = np.ascontiguousarray(queries, np.float32).reshape(-1, 3)
        assert nb.shape[0] == q.shape[0]
        out = np.zeros((nb.shape[0], 10), np.float32)
        fp = C.POINTER(C.c_float)
        self._chk(self._L.lisreg_test_fit_models(self._h, int(kind), nb.shape[0], nb.ctypes.data_as(fp), q.ctypes.data_as(fp), C.byref(params),
                                                 1 if exact else 0, out.ctypes.data_as(fp)))
        return out

    def front_end(self) -> int:
        """search front-end of the prepared batch (1 cell walk, 3 k-NN graph scan, ...)."""
        return self.get_option("front_end")

    def set_profiling_paused(self, paused: bool):
        """stop (or resume) recording HIP events without discarding the ones already recorded; batch_fetch collects them."""
        self._chk(self._L.lisreg_set_profiling(self._h, 0 if paused else 1))

    # -- target ------------------------------------------------------------------------------------------
    def set_target(self, corner: np.ndarray, surf: np.ndarray, slot: int = 0):
        """corner/surf: numpy arrays of PCL point structs (itemsize = stride)."""
        corner = np.ascontiguousarray(corner); surf = np.ascontiguousarray(surf)
        self._chk(self._L.lisreg_set_target_slot(self._h, slot, _vp(corner), len(corner), _vp(surf), len(surf),
                                                 corner.dtype.itemsize, _fmt_of(corner)))

    def set_target_device(self, corner_ptr: int, n_corner: int, surf_ptr: int, n_surf: int, slot: int = 0):
        self._chk(self._L.lisreg_set_target_slot(self._h, slot, C.c_void_p(corner_ptr), n_corner,
                                                 C.c_void_p(surf_ptr), n_surf, 16, FMT_DEVICE))

    def target_from_classes(self, pole, ground, building, dynamic, slot: int = 0):
        arrs = [np.ascontiguousarray(a) for a in (pole, ground, building, dynamic)]
        self._chk(self._L.lisreg_target_from_classes(self._h, slot, _vp(arrs[0]), len(arrs[0]), _vp(arrs[1]),
                                                     len(arrs[1]), _vp(arrs[2]), len(arrs[2]), _vp(arrs[3]),
                                                     len(arrs[3]), arrs[0].dtype.itemsize, _fmt_of(arrs[0])))

    # -- single registration ---------------------------------------------------------------------------------
    def align(self, src_corner: np.ndarray, src_surf: np.ndarray, T_init, params: Params, imu: Imu | None = None):
        """lisreg_align.  Returns (T, stats dict, trace[n,56])."""
        sc = np.ascontiguousarray(src_corner); ss = np.ascontiguousarray(src_surf)
        T = np.array(T_init, np.float32).copy()
        st = Stats()
        rc = self._L.lisreg_align(self._h, _vp(sc), len(sc), _vp(ss), len(ss), sc.dtype.itemsize, _fmt_of(sc),
                                  C.byref(params), C.byref(imu) if imu is not None else None,
                                  T.ctypes.data_as(C.POINTER(C.c_float)), C.byref(st))
        self._chk(rc, allow=(OK, NOT_ENOUGH_FEATURES, TOO_FEW_CORRESPONDENCES))
        bound = params.fixed_iters if params.fixed_iters > 0 else params.max_iters
        buf = np.zeros((max(bound, 1), TRACE_STRIDE), np.float32)
        n = self._L.lisreg_get_trace(self._h, buf.ctypes.data_as(C.POINTER(C.c_float)), max(bound, 1))
        d = stats_dict(st)
        if rc == NOT_ENOUGH_FEATURES:
            d["status"] = NOT_ENOUGH_FEATURES
        return T, d, buf[:n]

    # -- batches ---------------------------------------------------------------------------------------------
    def align_batch(self, items: list[dict], T_init: np.ndarray, params: Params):
        """items: dicts with src_corner, src_surf (PCL struct arrays), optional target, degenerate_in, imu."""
        n = len(items)
        arr = (Item * max(n, 1))()
        keep = []
        for i, it in enumerate(items):
            sc = np.ascontiguousarray(it["src_corner"]); ss = np.ascontiguousarray(it["src_surf"])
            keep += [sc, ss]
            arr[i].src_corner = _vp(sc); arr[i].n_corner = len(sc)
            arr[i].src_surf = _vp(ss); arr[i].n_surf = len(ss)
            arr[i].stride_bytes = sc.dtype.itemsize; arr[i].fmt = _fmt_of(sc)
            arr[i].target = it.get("target", 0); arr[i].degenerate_in = it.get("degenerate_in", 0)
            if it.get("imu") is not None:
                arr[i].imu = it["imu"]
        T = np.ascontiguousarray(T_init, np.float32).reshape(n, 6).copy()
        st = (Stats * max(n, 1))()
        self._chk(self._L.lisreg_align_batch(self._h, n, arr, C.byref(params),
                                             T.ctypes.data_as(C.POINTER(C.c_float)), st))
        return T, [stats_dict(st[i]) for i in range(n)]

    def batch_prepare_device(self, items: list[dict], T_init: np.ndarray, params: Params):
        """items: dicts with corner_ptr, n_corner, surf_ptr, n_surf (device pointers to 16-B records), target."""
        n = len(items)
        arr = (Item * max(n, 1))()
        for i, it in enumerate(items):
            arr[i].src_corner = C.c_void_p(it["corner_ptr"]) if it["n_corner"] else None
            arr[i].n_corner = it["n_corner"]
            arr[i].src_surf = C.c_void_p(it["surf_ptr"]) if it["n_surf"] else None
            arr[i].n_surf = it["n_surf"]
            arr[i].stride_bytes = 16; arr[i].fmt = FMT_DEVICE
            arr[i].target = it.get("target", 0); arr[i].degenerate_in = it.get("degenerate_in", 0)
            if it.get("imu") is not None:
                arr[i].imu = it["imu"]
        T = np.ascontiguousarray(T_init, np.float32).reshape(n, 6)
        self._n_items = n
        self._chk(self._L.lisreg_batch_prepare(self._h, n, arr, C.byref(params),
                                               T.ctypes.data_as(C.POINTER(C.c_float))))

    def batch_run(self):
        self._chk(self._L.lisreg_batch_run(self._h))

    def batch_fetch(self):
        n = self._n_items
        T = np.zeros((n, 6), np.float32)
        st = (Stats * max(n, 1))()
        self._chk(self._L.lisreg_batch_fetch(self._h, T.ctypes.data_as(C.POINTER(C.c_float)), st))
        return T, [stats_dict(st[i]) for i in range(n)]

    @property
    def result_device_ptr(self) -> int:
        return self._L.lisreg_batch_result_device(self._h) or 0

    def counters(self) -> np.ndarray:
        out = (C.c_ulonglong * 64)()
        self._chk(self._L.lisreg_get_counters(self._h, out, 64))
        return np.array(out[:], np.int64).reshape(32, 2)

    def target_index(self, slot: int = 0, kind: int = 1) -> dict:
        """Diagnostics: the device search index of a target (see lisreg_get_target_index)."""
        dims = (C.c_int * 5)(); geom = (C.c_float * 4)()
        self._chk(self._L.lisreg_get_target_index(self._h, slot, kind, dims, geom, None, 0, None, 0))
        n, nx, ny, nz, nc = [int(v) for v in dims]
        pts = np.zeros((max(n, 1), 4), np.float32); cs = np.zeros(nc + 1, np.int32)
        self._chk(self._L.lisreg_get_target_index(self._h, slot, kind, dims, geom, pts.ctypes.data, n, cs.ctypes.data, nc + 1))
        return dict(n=n, nx=nx, ny=ny, nz=nz, n_cells=nc, origin=np.array(geom[:3], np.float32), cell=float(geom[3]),
                    sorted=pts[:n], cell_start=cs)

    def target_graph(self, slot: int = 0, kind: int = 1) -> dict:
        """Diagnostics: the k-NN graph of a target (see lisreg_get_target_graph): ids [n, k] (sorted positions, -1 padded),
        xyz [n, k, 3], rho2 [n], count [n]."""
        n = self.target_index(slot, kind)["n"]
        k = C.c_int()
        self._chk(self._L.lisreg_get_target_graph(self._h, slot, kind, C.byref(k), None, None, 0))
        rows = np.zeros((max(n, 1), k.value, 4), np.float32); meta = np.zeros((max(n, 1), 2), np.float32)
        self._chk(self._L.lisreg_get_target_graph(self._h, slot, kind, C.byref(k), rows.ctypes.data_as(C.POINTER(C.c_float)),
                                                  meta.ctypes.data_as(C.POINTER(C.c_float)), n))
        return dict(k=k.value, ids=rows[:n, :, 3].copy().view(np.int32), xyz=rows[:n, :, :3], rho2=meta[:n, 0],
                    count=meta[:n, 1].copy().view(np.int32))

    def target_cell_rows(self, slot: int = 0, kind: int = 1) -> dict:
        """Diagnostics: the cell rows of a target (see lisreg_get_target_cell_rows): table [n_cells], ids [rows, k] (sorted positions,
        -1 padded), xyz [rows, k, 3], rho2 [rows], count [rows]."""
        n_rows, k = C.c_int(), C.c_int()
        # (the first call builds the rows if need be — which re-makes the target's grid with its two-cell margin: read the geometry after it)
        self._chk(self._L.lisreg_get_target_cell_rows(self._h, slot, kind, C.byref(n_rows), C.byref(k), None, 0, None, None, 0))
        nc = self.target_index(slot, kind)["n_cells"]
        table = np.zeros(max(nc, 1), np.int32)
        rows = np.zeros((max(n_rows.value, 1), k.value, 4), np.float32); meta = np.zeros((max(n_rows.value, 1), 2), np.float32)
        self._chk(self._L.lisreg_get_target_cell_rows(self._h, slot, kind, C.byref(n_rows), C.byref(k), table.ctypes.data_as(C.POINTER(C.c_int)), nc,
                                                      rows.ctypes.data_as(C.POINTER(C.c_float)), meta.ctypes.data_as(C.POINTER(C.c_float)), n_rows.value))
        r = n_rows.value
        return dict(k=k.value, n_rows=r, table=table[:nc], ids=rows[:r, :, 3].copy().view(np.int32), xyz=rows[:r, :, :3], rho2=meta[:r, 0],
                    count=meta[:r, 1].copy().view(np.int32))

    def raw_counters(self):
        out = (C.c_ulonglong * 128)()
        self._chk(self._L.lisreg_get_counters(self._h, out, 128))
        return [int(v) for v in out]

    def wave_counters(self) -> np.ndarray:
        """Graph front-end only: per GN iteration (wavefronts with at least one lane in the cell walk, wavefronts)."""
        out = (C.c_ulonglong * 128)()
        self._chk(self._L.lisreg_get_counters(self._h, out, 128))
        raw = np.array(out[64:96], np.uint64)
        return np.stack([(raw >> np.uint64(32)).astype(np.int64), (raw & np.uint64(0xffffffff)).astype(np.int64)], 1)

    # -- §8 f-1 -------------------------------------------------------------------------------------------
    def voxel_downsample(self, cloud: np.ndarray, leaf: float):
        """pcl::VoxelGrid replacement on a PCL struct array.  Returns (status, downsampled array)."""
        cloud = np.ascontiguousarray(cloud)
        out = np.zeros_like(cloud)
        n_out = C.c_int(0)
        rc = self._L.lisreg_voxel_downsample(self._h, _vp(cloud), len(cloud), cloud.dtype.itemsize, _fmt_of(cloud), leaf,
                                             out.ctypes.data_as(C.c_void_p), len(out), C.byref(n_out))
        self._chk(rc, allow=(OK, LEAF_TOO_SMALL))
        return rc, out[: n_out.value]

    def voxel_downsample_device(self, in_ptr: int, n: int, leaf: float, out_ptr: int, capacity: int, intensity: bool = False):
        """Device records in, device records out.  intensity=True: the payload is a float (PointXYZI clouds) and is averaged;
        otherwise it is a label and the voxel takes the majority."""
        n_out = C.c_int(0)
        rc = self._L.lisreg_voxel_downsample(self._h, C.c_void_p(in_ptr), n, 16, FMT_DEVICE_XYZI if intensity else FMT_DEVICE, leaf, C.c_void_p(out_ptr),
                                             capacity, C.byref(n_out))
        self._chk(rc, allow=(OK, LEAF_TOO_SMALL))
        return rc, n_out.value

    def voxel_downsample_multi_device(self, in_ptrs, counts, leafs, out_ptrs, capacities, intensity: bool = False):
        """K device clouds through ONE launch sequence (lisreg_voxel_downsample_multi); returns the K output counts."""
        k = len(in_ptrs)
        ins = (C.c_void_p * k)(*[C.c_void_p(int(p)) if c else None for p, c in zip(in_ptrs, counts)])
        outs = (C.c_void_p * k)(*[C.c_void_p(int(p)) if c else None for p, c in zip(out_ptrs, counts)])
        cnt = (C.c_int * k)(*[int(x) for x in counts]); cap = (C.c_int * k)(*[int(x) for x in capacities])
        lf = (C.c_float * k)(*[float(x) for x in leafs]); no = (C.c_int * k)()
        self._chk(self._L.lisreg_voxel_downsample_multi(self._h, k, ins, cnt, lf, FMT_DEVICE_XYZI if intensity else FMT_DEVICE, outs, cap, no))
        return [int(x) for x in no]

    def transform_cloud(self, cloud: np.ndarray, T):
        cloud = np.ascontiguousarray(cloud)
        out = np.zeros_like(cloud)
        Tf = np.ascontiguousarray(T, np.float32)
        self._chk(self._L.lisreg_transform_cloud(self._h, _vp(cloud), len(cloud), cloud.dtype.itemsize, _fmt_of(cloud),
                                                 Tf.ctypes.data_as(C.POINTER(C.c_float)), out.ctypes.data_as(C.c_void_p)))
        return out

    def transform_cloud_device(self, in_ptr: int, n: int, T, out_ptr: int):
        Tf = np.ascontiguousarray(T, np.float32)
        self._chk(self._L.lisreg_transform_cloud(self._h, C.c_void_p(in_ptr), n, 16, FMT_DEVICE,
                                                 Tf.ctypes.data_as(C.POINTER(C.c_float)), C.c_void_p(out_ptr)))

    # -- §8 f-2 -------------------------------------------------------------------------------------------
    def extract_features(self, cloud: np.ndarray, params: "FeatureParams", deskew: "Deskew | None" = None) -> dict:
        """LaserProcessing replacement on a PointXYZIRT struct array: the five clouds of cloud_info (optionally IMU-de-skewed)."""
        cloud = np.ascontiguousarray(cloud)
        cap = params.n_scan * params.horizon_scan
        names = ("deskewed", "corner", "surface", "corner_sharp", "surface_sharp")
        bufs = {k: np.zeros(cap, cloud.dtype) for k in names}
        fo = FeatureOut()
        for k in names:
            setattr(fo, k, bufs[k].ctypes.data_as(C.c_void_p)); setattr(fo, "cap_" + k, cap)
        self._chk(self._L.lisreg_extract_features_deskew(self._h, _vp(cloud), len(cloud), cloud.dtype.itemsize, FMT_XYZIRT,
                                                         C.byref(params), C.byref(deskew) if deskew is not None else None, C.byref(fo)))
        return {k: bufs[k][: getattr(fo, "n_" + k)] for k in names}

    def extract_features_device(self, in_ptr: int, n: int, params: "FeatureParams", out_ptrs: dict, cap: int, deskew=None) -> dict:
        fo = FeatureOut()
        for k, ptr in out_ptrs.items():
            setattr(fo, k, C.c_void_p(ptr)); setattr(fo, "cap_" + k, cap)
        self._chk(self._L.lisreg_extract_features_deskew(self._h, C.c_void_p(in_ptr), n, 16, FMT_DEVICE, C.byref(params),
                                                         C.byref(deskew) if deskew is not None else None, C.byref(fo)))
        return {k: getattr(fo, "n_" + k) for k in ("deskewed", "corner", "surface", "corner_sharp", "surface_sharp")}

    def extract_features_batch_device(self, in_ptrs, counts, params: "FeatureParams", out_ptrs: list, cap: int) -> list:
        """lisreg_extract_features_batch: in_ptrs / counts per sweep (device records), out_ptrs = one dict of five device buffers
        per sweep (capacity `cap` points each).  Returns the per-sweep count dicts."""
        S = len(in_ptrs)
        ptrs = (C.c_void_p * S)(*[C.c_void_p(int(p)) if n else None for p, n in zip(in_ptrs, counts)])
        ns = (C.c_int * S)(*[int(x) for x in counts])
        fos = (FeatureOut * S)()
        for s in range(S):
            for k, ptr in out_ptrs[s].items():
                setattr(fos[s], k, C.c_void_p(ptr)); setattr(fos[s], "cap_" + k, cap)
        self._chk(self._L.lisreg_extract_features_batch(self._h, S, ptrs, ns, C.byref(params), fos))
        return [{k: getattr(fos[s], "n_" + k) for k in ("deskewed", "corner", "surface", "corner_sharp", "surface_sharp")} for s in range(S)]

    def concat_device(self, in_ptrs, counts, out_ptr: int) -> int:
        """lisreg_concat_device: K device clouds end to end into out_ptr (stream-ordered, no wait).  Returns the total count."""
        k = len(in_ptrs)
        p = (C.c_void_p * k)(*[C.c_void_p(int(x)) for x in in_ptrs])
        n = (C.c_int * k)(*[int(x) for x in counts])
        tot = C.c_int(0)
        self._chk(self._L.lisreg_concat_device(self._h, k, p, n, C.c_void_p(int(out_ptr)), C.byref(tot)))
        return tot.value

    def upload_cloud(self, cloud: np.ndarray, dev_ptr: int) -> int:
        """lisreg_upload_cloud: a host PCL struct array (x, y, z at 0 / 4 / 8; the uint16 at byte 20 — label or ring — becomes the
        payload when the dtype has one) into a device buffer of len(cloud) 16-byte records.  Returns the count."""
        cloud = np.ascontiguousarray(cloud)
        names = cloud.dtype.names or ()
        has16 = any(cloud.dtype.fields[k][1] == 20 and cloud.dtype.fields[k][0].itemsize == 2 for k in names)
        self._chk(self._L.lisreg_upload_cloud(self._h, cloud.ctypes.data_as(C.c_void_p), len(cloud), cloud.dtype.itemsize,
                                              FMT_XYZIL if has16 else FMT_XYZI, C.c_void_p(dev_ptr)))
        return len(cloud)

    def semantic_split_device(self, in_ptr: int, n: int, out_ptrs, cap: int, using_label=None) -> list:
        """categoryMapping on device records (label in the payload): out_ptrs = five device buffers of `cap` records
        (dynamic, ground, building, pole, outlier).  Returns the five counts."""
        so = SemanticOut()
        for k in range(5):
            so.cloud[k] = int(out_ptrs[k]); so.cap[k] = cap
        m = (C.c_uint32 * 32)(*using_label) if using_label is not None else None
        self._chk(self._L.lisreg_semantic_split(self._h, C.c_void_p(in_ptr), n, 16, FMT_DEVICE, m, C.byref(so)))
        return [int(so.n[k]) for k in range(5)]

    def align_device(self, corner_ptr: int, n_corner: int, surf_ptr: int, n_surf: int, T_init, params: Params, imu: Imu | None = None):
        """lisreg_align on device-resident source records.  Returns (T, stats dict)."""
        T = np.array(T_init, np.float32).copy()
        st = Stats()
        rc = self._L.lisreg_align(self._h, C.c_void_p(corner_ptr) if n_corner else None, n_corner, C.c_void_p(surf_ptr) if n_surf else None,
                                  n_surf, 16, FMT_DEVICE, C.byref(params), C.byref(imu) if imu is not None else None,
                                  T.ctypes.data_as(C.POINTER(C.c_float)), C.byref(st))
        self._chk(rc, allow=(OK, NOT_ENOUGH_FEATURES, TOO_FEW_CORRESPONDENCES))
        d = stats_dict(st)
        if rc == NOT_ENOUGH_FEATURES:
            d["status"] = NOT_ENOUGH_FEATURES
        return T, d

    def semantic_split(self, cloud: np.ndarray, using_label=None) -> list:
        """categoryMapping replacement: [dynamic, ground, building, pole, outlier] from a PointXYZIL struct array."""
        cloud = np.ascontiguousarray(cloud)
        bufs = [np.zeros(len(cloud), cloud.dtype) for _ in range(5)]
        so = SemanticOut()
        for k in range(5):
            so.cloud[k] = bufs[k].ctypes.data_as(C.c_void_p).value if len(cloud) else None
            so.cap[k] = len(cloud)
        m = (C.c_uint32 * 32)(*using_label) if using_label is not None else None
        self._chk(self._L.lisreg_semantic_split(self._h, _vp(cloud), len(cloud), cloud.dtype.itemsize, FMT_XYZIL, m, C.byref(so)))
        return [bufs[k][: so.n[k]] for k in range(5)]

    # -- the odometry node's key-frame target, device-resident ------------------------------------------------
    def keyframes_reset(self, ring_id: int = 0):
        self._chk(self._L.lisreg_keyframes_reset(self._h, ring_id))

    def keyframes_push(self, ring_id: int, corner: np.ndarray, surf: np.ndarray, pose, max_keep: int = 19) -> dict:
        corner = np.ascontiguousarray(corner); surf = np.ascontiguousarray(surf)
        T = np.ascontiguousarray(pose, np.float32)
        info = KeyframesInfo()
        self._chk(self._L.lisreg_keyframes_push(self._h, ring_id, _vp(corner), len(corner), _vp(surf), len(surf), corner.dtype.itemsize,
                                                _fmt_of(corner), T.ctypes.data_as(C.POINTER(C.c_float)), max_keep, C.byref(info)))
        return dict(n_keyframes=info.n_keyframes)

    def keyframes_push_device(self, ring_id: int, corner_ptr: int, n_corner: int, surf_ptr: int, n_surf: int, pose, max_keep: int = 19,
                              label_payload: bool = False) -> dict:
        T = np.ascontiguousarray(pose, np.float32)
        info = KeyframesInfo()
        self._chk(self._L.lisreg_keyframes_push(self._h, ring_id, C.c_void_p(corner_ptr) if n_corner else None, n_corner,
                                                C.c_void_p(surf_ptr) if n_surf else None, n_surf, 16, FMT_DEVICE if label_payload else FMT_DEVICE_XYZI,
                                                T.ctypes.data_as(C.POINTER(C.c_float)), max_keep, C.byref(info)))
        return dict(n_keyframes=info.n_keyframes)

    def keyframes_target(self, ring_id: int, corner_leaf: float, surf_leaf: float, target_slot: int = 0) -> dict:
        info = KeyframesInfo()
        self._chk(self._L.lisreg_keyframes_target(self._h, ring_id, corner_leaf, surf_leaf, target_slot, C.byref(info)))
        return dict(n_keyframes=info.n_keyframes, n_target_corner=info.n_target_corner, n_target_surf=info.n_target_surf)

    # -- §8 f-3: device-resident sliding local map -----------------------------------------------------------
    def localmap_reset(self, map_id: int = 0):
        self._chk(self._L.lisreg_localmap_reset(self._h, map_id))

    def localmap_insert(self, map_id: int, clouds, pose, params: LocalMapParams) -> dict:
        """clouds: five PCL PointXYZIL struct arrays in LOCALMAP_CLASSES order (sensor frame, un-downsampled)."""
        arrs = [np.ascontiguousarray(a) for a in clouds]
        ptrs = (C.c_void_p * 5)(*[a.ctypes.data if len(a) else None for a in arrs])
        cnt = (C.c_int * 5)(*[len(a) for a in arrs])
        T = np.ascontiguousarray(pose, np.float32)
        info = LocalMapInfo()
        self._chk(self._L.lisreg_localmap_insert(self._h, map_id, ptrs, cnt, arrs[0].dtype.itemsize, _fmt_of(arrs[0]),
                                                 T.ctypes.data_as(C.POINTER(C.c_float)), C.byref(params), C.byref(info)))
        return _info_dict(info)

    def localmap_insert_device(self, map_id: int, ptrs, counts, pose, params: LocalMapParams) -> dict:
        p = (C.c_void_p * 5)(*[C.c_void_p(int(x)) if c else None for x, c in zip(ptrs, counts)])
        cnt = (C.c_int * 5)(*[int(x) for x in counts])
        T = np.ascontiguousarray(pose, np.float32)
        info = LocalMapInfo()
        self._chk(self._L.lisreg_localmap_insert(self._h, map_id, p, cnt, 16, FMT_DEVICE,
                                                 T.ctypes.data_as(C.POINTER(C.c_float)), C.byref(params), C.byref(info)))
        return _info_dict(info)

    def localmap_extract(self, map_id: int, cur_pose, params: LocalMapParams, target_slot: int = 0) -> dict:
        T = np.ascontiguousarray(cur_pose, np.float32)
        info = LocalMapInfo()
        self._chk(self._L.lisreg_localmap_extract(self._h, map_id, T.ctypes.data_as(C.POINTER(C.c_float)), C.byref(params),
                                                  target_slot, C.byref(info)))
        return _info_dict(info)

    # -- copy #3: submap_t + insert_submap + extractSubMapCloud (device-resident) ------------------------------------
    def submap_insert(self, map_id: int, clouds, relative_pose, submap_pose, params: LocalMapParams) -> dict:
        """clouds: the key frame's five DOWN-sampled class clouds (LOCALMAP_CLASSES order); relative_pose None = fisrt_submap."""
        arrs = [np.ascontiguousarray(a) for a in clouds]
        ptrs = (C.c_void_p * 5)(*[a.ctypes.data if len(a) else None for a in arrs])
        cnt = (C.c_int * 5)(*[len(a) for a in arrs])
        Tr = None if relative_pose is None else np.ascontiguousarray(relative_pose, np.float32)
        Ts = np.ascontiguousarray(submap_pose, np.float32)
        info = SubmapInfo()
        fp = C.POINTER(C.c_float)
        self._chk(self._L.lisreg_submap_insert(self._h, map_id, ptrs, cnt, arrs[0].dtype.itemsize, _fmt_of(arrs[0]),
                                               Tr.ctypes.data_as(fp) if Tr is not None else None, Ts.ctypes.data_as(fp), C.byref(params), C.byref(info)))
        return dict(n=list(info.n), feature_point_num=info.feature_point_num, local_bound=np.array(info.local_bound[:], np.float64),
                    bound=np.array(info.bound[:], np.float64))

    def submap_extract(self, pre_id: int, cur_id: int, pre_pose, cur_pose, pad: float = 10.0, corner_leaf: float = 0.2,
                       surf_leaf: float = 0.5, target_slot: int = 0) -> dict:
        Tp = np.ascontiguousarray(pre_pose, np.float32); Tc = np.ascontiguousarray(cur_pose, np.float32)
        out = SubmapExtractOut()
        fp = C.POINTER(C.c_float)
        self._chk(self._L.lisreg_submap_extract(self._h, pre_id, cur_id, Tp.ctypes.data_as(fp), Tc.ctypes.data_as(fp), pad, corner_leaf,
                                                surf_leaf, target_slot, C.byref(out)))
        return dict(isect=np.array(out.isect[:], np.float64), isect_local=np.array(out.isect_local[:], np.float64),
                    n_target_corner=out.n_target_corner, n_target_surf=out.n_target_surf,
                    src_corner_ptr=out.src_corner or 0, n_src_corner=out.n_src_corner, src_surf_ptr=out.src_surf or 0, n_src_surf=out.n_src_surf)

    def localmap_get(self, map_id: int, cls: int) -> np.ndarray:
        """[n, 4] float32 records (x, y, z, label bits) of class cls (0-4) or of the corner / surf target (5 / 6)."""
        n = C.c_int(0)
        self._L.lisreg_localmap_get(self._h, map_id, cls, None, 0, C.byref(n))
        out = np.zeros((max(n.value, 1), 4), np.float32)
        self._chk(self._L.lisreg_localmap_get(self._h, map_id, cls, out.ctypes.data_as(C.c_void_p), max(n.value, 1), C.byref(n)))
        return out[: n.value]

    # -- §8 f-3 -------------------------------------------------------------------------------------------
    def map_index_set(self, slot: int, cloud: np.ndarray):
        cloud = np.ascontiguousarray(cloud)
        self._chk(self._L.lisreg_map_index_set(self._h, slot, _vp(cloud), len(cloud), cloud.dtype.itemsize, _fmt_of(cloud)))

    def map_index_set_batch(self, slots, clouds):
        """lisreg_map_index_set_batch: clouds = host PCL-struct arrays (one dtype), or (device_ptr, n) tuples"""
        k = len(slots)
        sl = (C.c_int * max(k, 1))(*[int(v) for v in slots])
        ptrs = (C.c_void_p * max(k, 1))(); cnt = (C.c_int * max(k, 1))()
        keep, fmt, stride = [], FMT_DEVICE, 16
        for i, cl in enumerate(clouds):
            if isinstance(cl, tuple):
                ptrs[i], cnt[i] = C.c_void_p(cl[0]), cl[1]
            else:
                cl = np.ascontiguousarray(cl); keep.append(cl)
                ptrs[i], cnt[i] = C.c_void_p(cl.ctypes.data if len(cl) else 0), len(cl)
                fmt, stride = _fmt_of(cl), cl.dtype.itemsize
        self._chk(self._L.lisreg_map_index_set_batch(self._h, k, sl, ptrs, cnt, stride, fmt))

    def map_index_set_device(self, slot: int, ptr: int, n: int):
        self._chk(self._L.lisreg_map_index_set(self._h, slot, C.c_void_p(ptr), n, 16, FMT_DEVICE))

    def nearest(self, slot: int, query: np.ndarray, max_dist: float = 1e18):
        """k = 1 search: (idx [n] int32, -1 beyond max_dist; squared distance [n] float32)."""
        query = np.ascontiguousarray(query)
        idx = np.zeros(len(query), np.int32)
        sqd = np.zeros(len(query), np.float32)
        self._chk(self._L.lisreg_nearest(self._h, slot, _vp(query), len(query), query.dtype.itemsize, _fmt_of(query),
                                         max_dist, idx.ctypes.data_as(C.c_void_p), sqd.ctypes.data_as(C.c_void_p)))
        return idx, sqd

    def nearest_device(self, slot: int, q_ptr: int, n: int, max_dist: float, idx_ptr: int, sqd_ptr: int):
        self._chk(self._L.lisreg_nearest(self._h, slot, C.c_void_p(q_ptr), n, 16, FMT_DEVICE, max_dist,
                                         C.c_void_p(idx_ptr), C.c_void_p(sqd_ptr)))

    def dynamic_filter(self, slot: int, cloud: np.ndarray, center_radius: float, dist_thre_min: float = 3.4028234663852886e38,
                       dist_thre_max: float = 3.4028234663852886e38, near_dist_thre: float = 0.0):
        """map_scan_feature_pts_distance_removal: (filtered cloud, applied) — applied False where the reference returns false."""
        cloud = np.ascontiguousarray(cloud)
        out = np.zeros_like(cloud)
        m = C.c_int(0)
        rc = self._chk(self._L.lisreg_dynamic_filter(self._h, slot, _vp(cloud), len(cloud), cloud.dtype.itemsize, _fmt_of(cloud),
                                                     center_radius, dist_thre_min, dist_thre_max, near_dist_thre,
                                                     out.ctypes.data_as(C.c_void_p), C.byref(m)),
                       allow=(OK, NOT_ENOUGH_FEATURES))
        return out[: m.value], rc == OK

    def dynamic_filter_device(self, slot: int, in_ptr: int, n: int, center_radius: float, dist_thre_min: float,
                              dist_thre_max: float, near_dist_thre: float, out_ptr: int) -> int:
        m = C.c_int(0)
        self._chk(self._L.lisreg_dynamic_filter(self._h, slot, C.c_void_p(in_ptr), n, 16, FMT_DEVICE, center_radius,
                                                dist_thre_min, dist_thre_max, near_dist_thre, C.c_void_p(out_ptr), C.byref(m)),
                  allow=(OK, NOT_ENOUGH_FEATURES))
        return m.value

    def bbx_filter(self, cloud: np.ndarray, bounds, delete_box: bool = False) -> np.ndarray:
        """bounds = (min_x, min_y, min_z, max_x, max_y, max_z)"""
        cloud = np.ascontiguousarray(cloud)
        out = np.zeros_like(cloud)
        m = C.c_int(0)
        b = (C.c_double * 6)(*[float(v) for v in bounds])
        self._chk(self._L.lisreg_bbx_filter(self._h, _vp(cloud), len(cloud), cloud.dtype.itemsize, _fmt_of(cloud), b,
                                            1 if delete_box else 0, out.ctypes.data_as(C.c_void_p), C.byref(m)))
        return out[: m.value]

    def bbx_filter_device(self, in_ptr: int, n: int, bounds, delete_box: bool, out_ptr: int) -> int:
        m = C.c_int(0)
        b = (C.c_double * 6)(*[float(v) for v in bounds])
        self._chk(self._L.lisreg_bbx_filter(self._h, C.c_void_p(in_ptr), n, 16, FMT_DEVICE, b, 1 if delete_box else 0,
                                            C.c_void_p(out_ptr), C.byref(m)))
        return m.value

    def cloud_bounds(self, cloud: np.ndarray) -> np.ndarray:
        cloud = np.ascontiguousarray(cloud)
        b = (C.c_double * 6)()
        self._chk(self._L.lisreg_cloud_bounds(self._h, _vp(cloud), len(cloud), cloud.dtype.itemsize, _fmt_of(cloud), b))
        return np.array(list(b))

    # -- §8 f-4 -------------------------------------------------------------------------------------------
    def icp_align(self, slot: int, source: np.ndarray, params: "IcpParams", guess=None, want_aligned: bool = False):
        """pcl::IterativeClosestPoint::align against the map index in `slot`; returns the result dict (+ 'aligned')."""
        source = np.ascontiguousarray(source)
        res = IcpResult()
        g = None if guess is None else np.ascontiguousarray(guess, np.float32).ravel().ctypes.data_as(C.POINTER(C.c_float))
        out = np.zeros_like(source) if want_aligned else None
        self._chk(self._L.lisreg_icp_align(self._h, slot, _vp(source), len(source), source.dtype.itemsize, _fmt_of(source),
                                           C.byref(params), g, C.byref(res), _vp(out) if want_aligned else None))
        d = res.as_dict()
        if want_aligned:
            d["aligned"] = out
        return d

    def icp_align_batch(self, items, params: "IcpParams", chain_prev_mse: bool = False):
        """lisreg_icp_align_batch: items = [(slot, source, guess or None), ...] — host PCL-struct arrays (one dtype for the batch), or
        (slot, (device_ptr, n), guess) tuples for LISREG_FMT_DEVICE records.  Returns the list of result dicts, one per item."""
        n = len(items)
        arr = (IcpItem * max(n, 1))()
        keep = []
        fmt, stride = None, 16
        for k, (slot, source, guess) in enumerate(items):
            if isinstance(source, tuple):
                ptr, cnt = source
                arr[k].source = C.c_void_p(ptr); arr[k].n = cnt
                f, st = FMT_DEVICE, 16
            else:
                source = np.ascontiguousarray(source)
                keep.append(source)
                arr[k].source = C.c_void_p(source.ctypes.data if len(source) else 0); arr[k].n = len(source)
                f, st = _fmt_of(source), source.dtype.itemsize
            if fmt is None:
                fmt, stride = f, st
            elif (fmt, stride) != (f, st):
                raise ValueError("icp_align_batch: one point format per batch")
            arr[k].slot = slot
            if guess is not None:
                g = np.ascontiguousarray(guess, np.float32).ravel()
                keep.append(g)
                arr[k].guess = g.ctypes.data_as(C.POINTER(C.c_float))
        res = (IcpResult * max(n, 1))()
        self._chk(self._L.lisreg_icp_align_batch(self._h, arr, n, stride, FMT_DEVICE if fmt is None else fmt, C.byref(params),
                                                 1 if chain_prev_mse else 0, res))
        return [res[k].as_dict() for k in range(n)]

    def icp_gn_match(self, slot: int, source: np.ndarray, max_iterations: int, max_correspond_distance: float, predict_pose,
                     want_transformed: bool = False):
        """OptimizedICPGN::Match + GetFitnessScore against the map index in `slot`."""
        source = np.ascontiguousarray(source)
        res = IcpGnResult()
        g = np.ascontiguousarray(predict_pose, np.float32).ravel()
        out = np.zeros_like(source) if want_transformed else None
        self._chk(self._L.lisreg_icp_gn_match(self._h, slot, _vp(source), len(source), source.dtype.itemsize, _fmt_of(source),
                                              max_iterations, max_correspond_distance, g.ctypes.data_as(C.POINTER(C.c_float)),
                                              C.byref(res), _vp(out) if want_transformed else None))
        d = dict(T=np.array(list(res.final_transform), np.float32).reshape(4, 4), steps_applied=res.steps_applied,
                 n_corr_last=res.n_corr_last, fitness=res.fitness)
        if want_transformed:
            d["transformed"] = out
        return d

    def icp_align_device(self, slot: int, src_ptr: int, n: int, params: "IcpParams", guess=None, out_ptr: int = 0):
        res = IcpResult()
        g = None if guess is None else np.ascontiguousarray(guess, np.float32).ravel().ctypes.data_as(C.POINTER(C.c_float))
        self._chk(self._L.lisreg_icp_align(self._h, slot, C.c_void_p(src_ptr), n, 16, FMT_DEVICE, C.byref(params), g,
                                           C.byref(res), C.c_void_p(out_ptr) if out_ptr else None))
        return res.as_dict()

    def set_profiling(self, on: bool):
        self._chk(self._L.lisreg_set_profiling(self._h, 1 if on else 0))

    def timing(self) -> dict:
        out = (C.c_double * 5)()
        self._chk(self._L.lisreg_get_timing(self._h, out))
        return dict(assoc_ms=out[0], assoc_launches=int(out[1]), solve_ms=out[2], solve_launches=int(out[3]),
                    index_ms=out[4])


def pack_device_records(cloud: np.ndarray) -> np.ndarray:
    """PCL struct array -> contiguous [n,4] float32 view of lisreg_dpoint records (payload = label bits)."""
    n = len(cloud)
    out = np.zeros((n, 4), np.float32)
    out[:, 0], out[:, 1], out[:, 2] = cloud["x"], cloud["y"], cloud["z"]
    if cloud.dtype.names and "label" in cloud.dtype.names:
        out[:, 3] = cloud["label"].astype(np.uint32).view(np.float32)
    return out


_HIP = None


def hip_runtime():
    """The HIP runtime liblisreg.so is bound to — the copy ALREADY in the process (soname libamdhip64.so.7: the system's, or the one a
    PyTorch wheel brought along if torch was imported first), never a second one opened by file name."""
    global _HIP
    if _HIP is None:
        lib()
        for name in ("libamdhip64.so.7", "libamdhip64.so"):
            try:
                _HIP = C.CDLL(name, mode=os.RTLD_NOW | os.RTLD_NOLOAD)
                break
            except OSError:
                continue
        if _HIP is None:
            _HIP = C.CDLL("libamdhip64.so")
    return _HIP


class PinnedArray:
    """Page-locked host memory (hipHostMalloc through the library's own HIP runtime) viewed as a numpy array: what a caller's pinned
    clouds look like to the feeder's copy engine (lisreg_stage_host_items), without torch in the process."""

    def __init__(self, host: np.ndarray):
        self._hip = hip_runtime()
        host = np.ascontiguousarray(host)
        p = C.c_void_p()
        if self._hip.hipHostMalloc(C.byref(p), C.c_size_t(max(host.nbytes, 64)), C.c_uint(0)) != 0:
            raise MemoryError("hipHostMalloc failed")
        self.ptr = p.value
        self.nbytes = host.nbytes
        self.array = np.frombuffer((C.c_ubyte * max(host.nbytes, 1)).from_address(self.ptr), dtype=np.uint8, count=host.nbytes)
        self.array[:] = host.view(np.uint8).reshape(-1)

    def free(self):
        if getattr(self, "ptr", None):
            self.array = None
            self._hip.hipHostFree(C.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceArray:
    """Minimal device buffer through the SAME HIP runtime liblisreg.so uses (hipMalloc/hipMemcpy via ctypes), for
    callers that hand device-resident clouds to the library without torch."""

    def __init__(self, host: np.ndarray):
        self._hip = hip_runtime()
        host = np.ascontiguousarray(host)
        self.nbytes = host.nbytes
        self.shape = host.shape
        p = C.c_void_p()
        if self._hip.hipMalloc(C.byref(p), C.c_size_t(max(self.nbytes, 16))) != 0:
            raise MemoryError("hipMalloc failed")
        self.ptr = p.value
        if self.nbytes and self._hip.hipMemcpy(C.c_void_p(self.ptr), host.ctypes.data_as(C.c_void_p),
                                               C.c_size_t(self.nbytes), 1) != 0:
            raise RuntimeError("hipMemcpy H2D failed")

    def upload(self, host: np.ndarray, offset_bytes: int = 0):
        """H2D into this buffer (it must be large enough)."""
        host = np.ascontiguousarray(host)
        assert offset_bytes + host.nbytes <= max(self.nbytes, 16)
        if host.nbytes and self._hip.hipMemcpy(C.c_void_p(self.ptr + offset_bytes), host.ctypes.data_as(C.c_void_p), C.c_size_t(host.nbytes), 1) != 0:
            raise RuntimeError("hipMemcpy H2D failed")

    def copy_from_device(self, src_ptr: int, nbytes: int, offset_bytes: int = 0):
        assert offset_bytes + nbytes <= max(self.nbytes, 16)
        if nbytes and self._hip.hipMemcpy(C.c_void_p(self.ptr + offset_bytes), C.c_void_p(src_ptr), C.c_size_t(nbytes), 3) != 0:
            raise RuntimeError("hipMemcpy D2D failed")

    def download(self, n_rows: int) -> np.ndarray:
        out = np.zeros((n_rows,) + tuple(self.shape[1:]), np.float32)
        if out.nbytes and self._hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr), C.c_size_t(out.nbytes), 2) != 0:
            raise RuntimeError("hipMemcpy D2H failed")
        return out

    def free(self):
        if getattr(self, "ptr", None):
            self._hip.hipFree(C.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def comm_unique_id() -> bytes:
    buf = (C.c_ubyte * 128)()
    rc = lib().lisreg_comm_unique_id(buf)
    if rc:
        raise LisregError(rc, lib().lisreg_last_error(None).decode())
    return bytes(buf)


def _ctx_comm_init(self, rank: int, nranks: int, uid: bytes):
    buf = (C.c_ubyte * 128).from_buffer_copy(uid)
    self._chk(self._L.lisreg_comm_init(self._h, rank, nranks, buf))


def _ctx_gather_results(self, local_ptr: int, n_local: int, out_ptr: int):
    self._chk(self._L.lisreg_gather_results(self._h, C.c_void_p(local_ptr), n_local, C.c_void_p(out_ptr)))


def _ctx_comm_destroy(self):
    self._L.lisreg_comm_destroy(self._h)


def device_count() -> int:
    return int(lib().lisreg_device_count())


Context.comm_init = _ctx_comm_init
Context.gather_results = _ctx_gather_results
Context.comm_destroy = _ctx_comm_destroy


def device_to_host(ptr: int, shape, dtype=np.float32) -> np.ndarray:
    """Blocking D2H copy through the library's HIP runtime (tests only)."""
    out = np.zeros(shape, dtype)
    hip = hip_runtime()
    hip.hipDeviceSynchronize()
    if hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), C.c_size_t(out.nbytes), 2) != 0:
        raise RuntimeError("hipMemcpy D2H failed")
    return out


def default_feature_params() -> FeatureParams:
    p = FeatureParams()
    rc = lib().lisreg_default_feature_params(C.byref(p))
    if rc:
        raise LisregError(rc, "lisreg_default_feature_params")
    return p
