"""CPU restatement of the reference's sequential frame loop — TEST INFRASTRUCTURE ONLY (see lisreg_oracle.h).

Composes the C restatement's primitives (oracle_ctypes) in the order of
  SubMapOptmizationNode::makeSubMapThread     /root/reference/src/node/subMapOptmizationNode.cpp:597-755
  keyframeInit (per-class voxel grids)        :758-852
  currentCloudInit (source assembly)          :856-893
  updateInitialGuess (no IMU, no odometry)    :984-1020 (same arithmetic as odomEstimationNode.cpp:351-392)
  extractTargetCloud -> extractSlidingCloud   :1146-1200, 1369-1432
  scan2SubMapOptimization (copy #2)           :1509-1541  -> oracle_ctypes.align(variant 2)
  SubMapManager::insert_local_map             src/include/subMap.h:979-1059
on host clouds in the reference's PointXYZIL layout.  Only tests/, bench.py's cpu_baseline leg and tools/kitti_replay.py
--check may import this module."""
from __future__ import annotations

import numpy as np

import oracle_ctypes as oc

CLASSES = ("dynamic", "pole", "ground", "building", "outlier")        # append_feature / merge_feature_points order
FRAME_LEAF = dict(dynamic=0.2, pole=0.05, ground=0.6, building=0.4, outlier=0.6)     # keyframeInit :806-811
MAP_LEAF = (0.1, 0.05, 0.4, 0.2, 0.6)                                                 # extractSlidingCloud :1385-1389
CROP_BOX = (-70.0, -70.0, -10.0, 70.0, 70.0, 20.0)                                    # :1377-1379


def cat(clouds):
    """`*a += *b` on PCL struct arrays, keeping the padded 32-byte layout (np.concatenate would re-pack the dtype)."""
    out = np.zeros(sum(len(c) for c in clouds), clouds[0].dtype)
    o = 0
    for c in clouds:
        out[o:o + len(c)] = c
        o += len(c)
    return out


def pose_matrix_f32(T):
    """pcl::getTransformation in float (trans2Affine3f, common.cpp:54-57) via the oracle's own restatement."""
    Tf = np.ascontiguousarray(T, np.float32)
    M = np.zeros(12, np.float32)
    oc.lib().orc_pose_to_matrix(oc._fp(Tf), oc._fp(M))
    return M.reshape(3, 4)


def _libm():
    import ctypes, ctypes.util
    m = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
    for f in ("atan2f", "asinf"):
        getattr(m, f).restype = ctypes.c_float
    m.atan2f.argtypes = [ctypes.c_float, ctypes.c_float]; m.asinf.argtypes = [ctypes.c_float]
    return m


def predict_pose(T_last, T_cur):
    """updateInitialGuess, constant-velocity branch (subMapOptmizationNode.cpp:1003-1020, odomEstimationNode.cpp:351-392):
    transFinal = transTobe * (transLast.inverse() * transTobe) on Eigen::Affine3f — every operation in float, the inverse by cofactors
    (Eigen's 3x3 path), products accumulated left to right; then pcl::getTranslationAndEulerAngles with the C library's float atan2 / asin."""
    f = np.float32
    A, B = pose_matrix_f32(T_last), pose_matrix_f32(T_cur)
    Ai = _affine_inverse_f32(A)
    def mul(X, Y):                                             # affine product, rows of X times columns of Y, float, left to right
        Z = np.zeros((3, 4), np.float32)
        for r in range(3):
            for q in range(3):
                Z[r, q] = f(f(f(X[r, 0] * Y[0, q]) + f(X[r, 1] * Y[1, q])) + f(X[r, 2] * Y[2, q]))
            Z[r, 3] = f(f(f(f(X[r, 0] * Y[0, 3]) + f(X[r, 1] * Y[1, 3])) + f(X[r, 2] * Y[2, 3])) + X[r, 3])
        return Z
    F = mul(B, mul(Ai, B))
    m = _libm()
    return np.array([m.atan2f(float(F[2, 1]), float(F[2, 2])), m.asinf(float(-F[2, 0])), m.atan2f(float(F[1, 0]), float(F[0, 0])),
                     F[0, 3], F[1, 3], F[2, 3]], np.float32)


def _aff_mul_f32(X, Y):
    """Eigen::Affine3f product, float, accumulated left to right"""
    f = np.float32
    Z = np.zeros((3, 4), np.float32)
    for r in range(3):
        for q in range(3):
            Z[r, q] = f(f(f(X[r, 0] * Y[0, q]) + f(X[r, 1] * Y[1, q])) + f(X[r, 2] * Y[2, q]))
        Z[r, 3] = f(f(f(f(X[r, 0] * Y[0, 3]) + f(X[r, 1] * Y[1, 3])) + f(X[r, 2] * Y[2, 3])) + X[r, 3])
    return Z


def _aff_to_pose(F):
    m = _libm()
    return np.array([m.atan2f(float(F[2, 1]), float(F[2, 2])), m.asinf(float(-F[2, 0])), m.atan2f(float(F[1, 0]), float(F[0, 0])),
                     F[0, 3], F[1, 3], F[2, 3]], np.float32)


class InitialGuessOracle:
    """updateInitialGuess with its statics: variant 0 = odomEstimationNode.cpp:297-419, variant 1 = subMapOptmizationNode.cpp:896-1032.
    Written from the two listings branch by branch (test infrastructure; the product's lisreg_update_initial_guess is compared with it)."""

    def __init__(self, variant=0, use_imu_heading_initialization=False):
        self.variant, self.heading = variant, use_imu_heading_initialization
        self.firstTransAvailable = False
        self.lastImuPreTransAvailable = False
        self.first = False
        self.lastImuTransformation = None
        self.lastImuPreTransformation = None
        self.lastT = np.zeros(6, np.float32)

    @staticmethod
    def _incr(frm, to, T):
        transIncre = _aff_mul_f32(_affine_inverse_f32(frm), to)
        transFinal = _aff_mul_f32(pose_matrix_f32(T), transIncre)
        return _aff_to_pose(transFinal)

    def update(self, T, odom_available=False, imu_available=False, imu_rpy=(0.0, 0.0, 0.0), initial_guess=(0.0,) * 6):
        T = np.array(T, np.float32); pred = None
        imu = pose_matrix_f32([imu_rpy[0], imu_rpy[1], imu_rpy[2], 0, 0, 0])
        if not self.firstTransAvailable:                                  # :305-318 | :902-923
            T[0], T[1], T[2] = np.float32(imu_rpy[0]), np.float32(imu_rpy[1]), np.float32(imu_rpy[2])
            if not self.heading:
                T[2] = 0
            self.lastImuTransformation = imu
            self.firstTransAvailable = True
            return T, pred
        if odom_available:                                                # :322-347 | :928-959
            x, y, z, ro, pi, ya = initial_guess
            transBack = pose_matrix_f32([ro, pi, ya, x, y, z])
            if not self.lastImuPreTransAvailable:
                self.lastImuPreTransformation = transBack
                self.lastImuPreTransAvailable = True
                if self.variant == 1:
                    if imu_available:
                        self.lastImuTransformation = imu
                    return T, pred
                # copy #1 has no return here: control reaches the tests below
            else:
                T = self._incr(self.lastImuPreTransformation, transBack, T)
                pred = T.copy()
                self.lastImuPreTransformation = transBack
                if self.variant == 0 or imu_available:
                    self.lastImuTransformation = imu
                return T, pred
        if self.variant == 0:
            if not odom_available:                                        # :351
                return self._cv(T), pred
            if imu_available:                                             # :394-415
                T = self._incr(self.lastImuTransformation, imu, T)
                pred = T.copy()
                self.lastImuTransformation = imu
            return T, pred
        if imu_available:                                                 # :962-982
            T = self._incr(self.lastImuTransformation, imu, T)
            pred = T.copy()
            self.lastImuTransformation = imu
            return T, pred
        return self._cv(T), pred                                          # :986-1020 (neither input)

    def _cv(self, T):
        if not self.first:
            self.lastT = T.copy(); self.first = True
            return T
        transBack, transLast = pose_matrix_f32(T), pose_matrix_f32(self.lastT)
        self.lastT = T.copy()
        return self._incr(transLast, transBack, T)


class LocalMapOracle:
    """localMap_t + insert_local_map + extractSlidingCloud on host struct arrays."""

    def __init__(self, dtype):
        self.cls = [np.zeros(0, dtype) for _ in range(5)]
        self.feature_point_num = 0
        self.bound = np.array([np.finfo(np.float64).max] * 3 + [-np.finfo(np.float64).max] * 3)

    def insert(self, clouds, pose, max_num_pts=80000, dynamic_removal_on=True, center_radius=30.0, thre_min=0.3, thre_max=3.0,
               near_thre=0.03):
        """subMap.h:979-1059.  clouds: five un-downsampled class clouds (CLASSES order) in the sensor frame."""
        thre_max = max(np.float32(thre_max), np.float32(np.float64(np.float32(thre_min)) + 0.1))           # :1006
        moved = [oc.transform_cloud(c, pose, fmt=1) if len(c) else c for c in clouds[:4]]                # outlier: commented out (:1003)
        if dynamic_removal_on and self.feature_point_num > max_num_pts // 5 and len(self.cls[0]) > 0:
            moved[0], _ = oc.dynamic_filter(self.cls[0], moved[0], center_radius, float(np.float32(thre_min)), float(thre_max),
                                            float(np.float32(near_thre)))
        for k in range(4):
            self.cls[k] = cat([self.cls[k], moved[k]])
        self.feature_point_num = sum(len(c) for c in self.cls)
        allpts = cat(self.cls)
        self.bound = oc.cloud_bounds(allpts)

    def extract(self, cur_pose, leaf=MAP_LEAF, crop_box=CROP_BOX, pad=2.0):
        """subMapOptmizationNode.cpp:1369-1432.  Returns (corner target, surf target, bbx_intersection)."""
        M = pose_matrix_f32(cur_pose).astype(np.float64)             # float entries, double arithmetic (transform_bbx)
        bx = np.array(crop_box, np.float64)
        cp = 0.5 * (bx[:3] + bx[3:])
        cpo = M[:, 0] * cp[0] + M[:, 1] * cp[1] + M[:, 2] * cp[2] + M[:, 3]
        cur = np.concatenate([bx[:3] - cp + cpo, bx[3:] - cp + cpo])
        isect = np.concatenate([np.maximum(cur[:3], self.bound[:3]) - pad, np.minimum(cur[3:], self.bound[3:]) + pad])
        for k in range(5):
            if len(self.cls[k]):
                rc, ds = oc.voxel_grid(self.cls[k], float(np.float32(leaf[k])), fmt=1)
                self.cls[k] = ds
        for k in range(5):
            if len(self.cls[k]):
                self.cls[k] = oc.bbx_filter(self.cls[k], isect)
        return self.cls[1].copy(), cat([self.cls[2], self.cls[3], self.cls[0]]), isect


def _transform_bbx(b, M):
    """subMap.h:214-228: the centre goes through the float matrix in double arithmetic, the box keeps its extents."""
    M = M.astype(np.float64)
    cp = 0.5 * (b[:3] + b[3:])
    cpo = M[:, 0] * cp[0] + M[:, 1] * cp[1] + M[:, 2] * cp[2] + M[:, 3]
    return np.concatenate([b[:3] - cp + cpo, b[3:] - cp + cpo])


def _affine_inverse_f32(A):
    """Eigen::Affine3f::inverse(): cofactor inverse of the linear part, t' = -L^-1 t, every operation in float."""
    f = np.float32
    a, b, c, d, e, ff, g, h, i = [f(v) for v in A[:, :3].ravel()]
    c00, c01, c02 = f(f(e * i) - f(ff * h)), f(f(ff * g) - f(d * i)), f(f(d * h) - f(e * g))
    det = f(f(f(a * c00) + f(b * c01)) + f(c * c02))
    idet = f(f(1) / det)
    L = np.array([[f(c00 * idet), f(f(f(c * h) - f(b * i)) * idet), f(f(f(b * ff) - f(c * e)) * idet)],
                  [f(c01 * idet), f(f(f(a * i) - f(c * g)) * idet), f(f(f(c * d) - f(a * ff)) * idet)],
                  [f(c02 * idet), f(f(f(b * g) - f(a * h)) * idet), f(f(f(a * e) - f(b * d)) * idet)]], np.float32)
    t = A[:, 3].astype(np.float32)
    ti = np.array([-f(f(f(L[r, 0] * t[0]) + f(L[r, 1] * t[1])) + f(L[r, 2] * t[2])) for r in range(3)], np.float32)
    return np.concatenate([L, ti[:, None]], 1)


def submap_crop_boxes(pre_local_bound, pre_pose, cur_local_bound, cur_pose, pad=10.0):
    """extractSubMapCloud's two boxes (subMapOptmizationNode.cpp:3988-3995, 4055-4058)."""
    Mp, Mc = pose_matrix_f32(pre_pose), pose_matrix_f32(cur_pose)
    pre, cur = _transform_bbx(np.asarray(pre_local_bound, np.float64), Mp), _transform_bbx(np.asarray(cur_local_bound, np.float64), Mc)
    isect = np.concatenate([np.maximum(cur[:3], pre[:3]) - np.float64(np.float32(pad)), np.minimum(cur[3:], pre[3:]) + np.float64(np.float32(pad))])
    return isect, _transform_bbx(isect, _affine_inverse_f32(Mc))


class SubMapOracle(LocalMapOracle):
    """submap_t + fisrt_submap / insert_submap (subMap.h:785-978) on host struct arrays; `bound` is the LOCAL bound."""

    def insert(self, clouds, relative_pose, max_num_pts=80000, dynamic_removal_on=True, center_radius=30.0, thre_min=0.3, thre_max=3.0,
               near_thre=0.03):
        """clouds: the key frame's five DOWN-sampled class clouds; relative_pose None = fisrt_submap (appended as they are)."""
        thre_max = max(np.float32(thre_max), np.float32(np.float64(np.float32(thre_min)) + 0.1))           # :884
        if relative_pose is None:
            moved = list(clouds)
        else:
            moved = [oc.transform_cloud(c, relative_pose, fmt=1) if len(c) else c for c in clouds]         # all five (:878-882)
            if dynamic_removal_on and self.feature_point_num > max_num_pts // 5 and len(self.cls[0]) > 0:
                moved[0], _ = oc.dynamic_filter(self.cls[0], moved[0], center_radius, float(np.float32(thre_min)), float(thre_max),
                                                float(np.float32(near_thre)))
        for k in range(5):
            self.cls[k] = cat([self.cls[k], moved[k]])
        self.feature_point_num = sum(len(c) for c in self.cls)
        self.bound = oc.cloud_bounds(cat(self.cls))

    def global_bound(self, submap_pose):
        return _transform_bbx(self.bound, pose_matrix_f32(submap_pose))


def extract_submap_cloud(pre: "SubMapOracle", cur: "SubMapOracle", pre_pose, cur_pose, pad=10.0, corner_leaf=0.2, surf_leaf=0.5):
    """extractSubMapCloud (subMapOptmizationNode.cpp:3976-4081): (target corner, target surf, source corner, source surf, isect, isect_local)."""
    isect, isect_local = submap_crop_boxes(pre.bound, pre_pose, cur.bound, cur_pose, pad)
    tc = pre.cls[1]
    ts = cat([pre.cls[2], pre.cls[3], pre.cls[0]])
    tc = oc.transform_cloud(tc, pre_pose, fmt=1) if len(tc) else tc
    ts = oc.transform_cloud(ts, pre_pose, fmt=1) if len(ts) else ts
    tc = oc.bbx_filter(tc, isect) if len(tc) else tc
    ts = oc.bbx_filter(ts, isect) if len(ts) else ts
    sc = cur.cls[1]
    ss = cat([cur.cls[0], cur.cls[2], cur.cls[3]])
    sc = oc.bbx_filter(sc, isect_local) if len(sc) else sc
    ss = oc.bbx_filter(ss, isect_local) if len(ss) else ss
    sc = oc.voxel_grid(sc, float(np.float32(corner_leaf)), fmt=1)[1] if len(sc) else sc
    ss = oc.voxel_grid(ss, float(np.float32(surf_leaf)), fmt=1)[1] if len(ss) else ss
    return tc, ts, sc, ss, isect, isect_local


def split_and_downsample(labelled_cloud):
    """SemanticFusionNode::categoryMapping + keyframeInit's per-class voxel grids.  Returns (full, down) dicts keyed by class."""
    dyn, ground, building, pole, outlier = oc.semantic_split(labelled_cloud)
    full = dict(dynamic=dyn, pole=pole, ground=ground, building=building, outlier=outlier)
    down = {}
    for k, c in full.items():
        down[k] = oc.voxel_grid(c, FRAME_LEAF[k], fmt=1)[1] if len(c) else c
    return full, down


def replay(frames, n_threads=8, params=None, on_frame=None, guess_inputs=None):
    """frames: iterable of labelled PointXYZIL clouds (one sweep each).  Returns a list of per-frame dicts
    (T = transformTobeSubMapped after the frame, guess, stats, n_target_corner / n_target_surf, n_src_corner / n_src_surf).
    guess_inputs: per frame the cloudInfo fields updateInitialGuess reads (dict: odom_available, imu_available, imu_rpy, initial_guess) —
    the whole of subMapOptmizationNode.cpp:896-1032 then runs every frame, and an available IMU also enters transformUpdate."""
    p = params or oc.default_params(2)
    out = []
    lm = None
    T = np.zeros(6, np.float32)
    T_last = None
    have_last = False
    gs = InitialGuessOracle(1) if guess_inputs is not None else None
    for k, cloud in enumerate(frames):
        full, down = split_and_downsample(cloud)
        if lm is None:
            lm = LocalMapOracle(cloud.dtype)
        rec = dict(frame=k)
        imu = None
        if gs is not None:
            gi = guess_inputs[k]
            T, _ = gs.update(T, **gi)
            if gi.get("imu_available"):
                imu = oc.Imu(1, float(gi["imu_rpy"][0]), float(gi["imu_rpy"][1]))
        if k == 0:                                             # subMapFirstFlag branch (:634-651)
            rec.update(T=T.copy(), guess=T.copy(), stats=None)
        else:
            if gs is not None:
                guess = T.copy()
            elif not have_last:                                # updateInitialGuess: the first call only records (:1003-1011)
                T_last = T.copy(); have_last = True
                guess = T.copy()
            else:
                guess = predict_pose(T_last, T)
                T_last = T.copy()
            tc, ts, isect = lm.extract(guess)
            src_c = down["pole"]                                                                  # currentCloudInit :866-868
            src_s = cat([down["dynamic"], down["building"], down["ground"]])                     # :873-889
            Tn, st, _ = oc.align(tc, ts, src_c, src_s, guess, p, imu=imu, n_threads=n_threads, max_trace=1)
            T = Tn.astype(np.float32)
            rec.update(T=T.copy(), guess=guess.copy(), stats=st, n_target_corner=len(tc), n_target_surf=len(ts),
                       n_src_corner=len(src_c), n_src_surf=len(src_s), crop=isect)
        lm.insert([full[c] for c in CLASSES], T)
        rec.update(n_map=[len(c) for c in lm.cls], feature_point_num=lm.feature_point_num, bound=lm.bound.copy())
        out.append(rec)
        if on_frame:
            on_frame(rec)
    return out


# ---- the odometry loop of odomEstimationNode (copy #1) on raw sweeps -----------------------------------------------------
ODOM = dict(corner_leaf=0.2, surf_leaf=0.4, key_dist=1.4, key_yaw=0.5)          # /root/reference/config/params.yaml:132-141


def _xyzi(cloud, idx):
    """pcl::fromROSMsg into PointXYZI: x, y, z, intensity of the selected points (odomEstimationNode.cpp:266-267)."""
    from lisreg import synth                                         # layout helper only (struct dtype)
    c = cloud[idx]
    xyz = np.stack([c["x"], c["y"], c["z"]], 1).astype(np.float32)
    return synth.to_pcl(xyz, None, np.asarray(c["intensity"], np.float32))


def _increment(T_from, T_to):
    A = np.vstack([pose_matrix_f32(T_from), [[0, 0, 0, 1]]]).astype(np.float64)
    B = np.vstack([pose_matrix_f32(T_to), [[0, 0, 0, 1]]]).astype(np.float64)
    F = np.linalg.inv(A) @ B
    return np.array([np.arctan2(F[2, 1], F[2, 2]), np.arcsin(-F[2, 0]), np.arctan2(F[1, 0], F[0, 0]), F[0, 3], F[1, 3], F[2, 3]])


def replay_odom(sweeps, feature_params=None, n_threads=8, on_frame=None):
    """laserCloudInfoHandler (odomEstimationNode.cpp:164-232) on PointXYZIRT sweeps, every step through the C restatement:
    orc_extract_features, orc_voxel_grid, orc_transform_cloud, orc_align (variant 1).  Returns per-frame dicts like replay()."""
    fp = feature_params or oc.default_feature_params()
    p = oc.default_params(1)
    T = np.zeros(6, np.float32)
    T_last, calls = None, 0
    key_c, key_s = [], []
    T_pri = np.zeros(6, np.float32)
    key_id = 0
    out = []
    for k, sw in enumerate(sweeps):
        f = oc.extract_features(sw, fp)
        corner, surf = _xyzi(sw, f["corner"]), _xyzi(sw, f["surface"])
        calls += 1                                                  # updateInitialGuess (:298-384), no IMU / odometry input
        if calls > 1:
            if T_last is None:
                T_last = T.copy()
            else:
                g = predict_pose(T_last, T)
                T_last = T.copy()
                T = g
        rec = dict(frame=k, n_corner=len(corner), n_surf=len(surf), guess=T.copy(), stats=None, keyframe=False)

        def save():
            nonlocal T_pri, key_id
            key_c.append(oc.transform_cloud(corner, T, fmt=1) if len(corner) else corner)
            key_s.append(oc.transform_cloud(surf, T, fmt=1) if len(surf) else surf)
            while len(key_s) >= 20:
                key_s.pop(0); key_c.pop(0)
            T_pri = T.copy(); key_id += 1
            rec["keyframe"] = True

        if k == 0:
            save()
        else:
            tc, ts = cat(key_c[::-1]), cat(key_s[::-1])
            tc = oc.voxel_grid(tc, ODOM["corner_leaf"], fmt=1)[1] if len(tc) else tc
            ts = oc.voxel_grid(ts, ODOM["surf_leaf"], fmt=1)[1] if len(ts) else ts
            sc = oc.voxel_grid(corner, ODOM["corner_leaf"], fmt=1)[1] if len(corner) else corner
            ss = oc.voxel_grid(surf, ODOM["surf_leaf"], fmt=1)[1] if len(surf) else surf
            Tn, st, _ = oc.align(tc, ts, sc, ss, T, p, n_threads=n_threads, max_trace=1)
            T = Tn.astype(np.float32)
            rec.update(stats=st, n_target_corner=len(tc), n_target_surf=len(ts), n_src_corner=len(sc), n_src_surf=len(ss))
            if st["status"] == 0 and (st["deltaR"] < 0.005 or st["deltaT"] < 0.05):
                inc = _increment(T_pri, T)
                if key_id <= 5 or abs(inc[2]) >= ODOM["key_yaw"] or abs(inc[3]) >= ODOM["key_dist"] or abs(inc[4]) >= ODOM["key_dist"]:
                    save()
        rec.update(T=T.copy(), key_id=key_id)
        out.append(rec)
        if on_frame:
            on_frame(rec)
    return out
