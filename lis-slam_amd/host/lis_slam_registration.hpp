// lis_slam_registration.hpp — C++ host-side mirror of the reference's scan-to-submap interface, above the C ABI.
//
// The reference has no library seam for this path: three node classes each own a private
// scan2SubMapOptimization() that talks to ~15 member variables.  This header keeps their names
// and meaning so a maintainer can swap each body for one call:
//   OdomEstimationNode      /root/reference/src/node/odomEstimationNode.cpp:29-97 (members), :596-626 (driver)
//   SubMapOdometryNode      src/node/subMapOptmizationNode.cpp:151-203, :1509-1541
//   SubMapOptmizationNode   src/node/subMapOptmizationNode.cpp:3302-3333, :4485-4540
// plus POD mirrors of the message surface (msg/cloud_info.msg, msg/semantic_info.msg) and a PointCloud2
// byte-layout view, because ROS/PCL are not part of this build.  Header-only, C++17, no dependencies.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/lisreg.h"

namespace lis_slam {

// ---- point types (src/include/common.h:9,25-35): 32 bytes, 16-byte aligned, PCL_ADD_POINT4D + intensity -------
struct alignas(16) PointXYZI  { float x, y, z, _pad0; float intensity; float _pad1[3]; };
struct alignas(16) PointXYZIL { float x, y, z, _pad0; float intensity; uint16_t label; uint16_t _pad1; float _pad2[2]; };
static_assert(sizeof(PointXYZI) == 32 && sizeof(PointXYZIL) == 32, "PCL point layout");
using PointType = PointXYZI;                       // typedef pcl::PointXYZI PointType (common.h:9)

template <class P> struct PointCloud {             // the slice of pcl::PointCloud<P> the path touches
    std::vector<P> points;
    size_t size() const { return points.size(); }
    void clear() { points.clear(); }
    void push_back(const P& p) { points.push_back(p); }
};

// ---- sensor_msgs/PointCloud2 byte-layout view: pass a ROS message buffer straight through ---------------------
struct PointCloud2View {
    const uint8_t* data = nullptr;
    uint32_t width = 0, height = 1, point_step = 32;
    int off_x = 0, off_y = 4, off_z = 8, off_intensity = 16, off_label = -1;   // field offsets by name
    size_t size() const { return (size_t)width * height; }
    int fmt() const { return off_label >= 0 ? LISREG_FMT_XYZIL : LISREG_FMT_XYZI; }
};

// ---- msg/cloud_info.msg:1-25 and msg/semantic_info.msg:1-34 (scalars + the clouds the path reads) --------------
struct cloud_info {
    bool  imuAvailable = false, odomAvailable = false;
    float imuRollInit = 0, imuPitchInit = 0, imuYawInit = 0;
    float initialGuessX = 0, initialGuessY = 0, initialGuessZ = 0;
    float initialGuessRoll = 0, initialGuessPitch = 0, initialGuessYaw = 0;
    PointCloud2View cloud_deskewed, cloud_corner, cloud_surface, cloud_corner_sharp, cloud_surface_sharp;
};
struct semantic_info : cloud_info {
    PointCloud2View semantic_raw, semantic_dynamic, semantic_pole, semantic_ground, semantic_building, semantic_outlier;
};

enum class Variant { Odom = LISREG_VARIANT_ODOM, KeyFrame = LISREG_VARIANT_KEYFRAME, SubMap = LISREG_VARIANT_SUBMAP };

class RegistrationError : public std::runtime_error { public: int code; RegistrationError(int c, const std::string& m) : std::runtime_error(m), code(c) {} };

// One instance per reference node class (each owns a lisreg context = its own HIP stream; #2 and #3 run
// concurrently in one process, subMapOptmizationNode.cpp:5188-5195).
template <class PointT = PointType>
class Scan2SubMapRegistration {
public:
    // members the reference's callers read after the call (odomEstimationNode.cpp:66-71)
    float transformTobeMapped[6] = { 0, 0, 0, 0, 0, 0 };
    bool  isDegenerate = false;
    float deltaR = 100, deltaT = 100;
    int   iterCount = 0;
    int   laserCloudSelNum = 0;
    lisreg_params params;

    explicit Scan2SubMapRegistration(Variant v, int device = 0) {
        lisreg_default_params((int)v, &params);
        int rc = lisreg_create(device, &ctx_);
        if (rc != LISREG_OK) throw RegistrationError(rc, lisreg_last_error(nullptr));
    }
    ~Scan2SubMapRegistration() { lisreg_destroy(ctx_); }
    Scan2SubMapRegistration(const Scan2SubMapRegistration&) = delete;
    Scan2SubMapRegistration& operator=(const Scan2SubMapRegistration&) = delete;

    // kdtreeCornerFromMap->setInputCloud(...); kdtreeSurfFromMap->setInputCloud(...)   (:602-603)
    void setInputTarget(const PointCloud<PointT>& laserCloudCornerFromMapDS, const PointCloud<PointT>& laserCloudSurfFromMapDS) {
        check(lisreg_set_target(ctx_, laserCloudCornerFromMapDS.points.data(), (int)laserCloudCornerFromMapDS.size(),
                                laserCloudSurfFromMapDS.points.data(), (int)laserCloudSurfFromMapDS.size(),
                                (int)sizeof(PointT), fmt()));
    }
    // submap API (subMap.h:435-777): corner = pole; surf = ground + building + dynamic (:1408-1419)
    void setInputTargetFromClasses(const PointCloud<PointT>& pole, const PointCloud<PointT>& ground,
                                   const PointCloud<PointT>& building, const PointCloud<PointT>& dynamic) {
        check(lisreg_target_from_classes(ctx_, 0, pole.points.data(), (int)pole.size(), ground.points.data(), (int)ground.size(),
                                         building.points.data(), (int)building.size(), dynamic.points.data(), (int)dynamic.size(),
                                         (int)sizeof(PointT), fmt()));
    }

    // The body of scan2SubMapOptimization() (:596-626): guard, GN loop, transformUpdate.  Returns the lisreg
    // status: LISREG_OK, LISREG_NOT_ENOUGH_FEATURES (the reference's ROS_WARN branch, pose untouched) or
    // LISREG_TOO_FEW_CORRESPONDENCES.  transformTobeMapped is in/out exactly like the member.
    int scan2SubMapOptimization(const PointCloud<PointT>& laserCloudCornerLastDS, const PointCloud<PointT>& laserCloudSurfLastDS,
                                const cloud_info& cloudInfo) {
        lisreg_imu imu{ cloudInfo.imuAvailable ? 1 : 0, cloudInfo.imuRollInit, cloudInfo.imuPitchInit };
        lisreg_stats st{};
        int rc = lisreg_align(ctx_, laserCloudCornerLastDS.points.data(), (int)laserCloudCornerLastDS.size(),
                              laserCloudSurfLastDS.points.data(), (int)laserCloudSurfLastDS.size(), (int)sizeof(PointT), fmt(),
                              &params, &imu, transformTobeMapped, &st);
        if (rc < 0) throw RegistrationError(rc, lisreg_last_error(ctx_));
        iterCount = st.iters; laserCloudSelNum = st.n_corr_last;
        if (rc != LISREG_NOT_ENOUGH_FEATURES) { isDegenerate = st.degenerate != 0; deltaR = st.deltaR; deltaT = st.deltaT; }
        return rc;
    }
    // same, reading the feature clouds straight out of PointCloud2 message buffers
    int scan2SubMapOptimization(const PointCloud2View& corner, const PointCloud2View& surf, const cloud_info& cloudInfo) {
        lisreg_imu imu{ cloudInfo.imuAvailable ? 1 : 0, cloudInfo.imuRollInit, cloudInfo.imuPitchInit };
        lisreg_stats st{};
        int rc = lisreg_align(ctx_, corner.data, (int)corner.size(), surf.data, (int)surf.size(), (int)corner.point_step, corner.fmt(),
                              &params, &imu, transformTobeMapped, &st);
        if (rc < 0) throw RegistrationError(rc, lisreg_last_error(ctx_));
        iterCount = st.iters; laserCloudSelNum = st.n_corr_last;
        if (rc != LISREG_NOT_ENOUGH_FEATURES) { isDegenerate = st.degenerate != 0; deltaR = st.deltaR; deltaT = st.deltaT; }
        return rc;
    }
    // subMap2SubMapOptimization() of SubMapOptmizationNode is the same call with Variant::SubMap (:4485-4540)
    int subMap2SubMapOptimization(const PointCloud<PointT>& c, const PointCloud<PointT>& s, const cloud_info& ci) { return scan2SubMapOptimization(c, s, ci); }

    // the same with the sources already in HBM as 16-byte records (what SubMap<>::extractSubMapCloud and the device-resident maps hand over)
    int subMap2SubMapOptimization(const lisreg_submap_extract_out& x, const cloud_info& cloudInfo) {
        lisreg_imu imu{ cloudInfo.imuAvailable ? 1 : 0, cloudInfo.imuRollInit, cloudInfo.imuPitchInit };
        lisreg_stats st{};
        int rc = lisreg_align(ctx_, x.src_corner, x.n_src_corner, x.src_surf, x.n_src_surf, 16, LISREG_FMT_DEVICE, &params, &imu, transformTobeMapped, &st);
        if (rc < 0) throw RegistrationError(rc, lisreg_last_error(ctx_));
        iterCount = st.iters; laserCloudSelNum = st.n_corr_last;
        if (rc != LISREG_NOT_ENOUGH_FEATURES) { isDegenerate = st.degenerate != 0; deltaR = st.deltaR; deltaT = st.deltaT; }
        return rc;
    }

    lisreg_ctx* handle() { return ctx_; }

private:
    static constexpr int fmt() { return std::is_same<PointT, PointXYZIL>::value ? LISREG_FMT_XYZIL : LISREG_FMT_XYZI; }
    void check(int rc) { if (rc != LISREG_OK) throw RegistrationError(rc, lisreg_last_error(ctx_)); }
    lisreg_ctx* ctx_ = nullptr;
};

// ---- the pcl::VoxelGrid surface the nodes use ------------------------------------------------------
// downSizeFilterCorner / downSizeFilterSurf (odomEstimationNode.cpp:34-35, 110-111, 196-201, 272-277) and
// voxel_downsample_pcl (src/include/subMap.h:1207-1249) call exactly setLeafSize / setInputCloud / filter.
// Shares the registration object's context (and stream), so down-sampling and registration stay ordered.
template <class PointT = PointType>
class VoxelGrid {
public:
    explicit VoxelGrid(lisreg_ctx* ctx) : ctx_(ctx) {}
    void setLeafSize(float lx, float ly, float lz) {
        if (lx != ly || ly != lz) throw RegistrationError(LISREG_ERR_ARG, "VoxelGrid: the reference only uses cubic leaves");
        leaf_ = lx;
    }
    void setInputCloud(const PointCloud<PointT>* cloud) { input_ = cloud; }
    // returns LISREG_OK or LISREG_LEAF_TOO_SMALL (PCL's warning case: output = input)
    int filter(PointCloud<PointT>& output) {
        if (!input_) throw RegistrationError(LISREG_ERR_ARG, "VoxelGrid: no input cloud");
        std::vector<PointT> tmp(input_->size());
        int n_out = 0;
        const int fmt = std::is_same<PointT, PointXYZIL>::value ? LISREG_FMT_XYZIL : LISREG_FMT_XYZI;
        int rc = lisreg_voxel_downsample(ctx_, input_->points.data(), (int)input_->size(), (int)sizeof(PointT), fmt, leaf_,
                                         tmp.data(), (int)tmp.size(), &n_out);
        if (rc < 0) throw RegistrationError(rc, lisreg_last_error(ctx_));
        tmp.resize((size_t)n_out);
        output.points.swap(tmp);
        return rc;
    }
private:
    lisreg_ctx* ctx_;
    const PointCloud<PointT>* input_ = nullptr;
    float leaf_ = 0.4f;
};

// transformPointCloud(cloudIn, &pose6D) (src/core/common.cpp:130-173); pose = {roll, pitch, yaw, x, y, z}
template <class PointT>
inline PointCloud<PointT> transformPointCloud(lisreg_ctx* ctx, const PointCloud<PointT>& cloudIn, const float pose[6]) {
    PointCloud<PointT> out;
    out.points.resize(cloudIn.size());
    const int fmt = std::is_same<PointT, PointXYZIL>::value ? LISREG_FMT_XYZIL : LISREG_FMT_XYZI;
    int rc = lisreg_transform_cloud(ctx, cloudIn.points.data(), (int)cloudIn.size(), (int)sizeof(PointT), fmt, pose, out.points.data());
    if (rc != LISREG_OK) throw RegistrationError(rc, lisreg_last_error(ctx));
    return out;
}

// ---- the producer of cloud_info ------------------------------------------------------------------------
// PointXYZIRT (src/include/common.h:12-23) and the part of LaserProcessing that turns one sweep into the five clouds of
// cloud_info (src/core/laserProcessing.cpp:467-713: projectPointCloud, cloudExtraction, calculateSmoothness,
// markOccludedPoints, extractFeatures), without the IMU de-skew.
struct alignas(16) PointXYZIRT { float x, y, z, _pad0; float intensity; uint16_t ring; uint16_t _pad1; float time; float _pad2; };
static_assert(sizeof(PointXYZIRT) == 32, "PCL point layout");

struct ExtractedFeatures {             // what assignCouldInfo/publishClouds put into cloud_info (:718-760)
    PointCloud<PointXYZIRT> extractedCloud, cornerCloud, surfaceCloud, sharpCornerCloud, SharpSurfaceCloud;
};

class LaserProcessing {
public:
    lisreg_feature_params params;
    // de-skew state, named as in src/include/laserProcessing.h:105-123
    std::vector<double> imuTime, imuRotX, imuRotY, imuRotZ;
    int    imuPointerCur = 0;
    int    deskewFlag = 1;                  // 1: the cloud has a `time` channel (:158-170)
    bool   imuAvailable = false;            // cloudInfo.imuAvailable
    double timeScanCur = 0.0;
    explicit LaserProcessing(lisreg_ctx* ctx) : ctx_(ctx) { lisreg_default_feature_params(&params); }
    // imuDeskewInfo (laserProcessing.cpp:222-266): integrate the angular velocities of the IMU messages around this scan.
    // stamp[i], (ang_x, ang_y, ang_z)[i] = header stamp and ros-frame angular velocity of message i (after imuConverter).
    void imuDeskewInfo(const double* stamp, const double* ang_x, const double* ang_y, const double* ang_z, int n, double scanCur, double scanEnd) {
        timeScanCur = scanCur; imuAvailable = false; imuPointerCur = 0;
        imuTime.assign((size_t)n + 1, 0.0); imuRotX.assign((size_t)n + 1, 0.0); imuRotY.assign((size_t)n + 1, 0.0); imuRotZ.assign((size_t)n + 1, 0.0);
        for (int i = 0; i < n; ++i) {
            const double t = stamp[i];
            if (t > scanEnd + 0.01) break;
            if (imuPointerCur == 0) { imuRotX[0] = imuRotY[0] = imuRotZ[0] = 0; imuTime[0] = t; ++imuPointerCur; continue; }
            const double dt = t - imuTime[(size_t)imuPointerCur - 1];
            imuRotX[(size_t)imuPointerCur] = imuRotX[(size_t)imuPointerCur - 1] + ang_x[i] * dt;
            imuRotY[(size_t)imuPointerCur] = imuRotY[(size_t)imuPointerCur - 1] + ang_y[i] * dt;
            imuRotZ[(size_t)imuPointerCur] = imuRotZ[(size_t)imuPointerCur - 1] + ang_z[i] * dt;
            imuTime[(size_t)imuPointerCur] = t;
            ++imuPointerCur;
        }
        --imuPointerCur;
        if (imuPointerCur <= 0) return;
        imuAvailable = true;
    }
    ExtractedFeatures process(const PointCloud<PointXYZIRT>& laserCloudIn) {
        ExtractedFeatures f;
        const size_t cap = (size_t)params.n_scan * (size_t)params.horizon_scan;
        PointCloud<PointXYZIRT>* clouds[5] = { &f.extractedCloud, &f.cornerCloud, &f.surfaceCloud, &f.sharpCornerCloud, &f.SharpSurfaceCloud };
        for (auto* c : clouds) c->points.resize(cap);
        lisreg_feature_out o{};
        o.deskewed = f.extractedCloud.points.data();       o.cap_deskewed = (int)cap;
        o.corner = f.cornerCloud.points.data();            o.cap_corner = (int)cap;
        o.surface = f.surfaceCloud.points.data();          o.cap_surface = (int)cap;
        o.corner_sharp = f.sharpCornerCloud.points.data(); o.cap_corner_sharp = (int)cap;
        o.surface_sharp = f.SharpSurfaceCloud.points.data(); o.cap_surface_sharp = (int)cap;
        lisreg_deskew dk{};
        dk.enabled = (deskewFlag == 1 && imuAvailable) ? 1 : 0;       // deskewPoint :429
        dk.imu_pointer_cur = imuPointerCur; dk.imu_time = imuTime.data(); dk.imu_rot_x = imuRotX.data(); dk.imu_rot_y = imuRotY.data();
        dk.imu_rot_z = imuRotZ.data(); dk.time_scan_cur = timeScanCur;
        int rc = lisreg_extract_features_deskew(ctx_, laserCloudIn.points.data(), (int)laserCloudIn.size(), (int)sizeof(PointXYZIRT),
                                                LISREG_FMT_XYZIRT, &params, &dk, &o);
        if (rc != LISREG_OK) throw RegistrationError(rc, lisreg_last_error(ctx_));
        const int n[5] = { o.n_deskewed, o.n_corner, o.n_surface, o.n_corner_sharp, o.n_surface_sharp };
        for (int k = 0; k < 5; ++k) clouds[k]->points.resize((size_t)n[k]);
        return f;
    }
private:
    lisreg_ctx* ctx_;
};

// ---- the compute steps of SubMapManager (src/include/subMap.h) ---------------------------------------------
struct bounds_t { double min_x, min_y, min_z, max_x, max_y, max_z; };     // src/include/subMap.h:32-39
struct centerpoint_t { double x, y, z; };                                  // src/include/subMap.h:11-16

// pcl::search::KdTree<PointT> as SubMapManager uses it (setInputCloud once, k = 1 queries): an HBM-resident grid index.
template <class PointT>
class SearchTree {
public:
    SearchTree(lisreg_ctx* ctx, int slot) : ctx_(ctx), slot_(slot) {}
    void setInputCloud(const PointCloud<PointT>& cloud) {
        int rc = lisreg_map_index_set(ctx_, slot_, cloud.points.data(), (int)cloud.size(), (int)sizeof(PointT), fmt());
        if (rc != LISREG_OK) throw RegistrationError(rc, lisreg_last_error(ctx_));
    }
    int slot() const { return slot_; }
    lisreg_ctx* ctx() const { return ctx_; }
    static int fmt() { return std::is_same<PointT, PointXYZIL>::value ? LISREG_FMT_XYZIL : LISREG_FMT_XYZI; }
private:
    lisreg_ctx* ctx_;
    int slot_;
};

template <class PointT>
class SubMapManager {
public:
    explicit SubMapManager(lisreg_ctx* ctx) : ctx_(ctx) {}
    // subMap.h:131-163
    void get_cloud_bbx(const PointCloud<PointT>& cloud, bounds_t& bound) {
        double b[6];
        check(lisreg_cloud_bounds(ctx_, cloud.points.data(), (int)cloud.size(), (int)sizeof(PointT), SearchTree<PointT>::fmt(), b));
        bound = { b[0], b[1], b[2], b[3], b[4], b[5] };
    }
    // subMap.h:173-181 (host arithmetic)
    static void get_intersection_bbx(const bounds_t& a, const bounds_t& b, bounds_t& out, float pad = 2.0f) {
        out.min_x = std::max(a.min_x, b.min_x) - pad; out.min_y = std::max(a.min_y, b.min_y) - pad; out.min_z = std::max(a.min_z, b.min_z) - pad;
        out.max_x = std::min(a.max_x, b.max_x) + pad; out.max_y = std::min(a.max_y, b.max_y) + pad; out.max_z = std::min(a.max_z, b.max_z) + pad;
    }
    // subMap.h:123-128 get_bound_cpt and :214-228 transform_bbx (host arithmetic; transCur = row-major 3x4 [R|t] in float, the
    // products are float x double like Eigen::Affine3f(i, j) * double in the reference)
    static void get_bound_cpt(const bounds_t& b, centerpoint_t& cp) { cp = { 0.5 * (b.min_x + b.max_x), 0.5 * (b.min_y + b.max_y), 0.5 * (b.min_z + b.max_z) }; }
    static void transform_bbx(const bounds_t& bound_in, const centerpoint_t& cp_in, bounds_t& bound_out, centerpoint_t& cp_out, const float transCur[12]) {
        const bounds_t b = bound_in; const centerpoint_t c = cp_in;
        cp_out.x = transCur[0] * c.x + transCur[1] * c.y + transCur[2] * c.z + transCur[3];
        cp_out.y = transCur[4] * c.x + transCur[5] * c.y + transCur[6] * c.z + transCur[7];
        cp_out.z = transCur[8] * c.x + transCur[9] * c.y + transCur[10] * c.z + transCur[11];
        bound_out.max_x = b.max_x - c.x + cp_out.x; bound_out.max_y = b.max_y - c.y + cp_out.y; bound_out.max_z = b.max_z - c.z + cp_out.z;
        bound_out.min_x = b.min_x - c.x + cp_out.x; bound_out.min_y = b.min_y - c.y + cp_out.y; bound_out.min_z = b.min_z - c.z + cp_out.z;
    }
    // subMap.h:1124-1152
    bool bbx_filter(PointCloud<PointT>& cloud_in_out, const bounds_t& bbx, bool delete_box = false) {
        const double b[6] = { bbx.min_x, bbx.min_y, bbx.min_z, bbx.max_x, bbx.max_y, bbx.max_z };
        int n_out = 0;
        check(lisreg_bbx_filter(ctx_, cloud_in_out.points.data(), (int)cloud_in_out.size(), (int)sizeof(PointT), SearchTree<PointT>::fmt(), b,
                                delete_box ? 1 : 0, cloud_in_out.points.data(), &n_out));
        cloud_in_out.points.resize((size_t)n_out);
        return true;
    }
    // subMap.h:1064-1100; returns false (cloud untouched) for <= 10 points like the reference
    bool map_scan_feature_pts_distance_removal(PointCloud<PointT>& feature_pts, const SearchTree<PointT>& map_kdtree, float center_radius,
                                               float dynamic_dist_thre_min = 3.402823466e38f, float dynamic_dist_thre_max = 3.402823466e38f,
                                               float near_dist_thre = 0.0f) {
        int n_out = 0;
        int rc = lisreg_dynamic_filter(ctx_, map_kdtree.slot(), feature_pts.points.data(), (int)feature_pts.size(), (int)sizeof(PointT),
                                       SearchTree<PointT>::fmt(), center_radius, dynamic_dist_thre_min, dynamic_dist_thre_max, near_dist_thre,
                                       feature_pts.points.data(), &n_out);
        if (rc < 0) throw RegistrationError(rc, lisreg_last_error(ctx_));
        feature_pts.points.resize((size_t)n_out);
        return rc == LISREG_OK;
    }
private:
    void check(int rc) { if (rc != LISREG_OK) throw RegistrationError(rc, lisreg_last_error(ctx_)); }
    lisreg_ctx* ctx_;
};

// ---- pcl::IterativeClosestPoint as the loop-closure code uses it --------------------------------------------
// (src/node/subMapOptmizationNode.cpp:2763-2833: setters, setInputTarget, setInputSource, align, getFitnessScore, hasConverged,
// getFinalTransformation).  Like the reference's `static` objects, an instance carries correspondences_prev_mse_ across align().
template <class PointT>
class IterativeClosestPoint {
public:
    IterativeClosestPoint(lisreg_ctx* ctx, int slot) : tree_(ctx, slot) { lisreg_icp_default_params(0, &prm_); prm_.max_corr_dist = 1.3407807929942596e154; /* sqrt(DBL_MAX), PCL's default */ prm_.max_iters = 10; prm_.transformation_epsilon = 0; prm_.euclidean_fitness_epsilon = -1.7976931348623157e308; }
    void setMaxCorrespondenceDistance(double d) { prm_.max_corr_dist = d; }
    void setMaximumIterations(int n) { prm_.max_iters = n; }
    void setTransformationEpsilon(double e) { prm_.transformation_epsilon = e; }
    void setEuclideanFitnessEpsilon(double e) { prm_.euclidean_fitness_epsilon = e; }
    void setRANSACIterations(int n) { if (n != 0) throw RegistrationError(LISREG_ERR_ARG, "ICP: the reference runs with 0 RANSAC iterations"); }
    void setInputTarget(const PointCloud<PointT>& cloud) { tree_.setInputCloud(cloud); }
    void setInputSource(const PointCloud<PointT>* cloud) { source_ = cloud; }
    void align(PointCloud<PointT>& output, const float* guess = nullptr) {
        if (!source_) throw RegistrationError(LISREG_ERR_ARG, "ICP: no input source");
        output.points.resize(source_->size());
        int rc = lisreg_icp_align(tree_.ctx(), tree_.slot(), source_->points.data(), (int)source_->size(), (int)sizeof(PointT),
                                  SearchTree<PointT>::fmt(), &prm_, guess, &res_, output.points.data());
        if (rc != LISREG_OK) throw RegistrationError(rc, lisreg_last_error(tree_.ctx()));
        prm_.prev_mse = res_.prev_mse;
    }
    bool hasConverged() const { return res_.converged != 0; }
    double getFitnessScore() const { return res_.fitness; }
    const float* getFinalTransformation() const { return res_.final_transform; }     // row-major 4x4
    int nr_iterations() const { return res_.iters; }
private:
    SearchTree<PointT> tree_;
    const PointCloud<PointT>* source_ = nullptr;
    lisreg_icp_params prm_{};
    lisreg_icp_result res_{};
};

// OptimizedICPGN (src/include/registration.h:44-70, src/core/registration.cpp:8-115): same constructor and calls
template <class PointT>
class OptimizedICPGN {
public:
    OptimizedICPGN(lisreg_ctx* ctx, int slot, unsigned max_iterations, float max_correspond_distance)
        : tree_(ctx, slot), max_iterations_(max_iterations), max_correspond_distance_(max_correspond_distance) {}
    bool SetTargetCloud(const PointCloud<PointT>& target_cloud) { tree_.setInputCloud(target_cloud); return true; }
    bool Match(const PointCloud<PointT>& source_cloud, const float predict_pose[16], PointCloud<PointT>& transformed_source_cloud, float result_pose[16]) {
        transformed_source_cloud.points.resize(source_cloud.size());
        int rc = lisreg_icp_gn_match(tree_.ctx(), tree_.slot(), source_cloud.points.data(), (int)source_cloud.size(), (int)sizeof(PointT),
                                     SearchTree<PointT>::fmt(), max_iterations_, max_correspond_distance_, predict_pose, &res_,
                                     transformed_source_cloud.points.data());
        if (rc != LISREG_OK) throw RegistrationError(rc, lisreg_last_error(tree_.ctx()));
        std::memcpy(result_pose, res_.final_transform, sizeof res_.final_transform);
        return true;
    }
    float GetFitnessScore() const { return res_.fitness; }
    bool HasConverged() const { return true; }                      // registration.cpp:110-113
private:
    SearchTree<PointT> tree_;
    unsigned max_iterations_;
    float max_correspond_distance_;
    lisreg_icpgn_result res_{};
};

// ---- SURVEY.md §8 f-3 composite: the sliding local map kept in HBM -------------------------------------------------------
// localMap_t + SubMapManager::insert_local_map (src/include/subMap.h:979-1059) + SubMapOptmizationNode::extractSlidingCloud
// (src/node/subMapOptmizationNode.cpp:1369-1432).  Class order of the arrays: dynamic, pole, ground, building, outlier
// (append_feature, subMap.h:742-753).  The makeSubMapThread loop (:597-755) becomes
//     localMap.extractSlidingCloud(transformTobeSubMapped);  reg.scan2SubMapOptimization(...);  localMap.insert_local_map(cls, pose);
template <class PointT = PointXYZIL>
class LocalMap {
public:
    lisreg_localmap_params params;          // local_map_radius etc. of makeSubMapThread (:603-612) that the path actually reads
    LocalMap(lisreg_ctx* ctx, int map_id = 0) : ctx_(ctx), id_(map_id) {
        lisreg_localmap_default_params(&params);
        int rc = lisreg_localmap_reset(ctx_, id_);
        if (rc != LISREG_OK) throw RegistrationError(rc, lisreg_last_error(ctx_));
    }
    // this->insert_local_map(localMap, currentKeyFrame, ...): the key frame's un-downsampled class clouds + its optimized_pose
    lisreg_localmap_info insert_local_map(const PointCloud<PointT> cls[5], const float optimized_pose[6]) {
        const void* ptr[5]; int n[5];
        for (int k = 0; k < 5; ++k) { ptr[k] = cls[k].points.data(); n[k] = (int)cls[k].size(); }
        lisreg_localmap_info info{};
        int rc = lisreg_localmap_insert(ctx_, id_, ptr, n, (int)sizeof(PointT), fmt(), optimized_pose, &params, &info);
        if (rc != LISREG_OK) throw RegistrationError(rc, lisreg_last_error(ctx_));
        return info;
    }
    // extractSlidingCloud(currentKeyFrame, cur_pose) + both kd-tree setInputCloud calls: the target lands in `target_slot`
    lisreg_localmap_info extractSlidingCloud(const float cur_pose[6], int target_slot = 0) {
        lisreg_localmap_info info{};
        int rc = lisreg_localmap_extract(ctx_, id_, cur_pose, &params, target_slot, &info);
        if (rc != LISREG_OK) throw RegistrationError(rc, lisreg_last_error(ctx_));
        return info;
    }
private:
    static constexpr int fmt() { return std::is_same<PointT, PointXYZIL>::value ? LISREG_FMT_XYZIL : LISREG_FMT_XYZI; }
    lisreg_ctx* ctx_;
    int id_;
};

// Copy #3's side: submap_t + SubMapManager::fisrt_submap / insert_submap (src/include/subMap.h:785-978) and
// SubMapOptmizationNode::extractSubMapCloud (src/node/subMapOptmizationNode.cpp:3976-4081), kept in HBM.  subMapOptmizationThread becomes
//     cur.fisrt_submap(down, pose); cur.insert_submap(down, relative_pose) ...;
//     auto x = SubMap<>::extractSubMapCloud(ctx, pre, cur, transformTobeMapped);          // target installed, sources as device records
//     reg.subMap2SubMapOptimization(x, cloudInfo);                                        // copy #3 on the device records
template <class PointT = PointXYZIL>
class SubMap {
public:
    lisreg_localmap_params params;          // max_num_pts, map_based_dynamic_removal_on, ... of makeSubMapThread (:603-612)
    float submap_pose_6D_optimized[6] = { 0, 0, 0, 0, 0, 0 };
    lisreg_submap_info info{};
    SubMap(lisreg_ctx* ctx, int map_id) : ctx_(ctx), id_(map_id) {
        lisreg_localmap_default_params(&params);
        int rc = lisreg_localmap_reset(ctx_, id_);
        if (rc != LISREG_OK) throw RegistrationError(rc, lisreg_last_error(ctx_));
    }
    int id() const { return id_; }
    // this->fisrt_submap(currentSubMap, currentKeyFrame): the key frame's *_down class clouds as they are; the submap takes its optimized_pose
    void fisrt_submap(const PointCloud<PointT> down[5], const float optimized_pose[6]) {
        for (int k = 0; k < 6; ++k) submap_pose_6D_optimized[k] = optimized_pose[k];
        insert(down, nullptr);
    }
    // this->insert_submap(currentSubMap, currentKeyFrame, ...): the *_down clouds moved by the frame's relative_pose, all five classes
    void insert_submap(const PointCloud<PointT> down[5], const float relative_pose[6]) { insert(down, relative_pose); }
    // extractSubMapCloud() for (preSubMap, curSubMapPtr) under the guess transformTobeMapped; bbx pad 10 m, voxel grids 0.2 / 0.5
    static lisreg_submap_extract_out extractSubMapCloud(lisreg_ctx* ctx, const SubMap& pre, const SubMap& cur, const float transformTobeMapped[6],
                                                        int target_slot = 0) {
        lisreg_submap_extract_out out{};
        int rc = lisreg_submap_extract(ctx, pre.id_, cur.id_, pre.submap_pose_6D_optimized, transformTobeMapped, 10.0f, 0.2f, 0.5f, target_slot, &out);
        if (rc != LISREG_OK) throw RegistrationError(rc, lisreg_last_error(ctx));
        return out;
    }
private:
    void insert(const PointCloud<PointT> down[5], const float* rel) {
        const void* ptr[5]; int n[5];
        for (int k = 0; k < 5; ++k) { ptr[k] = down[k].points.data(); n[k] = (int)down[k].size(); }
        int rc = lisreg_submap_insert(ctx_, id_, ptr, n, (int)sizeof(PointT), fmt(), rel, submap_pose_6D_optimized, &params, &info);
        if (rc != LISREG_OK) throw RegistrationError(rc, lisreg_last_error(ctx_));
    }
    static constexpr int fmt() { return std::is_same<PointT, PointXYZIL>::value ? LISREG_FMT_XYZIL : LISREG_FMT_XYZI; }
    lisreg_ctx* ctx_;
    int id_;
};

// The odometry node's multi-frame target (odomEstimationNode.cpp, USING_MULTI_FRAME_TARGET), device-resident: replaces
// laserCloud{Corner,Surf}Vec + the concatenation loop + the two VoxelGrid filters of laserCloudInfoHandler (:185-207) and the
// transformPointCloud / push_back / erase of saveKeyFrames (:452-467).  The frame loop becomes
//     keyframes.extractTarget(mappingCornerLeafSize, mappingSurfLeafSize);  reg.scan2SubMapOptimization(...);
//     if (key frame) keyframes.saveKeyFrame(*laserCloudCornerLast, *laserCloudSurfLast, reg.transformTobeMapped);
template <class PointT = PointType>
class KeyframeTarget {
public:
    explicit KeyframeTarget(lisreg_ctx* ctx, int ring_id = 0, int max_keep = 19) : ctx_(ctx), id_(ring_id), keep_(max_keep) {
        int rc = lisreg_keyframes_reset(ctx_, id_);
        if (rc != LISREG_OK) throw RegistrationError(rc, lisreg_last_error(ctx_));
    }
    // saveKeyFrames(): the frame's FULL feature clouds (sensor frame) + thisPose6D
    int saveKeyFrame(const PointCloud<PointT>& cornerLast, const PointCloud<PointT>& surfLast, const float pose[6]) {
        lisreg_keyframes_info info{};
        int rc = lisreg_keyframes_push(ctx_, id_, cornerLast.points.data(), (int)cornerLast.size(), surfLast.points.data(), (int)surfLast.size(),
                                       (int)sizeof(PointT), fmt(), pose, keep_, &info);
        if (rc != LISREG_OK) throw RegistrationError(rc, lisreg_last_error(ctx_));
        return info.n_keyframes;
    }
    // laserCloud{Corner,Surf}FromMapDS + both kd-tree setInputCloud calls: the target lands in `target_slot`
    lisreg_keyframes_info extractTarget(float cornerLeaf, float surfLeaf, int target_slot = 0) {
        lisreg_keyframes_info info{};
        int rc = lisreg_keyframes_target(ctx_, id_, cornerLeaf, surfLeaf, target_slot, &info);
        if (rc != LISREG_OK) throw RegistrationError(rc, lisreg_last_error(ctx_));
        return info;
    }
private:
    static constexpr int fmt() { return std::is_same<PointT, PointXYZIL>::value ? LISREG_FMT_XYZIL : LISREG_FMT_XYZI; }
    lisreg_ctx* ctx_;
    int id_, keep_;
};

// updateInitialGuess with neither IMU nor odometry (odomEstimationNode.cpp:351-392): constant-velocity pose guess
inline void updateInitialGuess(const float lastTransformTobeMapped[6], float transformTobeMapped[6]) {
    float g[6];
    lisreg_predict_pose(lastTransformTobeMapped, transformTobeMapped, g);
    for (int k = 0; k < 6; ++k) transformTobeMapped[k] = g[k];
}

// updateInitialGuess as a whole (odomEstimationNode.cpp:297-419 = NodeCopy::Odom; subMapOptmizationNode.cpp:896-1032 = NodeCopy::SubMap):
// the function-local statics of the reference are this object's state.  One instance per node; call once per sweep BEFORE
// scan2SubMapOptimization, exactly where the reference calls updateInitialGuess().
enum class NodeCopy { Odom = 0, SubMap = 1 };
class InitialGuess {
public:
    explicit InitialGuess(NodeCopy copy, bool useImuHeadingInitialization = false) : copy_(copy), heading_(useImuHeadingInitialization) {
        lisreg_guess_state_init(&st_);
    }
    // transformTobeMapped is updated in place; transPredictionMapped (may be null) where the reference assigns it
    void updateInitialGuess(const cloud_info& cloudInfo, float transformTobeMapped[6], float* transPredictionMapped = nullptr) {
        lisreg_guess_input in{ cloudInfo.odomAvailable ? 1 : 0, cloudInfo.imuAvailable ? 1 : 0,
                               cloudInfo.imuRollInit, cloudInfo.imuPitchInit, cloudInfo.imuYawInit,
                               cloudInfo.initialGuessX, cloudInfo.initialGuessY, cloudInfo.initialGuessZ,
                               cloudInfo.initialGuessRoll, cloudInfo.initialGuessPitch, cloudInfo.initialGuessYaw };
        lisreg_update_initial_guess((int)copy_, heading_ ? 1 : 0, &in, &st_, transformTobeMapped, transPredictionMapped);
    }
    const lisreg_guess_state& state() const { return st_; }
private:
    NodeCopy copy_; bool heading_; lisreg_guess_state st_;
};

// The candidate loop of detectLoopClosureForSubMap (subMapOptmizationNode.cpp:2776-2840) as one call: candidate k = (target slot set by
// IterativeClosestPoint::setInputTarget / lisreg_map_index_set, source cloud, guess).  chain = true reproduces the reference's `static`
// ICP object (correspondences_prev_mse_ carried from one align() to the next, :2763).
template <class PointT>
inline std::vector<lisreg_icp_result> alignLoopCandidates(lisreg_ctx* ctx, const std::vector<lisreg_icp_item>& candidates,
                                                          const lisreg_icp_params& params, bool chain = true) {
    std::vector<lisreg_icp_result> res(candidates.size());
    const int fmt = std::is_same<PointT, PointXYZIL>::value ? LISREG_FMT_XYZIL : LISREG_FMT_XYZI;
    const int rc = lisreg_icp_align_batch(ctx, candidates.data(), (int)candidates.size(), (int)sizeof(PointT), fmt, &params, chain ? 1 : 0, res.data());
    if (rc != LISREG_OK) throw RegistrationError(rc, lisreg_last_error(ctx));
    return res;
}

}  // namespace lis_slam
