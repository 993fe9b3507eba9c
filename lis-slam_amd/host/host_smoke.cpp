// host_smoke.cpp — C++ caller of liblisreg through the reference-shaped host mirror (plain g++, no HIP headers).
// With a GPU: registers a small synthetic scene (two walls + floor + poles) and checks the pose moved toward truth.
// Without a GPU: verifies the loud failure path (no CPU fallback) and exits 0.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

#include "lis_slam_registration.hpp"

using namespace lis_slam;

static void add(PointCloud<PointType>& c, float x, float y, float z) { PointType p{}; p.x = x; p.y = y; p.z = z; c.push_back(p); }

int main()
{
    if (lisreg_device_count() == 0) {
        try { Scan2SubMapRegistration<> reg(Variant::Odom); }
        catch (const RegistrationError& e) { std::printf("no HIP device: constructor failed loudly as designed (%d: %s)\n", e.code, e.what()); return 0; }
        std::printf("ERROR: context creation succeeded without a device\n");
        return 1;
    }
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> U(0.f, 1.f);
    std::normal_distribution<float> N(0.f, 0.01f);
    PointCloud<PointType> mapCorner, mapSurf, corner, surf;
    for (int i = 0; i < 40000; ++i) {            // floor z=0, wall x=10, wall y=-8
        float u = U(rng) * 40 - 20, v = U(rng) * 40 - 20, h = U(rng) * 6;
        add(mapSurf, u + N(rng), v + N(rng), N(rng));
        if (i % 2 == 0) add(mapSurf, 10 + N(rng), v + N(rng), h);
        else add(mapSurf, u + N(rng), -8 + N(rng), h);
    }
    const float poles[6][2] = { { 3, 4 }, { -5, 6 }, { 7, -3 }, { -6, -4 }, { 1, -6 }, { -2, 9 } };
    for (int i = 0; i < 6000; ++i) { int k = i % 6; float a = U(rng) * 6.2831853f; add(mapCorner, poles[k][0] + 0.1f * std::cos(a) + N(rng), poles[k][1] + 0.1f * std::sin(a) + N(rng), U(rng) * 5); }
    // source = subsample of the map moved by the inverse of a known pose (yaw 0.03, t = (0.2,-0.15,0.05))
    const float yaw = 0.03f, tx = 0.2f, ty = -0.15f, tz = 0.05f, cy = std::cos(yaw), sy = std::sin(yaw);
    auto inv = [&](const PointType& p) { PointType q{}; float x = p.x - tx, y = p.y - ty; q.x = cy * x + sy * y; q.y = -sy * x + cy * y; q.z = p.z - tz; return q; };
    for (size_t i = 0; i < mapSurf.size(); i += 7) surf.push_back(inv(mapSurf.points[i]));
    for (size_t i = 0; i < mapCorner.size(); i += 5) corner.push_back(inv(mapCorner.points[i]));

    Scan2SubMapRegistration<> reg(Variant::Odom);
    // the reference's per-frame order: voxel-grid the local map and the incoming features, then register
    // (odomEstimationNode.cpp:196-201, 272-277, 596)
    VoxelGrid<> downSizeFilterCorner(reg.handle()), downSizeFilterSurf(reg.handle());
    downSizeFilterCorner.setLeafSize(0.2f, 0.2f, 0.2f);
    downSizeFilterSurf.setLeafSize(0.4f, 0.4f, 0.4f);
    PointCloud<PointType> mapCornerDS, mapSurfDS, cornerDS, surfDS;
    downSizeFilterCorner.setInputCloud(&mapCorner); downSizeFilterCorner.filter(mapCornerDS);
    downSizeFilterSurf.setInputCloud(&mapSurf);     downSizeFilterSurf.filter(mapSurfDS);
    downSizeFilterCorner.setInputCloud(&corner);    downSizeFilterCorner.filter(cornerDS);
    downSizeFilterSurf.setInputCloud(&surf);        downSizeFilterSurf.filter(surfDS);
    std::printf("VoxelGrid: map %zu/%zu -> %zu/%zu, scan %zu/%zu -> %zu/%zu\n", mapCorner.size(), mapSurf.size(), mapCornerDS.size(),
                mapSurfDS.size(), corner.size(), surf.size(), cornerDS.size(), surfDS.size());
    mapCorner = mapCornerDS; mapSurf = mapSurfDS; corner = cornerDS; surf = surfDS;
    reg.setInputTarget(mapCorner, mapSurf);
    cloud_info info;
    int rc = reg.scan2SubMapOptimization(corner, surf, info);
    const float* T = reg.transformTobeMapped;
    std::printf("rc=%d iterCount=%d deltaR=%g deltaT=%g isDegenerate=%d nSel=%d T=[%g %g %g %g %g %g]\n", rc, reg.iterCount, reg.deltaR,
                reg.deltaT, (int)reg.isDegenerate, reg.laserCloudSelNum, T[0], T[1], T[2], T[3], T[4], T[5]);
    {   // host-inclusive latency of one registration through the mirror (pageable host clouds in, pose out), same problem re-run
        std::vector<double> us;
        for (int rep = 0; rep < 40; ++rep) {
            for (int k = 0; k < 6; ++k) reg.transformTobeMapped[k] = 0.f;
            const auto t0 = std::chrono::steady_clock::now();
            reg.scan2SubMapOptimization(corner, surf, info);
            us.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
        }
        std::sort(us.begin(), us.end());
        std::printf("latency: scan2SubMapOptimization (%zu + %zu source points vs %zu + %zu, %d iterations): median %.1f us, min %.1f us\n",
                    corner.size(), surf.size(), mapCorner.size(), mapSurf.size(), reg.iterCount + 1, us[us.size() / 2], us[0]);
    }
    bool ok = rc == LISREG_OK && std::fabs(T[2] - yaw) < 5e-3f && std::fabs(T[3] - tx) < 2e-2f && std::fabs(T[4] - ty) < 2e-2f && std::fabs(T[5] - tz) < 2e-2f;
    // §8 f-3: crop the map to the padded intersection with the scan's box, then drop scan points that sit on the map
    // (subMapOptmizationNode.cpp:1392-1405, subMap.h:889-899)
    SubMapManager<PointType> mgr(reg.handle());
    bounds_t bScan, bMap, bInter;
    mgr.get_cloud_bbx(surf, bScan); mgr.get_cloud_bbx(mapSurf, bMap);
    SubMapManager<PointType>::get_intersection_bbx(bScan, bMap, bInter, 2.0f);
    PointCloud<PointType> cropped = mapSurf;
    mgr.bbx_filter(cropped, bInter);
    SearchTree<PointType> tree(reg.handle(), 0);
    tree.setInputCloud(mapSurf);
    PointCloud<PointType> moved = transformPointCloud(reg.handle(), surf, T), kept = moved;
    bool applied = mgr.map_scan_feature_pts_distance_removal(kept, tree, 100.f, 0.3f, 1.0f, 0.05f);
    std::printf("bbx_filter: map %zu -> %zu; distance_removal(applied=%d): scan %zu -> %zu\n", mapSurf.size(), cropped.size(), (int)applied,
                moved.size(), kept.size());
    ok = ok && applied && cropped.size() <= mapSurf.size() && cropped.size() > 0 && kept.size() < moved.size();
    // §8 f-4: ICP of the un-registered scan against the map, loop-closure settings (subMapOptmizationNode.cpp:2763-2769)
    IterativeClosestPoint<PointType> icp(reg.handle(), 1);
    icp.setMaxCorrespondenceDistance(10); icp.setMaximumIterations(30); icp.setTransformationEpsilon(1e-4);
    icp.setEuclideanFitnessEpsilon(1e-4); icp.setRANSACIterations(0);
    icp.setInputTarget(mapSurf);
    icp.setInputSource(&surf);
    PointCloud<PointType> unused_result;
    icp.align(unused_result);
    const float* Fm = icp.getFinalTransformation();
    std::printf("ICP: converged=%d iterations=%d fitness=%g t=[%g %g %g] yaw=%g\n", (int)icp.hasConverged(), icp.nr_iterations(),
                icp.getFitnessScore(), Fm[3], Fm[7], Fm[11], std::atan2(Fm[4], Fm[0]));
    ok = ok && icp.hasConverged() && std::fabs(Fm[3] - tx) < 5e-2f && std::fabs(Fm[7] - ty) < 5e-2f && std::fabs(std::atan2(Fm[4], Fm[0]) - yaw) < 5e-3f;
    // OptimizedICPGN (registration.cpp:19-115) on the same pair
    OptimizedICPGN<PointType> gn(reg.handle(), 2, 15, 4.0f);
    gn.SetTargetCloud(mapSurf);
    const float predict[16] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1 };
    float result_pose[16];
    PointCloud<PointType> gnOut;
    gn.Match(surf, predict, gnOut, result_pose);
    std::printf("OptimizedICPGN: fitness=%g t=[%g %g %g] yaw=%g\n", gn.GetFitnessScore(), result_pose[3], result_pose[7], result_pose[11],
                std::atan2(result_pose[4], result_pose[0]));
    ok = ok && std::fabs(result_pose[3] - tx) < 5e-2f && std::fabs(result_pose[7] - ty) < 5e-2f && std::fabs(std::atan2(result_pose[4], result_pose[0]) - yaw) < 5e-3f;
    bounds_t bMoved; centerpoint_t cMap, cMoved;
    SubMapManager<PointType>::get_bound_cpt(bMap, cMap);
    SubMapManager<PointType>::transform_bbx(bMap, cMap, bMoved, cMoved, result_pose);       // rows 0..2 of the 4x4 = [R|t]
    ok = ok && std::fabs((bMoved.max_x - bMoved.min_x) - (bMap.max_x - bMap.min_x)) < 1e-9;
    // §8 f-3 composite: the device-resident sliding local map (insert_local_map / extractSlidingCloud), labelled clouds
    {
        Scan2SubMapRegistration<PointXYZIL> reg2(Variant::SubMap);
        LocalMap<PointXYZIL> localMap(reg2.handle(), 0);
        PointCloud<PointXYZIL> cls[5];                       // dynamic, pole, ground, building, outlier
        for (size_t i = 0; i < mapSurf.size(); ++i) { PointXYZIL p{}; p.x = mapSurf.points[i].x; p.y = mapSurf.points[i].y; p.z = mapSurf.points[i].z; p.label = p.z < 0.05f ? 9 : 13; cls[p.label == 9 ? 2 : 3].push_back(p); }
        for (size_t i = 0; i < mapCorner.size(); ++i) { PointXYZIL p{}; p.x = mapCorner.points[i].x; p.y = mapCorner.points[i].y; p.z = mapCorner.points[i].z; p.label = 18; cls[1].push_back(p); }
        const float pose0[6] = { 0, 0, 0, 0, 0, 0 };
        lisreg_localmap_info inf = localMap.insert_local_map(cls, pose0);
        const float cur[6] = { 0, 0, 0.01f, 0.1f, 0, 0 };
        lisreg_localmap_info ex = localMap.extractSlidingCloud(cur, 0);           // leaves the registration target in slot 0
        std::printf("LocalMap: inserted %d points, bound x [%g, %g]; extract -> corner %d, surf %d\n", inf.feature_point_num, inf.bound[0],
                    inf.bound[3], ex.n_target_corner, ex.n_target_surf);
        ok = ok && inf.feature_point_num == (int)(mapSurf.size() + mapCorner.size()) && ex.n_target_corner > 0 && ex.n_target_surf > 0 &&
             ex.n_target_surf <= (int)mapSurf.size();
    }
    // the odometry node's key-frame target (saveKeyFrames + the target assembly of laserCloudInfoHandler), kept in HBM
    {
        KeyframeTarget<PointType> keyframes(reg.handle(), 0, 2);
        const float p0[6] = { 0, 0, 0, 0, 0, 0 }, p1[6] = { 0, 0, 0.02f, 0.5f, 0.1f, 0 }, p2[6] = { 0, 0, 0.04f, 1.0f, 0.2f, 0 };
        keyframes.saveKeyFrame(mapCorner, mapSurf, p0);
        keyframes.saveKeyFrame(mapCorner, mapSurf, p1);
        const int kept = keyframes.saveKeyFrame(mapCorner, mapSurf, p2);             // max_keep = 2: the oldest frame is dropped
        lisreg_keyframes_info ti = keyframes.extractTarget(0.2f, 0.4f, 0);
        std::printf("KeyframeTarget: %d frames kept; target corner %d, surf %d\n", kept, ti.n_target_corner, ti.n_target_surf);
        ok = ok && kept == 2 && ti.n_keyframes == 2 && ti.n_target_surf > 0 && ti.n_target_surf <= 2 * (int)mapSurf.size();
    }
    // loop-closure verification: the candidate loop as one call (detectLoopClosureForSubMap, :2776-2840) — two candidates, same target
    {
        std::vector<lisreg_icp_item> cand(2);
        const float g2[16] = { 1, 0, 0, 0.05f, 0, 1, 0, -0.05f, 0, 0, 1, 0, 0, 0, 0, 1 };
        cand[0] = lisreg_icp_item{ surf.points.data(), (int)surf.size(), 2, nullptr };       // slot 2: OptimizedICPGN::SetTargetCloud above
        cand[1] = lisreg_icp_item{ surf.points.data(), (int)surf.size(), 2, g2 };
        lisreg_icp_params ip; lisreg_icp_default_params(0, &ip);
        std::vector<lisreg_icp_result> rr = alignLoopCandidates<PointType>(reg.handle(), cand, ip, true);
        std::printf("alignLoopCandidates: fitness %g / %g, iterations %d / %d\n", rr[0].fitness, rr[1].fitness, rr[0].iters, rr[1].iters);
        ok = ok && rr[0].converged && rr[1].converged && std::fabs(rr[0].final_transform[3] - tx) < 5e-2f && std::fabs(rr[1].final_transform[3] - tx) < 5e-2f;
    }
    // updateInitialGuess with its statics (odomEstimationNode.cpp:297-419): first call = IMU attitude, then odometry increments
    {
        InitialGuess guess(NodeCopy::Odom);
        cloud_info ci; ci.imuAvailable = true; ci.odomAvailable = true; ci.imuRollInit = 0.01f; ci.imuPitchInit = -0.02f; ci.imuYawInit = 0.3f;
        float T[6] = { 0, 0, 0, 0, 0, 0 }, pred[6] = { 0, 0, 0, 0, 0, 0 };
        guess.updateInitialGuess(ci, T);                                         // first: roll / pitch from the IMU, yaw 0
        ok = ok && T[0] == 0.01f && T[1] == -0.02f && T[2] == 0.f;
        ci.initialGuessX = 1.f; guess.updateInitialGuess(ci, T);                 // first odometry message: recorded (+ IMU increment: none)
        ci.initialGuessX = 2.f; guess.updateInitialGuess(ci, T, pred);           // second: the increment (1 m along x of the guess frame) applied
        std::printf("updateInitialGuess: T = [%g %g %g %g %g %g]\n", T[0], T[1], T[2], T[3], T[4], T[5]);
        ok = ok && std::fabs(std::sqrt(T[3] * T[3] + T[4] * T[4] + T[5] * T[5]) - 1.f) < 1e-4f && pred[3] == T[3];
    }
    std::printf(ok ? "host_smoke ok\n" : "host_smoke FAILED\n");
    return ok ? 0 : 1;
}
