/*
 * lisreg_oracle.h — CPU restatement (plain C) of LIS-SLAM's scan-to-submap registration hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library, and only as the checker / the timed CPU baseline.
 * liblisreg.so never links, loads or calls it.
 *
 * PARITY UNPINNED: the reference ships no tests, fixtures or golden vectors for this path and cannot be
 * compiled here (ROS, PCL/FLANN, OpenCV, Eigen absent — SURVEY.md §8c), so this restatement is pinned only by
 * (a) operation-for-operation fidelity to the cited reference lines, (b) cross-checks against independent
 * library routines (scipy cKDTree, numpy.linalg) in tests/, and (c) an independent numpy mirror
 * (oracle/lisreg_numpy.py) whose outputs are committed under tests/golden/.
 *
 * Third-party semantics restated from their published algorithms (not under /root/reference; versions implied
 * by ROS Melodic: PCL 1.8.1, FLANN 1.9.1, OpenCV 3.2.0, Eigen 3.3.4):
 *   pcl::KdTreeFLANN::nearestKSearch  -> exact k-NN, squared L2 in float, ascending     (orc_kdtree_*)
 *   cv::eigen (symmetric, CV_32F)     -> Jacobi, eigenvalues descending, eigenvectors in rows (orc_eigen_sym)
 *   Eigen colPivHouseholderQr().solve -> column-pivoted Householder QR least squares     (orc_lstsq5x3)
 *   cv::solve(..., DECOMP_QR)         -> Householder QR solve in float                   (orc_solve6)
 *   cv::Mat::inv() (DECOMP_LU)        -> LU with partial pivoting in float               (orc_inv6)
 *   cv GEMM for CV_32F                -> float inputs, double accumulation, float result (orc_normal_equations)
 *   pcl::getTransformation            -> ZYX Euler to affine                             (orc_pose_to_matrix)
 *   tf::Quaternion setRPY/slerp, tf::Matrix3x3::getRPY                                   (orc_transform_update)
 *   pcl::IterativeClosestPoint + Eigen::umeyama                                          (orc_icp_align, orc_umeyama)
 */
#ifndef LISREG_ORACLE_H_
#define LISREG_ORACLE_H_

#include "../include/lisreg.h"   /* shares the ABI structs (params, imu, stats) so tests read symmetrically */

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_kdtree orc_kdtree;

/* common.cpp:54-57 */
void orc_pose_to_matrix(const float T[6], float M[12]);

/* exact kNN over xyz[n][3] (float). leaf_size 15 mirrors FLANN KDTreeSingleIndexParams(15). */
orc_kdtree* orc_kdtree_build(const float* xyz, int n, int leaf_size);
void        orc_kdtree_free(orc_kdtree* t);
/* returns number found (min(k,n)); idx/sqd ascending by distance */
int         orc_kdtree_knn(const orc_kdtree* t, const float q[3], int k, int* idx, float* sqd);
/* brute-force reference for tests */
int         orc_bruteforce_knn(const float* xyz, int n, const float q[3], int k, int* idx, float* sqd);

/* Jacobi eigen-decomposition of a symmetric n x n float matrix (n <= 6): w descending, V rows = eigenvectors */
void orc_eigen_sym(const float* A, int n, float* w, float* V);
/* least squares A(5x3) x = b(5) by column-pivoted Householder QR */
void orc_lstsq5x3(const float A[15], const float b[5], float x[3]);
/* 6x6 solve by Householder QR; returns 0 if singular (x zeroed) */
int  orc_solve6(const float A[36], const float b[6], float x[6]);
/* 6x6 inverse by LU with partial pivoting; returns 0 if singular (out zeroed, as cv::invert does) */
int  orc_inv6(const float A[36], float out[36]);

/* cornerOptimization body for one point (odomEstimationNode.cpp:657-742). nb = 5 neighbours xyz, ascending.
 * psel = transformed point. w = label weight (1 when unused). Returns 1 and coeff[4] if accepted. */
int  orc_corner_coeff(const float nb[15], const float psel[3], float w, const lisreg_params* p, float coeff[4]);
/* surfOptimization body for one point (odomEstimationNode.cpp:776-821). */
int  orc_surf_coeff(const float nb[15], const float psel[3], float w, const lisreg_params* p, float coeff[4]);
/* one Jacobian row + rhs (LMOptimization :889-915): ori = UNtransformed source point */
void orc_jacobian_row(const float T[6], const float ori[3], const float coeff[4], float row[6], float* b);

/* transformUpdate (odomEstimationNode.cpp:976-1006) */
void orc_transform_update(const lisreg_params* p, const lisreg_imu* imu, float T[6]);

/* Whole scan2SubMapOptimization().  Clouds use the ABI's host layouts (LISREG_FMT_XYZI / _XYZIL).
 * degenerate: in/out isDegenerate member.  trace: NULL or max_trace*LISREG_TRACE_STRIDE floats.
 * n_threads: OpenMP threads for the three hot loops (1 = the reference as built).
 * use_kdtree: 1 = kd-tree (the timed baseline, includes the two builds), 0 = brute force (small tests). */
int  orc_align(const void* tgt_corner, int n_tc, const void* tgt_surf, int n_ts,
               const void* src_corner, int n_sc, const void* src_surf, int n_ss,
               int stride_bytes, int fmt,
               const lisreg_params* params, const lisreg_imu* imu,
               float T[6], int* degenerate, lisreg_stats* stats,
               float* trace, int max_trace, int n_threads, int use_kdtree);

/* Single-iteration building block for kernel-level parity: per-point accept flag + coeff for one stage.
 * kind 0 = corner, 1 = surf.  flags[n_src] (0/1), coeffs[n_src][4]. */
void orc_stage_coeffs(int kind, const void* tgt, int n_t, const void* src, int n_s, int stride_bytes, int fmt,
                      const lisreg_params* params, const float T[6], unsigned char* flags, float* coeffs);

/* ---- §8 f-1: the step immediately before the registration on both inputs ----------------------------------- */
/* pcl::VoxelGrid<PointT>::filter with default settings (call sites odomEstimationNode.cpp:196-201, 272-277;
 * subMap.h:1207-1249), restated from PCL 1.8.1 filters/impl/voxel_grid.hpp + common/impl/accumulators.hpp:
 * getMinMax3D -> min_b/div_b from floor(min * inverse_leaf) -> per point idx = ijk0 + ijk1*div0 + ijk2*div0*div1 with
 * ijk = (int)(floor(p * inverse_leaf) - (float)min_b) -> sort by idx -> per voxel CentroidPoint: xyz and intensity are
 * float sums / n, label = most frequent (smallest label on ties) -> output in ascending idx.  PCL's std::sort is
 * unstable, so the summation order inside a voxel is implementation-defined there; the restatement fixes it to
 * ascending input index.  Returns 0, or 3 when dx*dy*dz would overflow int32 (PCL warns and copies the input).
 * `out` has room for n points of the input layout; label written only for LISREG_FMT_XYZIL. */
int  orc_voxel_grid(const void* in, int n, int stride_bytes, int fmt, float leaf, void* out, int* n_out);
/* transformPointCloud (src/core/common.cpp:112-173 and the PointXYZIL overload): p' = R(T) p + t, other fields copied. */
void orc_transform_cloud(const void* in, int n, int stride_bytes, int fmt, const float T[6], void* out);

/* ---- §8 f-2: range-image projection + feature extraction (src/core/laserProcessing.cpp:467-713) ---------------- */
/* cloud: PointXYZIRT host structs.  Outputs are index lists into the INPUT cloud (the points themselves are copies):
 * deskewed[n_deskewed] = extractedCloud in row-major pixel order, and the four feature lists in the reference's push
 * order.  All output arrays need n_scan*horizon_scan ints.  counts[5] = {deskewed, corner, surface, corner_sharp,
 * surface_sharp}.  Defined behaviour where the reference has none: the per-frame arrays are zero-initialised every call
 * (cloudSmoothness[i].ind = i), the +-5 neighbour accesses are bounds-checked, equal curvatures sort by index. */
void orc_extract_features(const void* cloud, int n, int stride_bytes, const lisreg_feature_params* p,
                          int* deskewed, int* corner, int* surface, int* corner_sharp, int* surface_sharp, int counts[5]);

/* deskewPoint / findRotation (src/core/laserProcessing.cpp:368-399, 427-462) for the pixel-owning points idx[m] (input
 * indices); xyz_out[m][3].  Float arithmetic as in the reference: pcl::getTransformation, Eigen's cofactor 3x3 inverse,
 * the 3x3 product, then the row-by-row point transform. */
void orc_deskew_points(const void* cloud, int stride_bytes, const lisreg_deskew* dk, const int* idx, int m, float* xyz_out);

/* categoryMapping (src/node/semanticFusionNode.cpp:173-189): class of every point (0 dynamic, 1 ground, 2 building,
 * 3 pole, 4 outlier) from using_label[label & 31]; the five output clouds are the stable partition by class. */
void orc_semantic_classes(const void* cloud, int n, int stride_bytes, const unsigned int using_label[32], unsigned char* cls);

/* ---- §8 f-3: local-map maintenance filters (src/include/subMap.h) --------------------------------------------- */
void orc_cloud_bounds(const void* cloud, int n, int stride_bytes, double bounds[6]);                      /* :131-163 */
void orc_bbx_filter(const void* cloud, int n, int stride_bytes, const double bounds[6], int delete_box,
                    int* keep, int* n_keep);                                                              /* :1124-1152 */
int  orc_dynamic_filter(const void* map, int n_map, const void* cloud, int n, int stride_bytes, float center_radius,
                        float dist_thre_min, float dist_thre_max, float near_dist_thre, int* keep, int* n_keep); /* :1064-1100 */
void orc_nearest(const void* map, int n_map, const void* query, int n, int stride_bytes, float max_dist, int* idx, float* sqd);

/* ---- §8 f-4: pcl::IterativeClosestPoint (PCL 1.8.1 icp.hpp / correspondence_estimation.hpp /
 * default_convergence_criteria.hpp / transformation_estimation_svd.hpp, restated; call sites in lisreg.h) ------------ */
/* float_sums: 1 = means by a sequential float sum like Eigen (order-dependent, ~1e-4..1e-3 m of noise at 1e5 points),
 * 0 = means accumulated in double (the parity target of the GPU's parallel reduction) */
void orc_umeyama(const float* src_xyz, const float* dst_xyz, int n, int float_sums, float T[16]);
void orc_icp_align(const void* target, int n_t, const void* source, int n_s, int stride_bytes, const lisreg_icp_params* prm,
                   const float* guess /* 16 or NULL */, int float_sums, lisreg_icp_result* res);

/* OptimizedICPGN::Match + GetFitnessScore (src/core/registration.cpp:19-115; Sophus SO3::exp from src/sophus/so3.hpp:279-312) */
void orc_icp_gn(const void* target, int n_t, const void* source, int n_s, int stride_bytes, unsigned max_iterations,
                float max_correspond_distance, const float predict_pose[16], int float_sums, lisreg_icpgn_result* res);

#ifdef __cplusplus
}
#endif
#endif
