/*
 * lisreg.h — C ABI of liblisreg.so: MI355X-native (gfx950, HIP) LOAM-style scan-to-submap registration.
 *
 * This header is the drop-in boundary for ONE hot path of QingzhiWang/LIS-SLAM: the private member
 * functions scan2SubMapOptimization() / subMap2SubMapOptimization() and everything they call
 *   copy #1  src/node/odomEstimationNode.cpp:596-1006      (plain PointXYZI, tau=1.0, 15 iters)
 *   copy #2  src/node/subMapOptmizationNode.cpp:1509-2001  (PointXYZIL, label weights, tau=2.0, 20 iters)
 *   copy #3  src/node/subMapOptmizationNode.cpp:4485-4977  (as #2, 30 iters, no IMU blend)
 * The reference has no FFI / plugin layer for this path; this header creates the seam.
 * Each entry point below cites the reference lines it replaces.  Plain pointers and sizes only; no C++ or
 * torch types; never throws; every function returns an int status (0 = OK) unless stated otherwise.
 *
 * Conventions
 *   pose      T[6] = {roll, pitch, yaw, x, y, z}, float32 — the reference's transformTobeMapped[6]
 *             (odomEstimationNode.cpp:66).  M(T) = pcl::getTransformation(x,y,z,roll,pitch,yaw) =
 *             Rz(yaw)*Ry(pitch)*Rx(roll) + t (src/core/common.cpp:54-57).
 *   clouds    host clouds are arrays of PCL point structs as the reference holds them
 *             (`cloud->points.data()`), i.e. stride 32 B, 16-B aligned: PointXYZI = {x,y,z,pad,intensity,pad*3},
 *             PointXYZIL = {x,y,z,pad,intensity,uint16 label,pad} (src/include/common.h:9,25-35).  Any
 *             stride >= 12 is accepted; `fmt` says where (whether) a label lives.
 *   device    device-resident clouds are arrays of lisreg_dpoint (16 B): x,y,z + 32-bit payload whose low
 *             16 bits are the label (0 when unlabelled).
 */
#ifndef LISREG_H_
#define LISREG_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LISREG_VERSION 1

/* ---- status codes (error conventions: SURVEY.md §8b "Error conventions") -------------------------------- */
#define LISREG_OK                        0
#define LISREG_NOT_ENOUGH_FEATURES       1   /* guard at odomEstimationNode.cpp:598,623-625 failed: T untouched */
#define LISREG_TOO_FEW_CORRESPONDENCES   2   /* every iteration hit the `< 50` early return (:870-872): T untouched */
#define LISREG_LEAF_TOO_SMALL            3   /* VoxelGrid: dx*dy*dz overflows int32 — PCL warns and copies the input  */
#define LISREG_ERR_ARG                  -1
#define LISREG_ERR_HIP                  -2
#define LISREG_ERR_NO_TARGET            -3
#define LISREG_ERR_NOMEM                -4
#define LISREG_ERR_COMM                 -5

/* ---- cloud formats --------------------------------------------------------------------------------------- */
#define LISREG_FMT_XYZI    0   /* pcl::PointXYZI   : floats x@0 y@4 z@8, intensity@16                        */
#define LISREG_FMT_XYZIL   1   /* PointXYZIL       : as XYZI + uint16 label@20   (common.h:25-35)            */
#define LISREG_FMT_DEVICE  2   /* pointer is a DEVICE pointer to lisreg_dpoint[n] (stride ignored)           */
#define LISREG_FMT_DEVICE_XYZI 4 /* the same, but the 32-bit payload is a float intensity, not a label: only where a function says so
                                  * (lisreg_voxel_downsample averages it like PCL's PointXYZI centroid instead of taking a label vote;
                                  * lisreg_keyframes_push remembers it for the ring's voxel grids)                                      */

typedef struct lisreg_dpoint {   /* 16-B device record (SURVEY.md §8d "Device record")                        */
    float    x, y, z;
    uint32_t payload;            /* low 16 bits: semantic label (RangeNet learning class 0..19)                */
} lisreg_dpoint;

/* ---- the three reference call sites ---------------------------------------------------------------------- */
#define LISREG_VARIANT_ODOM      1   /* OdomEstimationNode::scan2SubMapOptimization      odomEstimationNode.cpp:596  */
#define LISREG_VARIANT_KEYFRAME  2   /* SubMapOdometryNode::scan2SubMapOptimization      subMapOptmizationNode.cpp:1509 */
#define LISREG_VARIANT_SUBMAP    3   /* SubMapOptmizationNode::subMap2SubMapOptimization subMapOptmizationNode.cpp:4485 */

/* Every literal / rosparam the path reads (SURVEY.md §5 "Config / flags"), gathered in one struct. */
typedef struct lisreg_params {
    int   max_iters;            /* loop bound 15 | 20 | 30            (:606 | :1520 | :4501)                   */
    int   fixed_iters;          /* >0: benchmark mode — run exactly this many iterations, no early exit        */
    float knn_sq_thresh;        /* pointSearchSqDis[4] < tau: 1.0 | 2.0 | 2.0   (:657,776 | :1610 | :4609)     */
    float conv_deg;             /* deltaR bound 0.005 | 0.003 | 0.002 (:969 | :1963 | :4962)                   */
    float conv_cm;              /* deltaT bound 0.05  | 0.03  | 0.02                                            */
    int   min_corr;             /* laserCloudSelNum < 50 -> no-op iteration (:870)                             */
    float eig_thresh;           /* degeneracy eigenvalue threshold 100 (:932)                                  */
    int   edge_min;             /* edgeFeatureMinValidNum (config/params.yaml:119 = -1)                        */
    int   surf_min;             /* surfFeatureMinValidNum (config/params.yaml:120 = 100)                       */
    float line_ratio;           /* lambda0 > 3*lambda1 (:692)                                                  */
    float plane_tol;            /* |n.p + d| > 0.2 invalidates the plane (:798)                                */
    float accept_s;             /* keep correspondence iff s > 0.1 (:734,814)                                  */
    int   use_label_weight;     /* w = 2 - LabelSorce[label] (subMapOptmizationNode.cpp:1671,1795)             */
    float label_score[32];      /* LabelSorce table, config/label.yaml:214-234, indexed by label; 20..31 default to 0 and labels
                                 * >= 32 read as 0: std::map::operator[] on a label the yaml does not list gives w = 2.0      */
    int   emulate_matp_shadow;  /* 1: reproduce the local-matP quirk; 0: keep P from iter 0 */
    int   skip_empty_target;    /* variant #3 skips a stage whose target cloud is empty (:4505-4509)           */
    int   use_imu_blend;        /* transformUpdate slerps roll/pitch toward IMU (#1,#2) or not (#3)            */
    float imu_rpy_weight;       /* imuRPYWeight (config/params.yaml:88 = 0.1)                                  */
    float rotation_tol;         /* rotation_tollerance (config/params.yaml:124 = 1000)                         */
    float z_tol;                /* z_tollerance (config/params.yaml:123 = 1000)                                */
} lisreg_params;

/* IMU scalars the path reads from cloud_info / semantic_info (msg/cloud_info.msg:4-10). */
typedef struct lisreg_imu {
    int   imu_available;        /* cloudInfo.imuAvailable   */
    float imu_roll_init;        /* cloudInfo.imuRollInit    */
    float imu_pitch_init;       /* cloudInfo.imuPitchInit   */
} lisreg_imu;

typedef struct lisreg_stats {
    int   iters;                /* iterCount as the reference prints it (:620 | :1535 | :4517)                 */
    float deltaR, deltaT;       /* last computed (deg, cm); 100 if no iteration solved (member init :70-71)    */
    int   degenerate;           /* isDegenerate after the call (persists in the context, :67)                  */
    int   n_corr_last;          /* laserCloudSelNum of the last iteration run                                  */
    int   status;               /* LISREG_OK | LISREG_NOT_ENOUGH_FEATURES | LISREG_TOO_FEW_CORRESPONDENCES     */
} lisreg_stats;

/* Optional per-iteration trace (parity tests): LISREG_TRACE_STRIDE floats per iteration:
 *   [0] n_corr  [1..36] AtA row-major  [37..42] AtB  [43..48] X (after projection)  [49..54] T after update
 *   [55] 1 if this iteration solved (n_corr >= min_corr) else 0 */
#define LISREG_TRACE_STRIDE 56

/* One independent registration of a batch (independent frames / loop-closure candidate pairs). */
typedef struct lisreg_item {
    const void* src_corner; int n_corner;     /* source edge features   (laserCloudSharpCornerLast, :641)      */
    const void* src_surf;   int n_surf;       /* source planar features (laserCloudSharpSurfLast,  :757)      */
    int         stride_bytes;                 /* 32 for PCL structs; ignored for LISREG_FMT_DEVICE             */
    int         fmt;                          /* LISREG_FMT_*                                                  */
    int         target;                       /* target slot id from lisreg_set_target_slot (0 = default)      */
    int         degenerate_in;                /* isDegenerate carried in from the previous frame (usually 0)   */
    lisreg_imu  imu;
} lisreg_item;

typedef struct lisreg_ctx lisreg_ctx;

/* ---- lifecycle ------------------------------------------------------------------------------------------- */
int  lisreg_device_count(void);                                   /* number of visible HIP devices (0 if none)  */
int  lisreg_create(int device, lisreg_ctx** out);                 /* one context per caller thread (§8b Callers) */
void lisreg_destroy(lisreg_ctx* ctx);
const char* lisreg_last_error(const lisreg_ctx* ctx);             /* never NULL                                 */
/* Use an existing HIP stream (e.g. the caller's); NULL restores the context's own stream. */
int  lisreg_set_stream(lisreg_ctx* ctx, void* hip_stream);
void* lisreg_get_stream(const lisreg_ctx* ctx);

/* Fill `p` with the literals of one of the three reference copies. */
int  lisreg_default_params(int variant, lisreg_params* p);

/* ---- target (local map / submap) ------------------------------------------------------------------------- */
/* Replaces kdtreeCornerFromMap->setInputCloud / kdtreeSurfFromMap->setInputCloud
 * (odomEstimationNode.cpp:602-603 | subMapOptmizationNode.cpp:1516-1517 | :4496-4497): builds the device search index
 * of the two target clouds.  Host clouds are copied to the device; LISREG_FMT_DEVICE clouds are REFERENCED, not copied —
 * the caller's records must stay valid and unchanged until the slot is set again or the context is destroyed (with the
 * option "rebuild_targets_each_run" every lisreg_batch_run re-reads them).  The index persists until the next call for
 * that slot, so a target shared by many registrations is built once.  Slot 0 is what lisreg_align uses.  A cloud with an
 * infinite coordinate, or with no finite point at all, is refused (LISREG_ERR_ARG); NaN points are simply never neighbours.
 * A cloud is limited to 2^28 - 1 points (the kernels address its 16-byte records by 32-bit byte offsets). */
int  lisreg_set_target(lisreg_ctx* ctx, const void* corner, int n_corner,
                       const void* surf, int n_surf, int stride_bytes, int fmt);
int  lisreg_set_target_slot(lisreg_ctx* ctx, int slot, const void* corner, int n_corner,
                            const void* surf, int n_surf, int stride_bytes, int fmt);
/* Adapter for the submap API (subMap.h:435-777): concatenates class clouds in the reference's order —
 * corner = pole; surf = ground, building, dynamic (extractSlidingCloud subMapOptmizationNode.cpp:1408-1419;
 * extractSubMapCloud :3996-4007) — then behaves as lisreg_set_target_slot.  NULL/0 classes are skipped. */
int  lisreg_target_from_classes(lisreg_ctx* ctx, int slot,
                                const void* pole, int n_pole, const void* ground, int n_ground,
                                const void* building, int n_building, const void* dynamic, int n_dynamic,
                                int stride_bytes, int fmt);

/* ---- registration ---------------------------------------------------------------------------------------- */
/* Replaces the whole body of scan2SubMapOptimization() (odomEstimationNode.cpp:596-626 and the two copies):
 * guard, GN loop {cornerOptimization :633, surfOptimization :749, combineOptimizationCoeffs :829,
 * LMOptimization :852}, transformUpdate :976.  T: in = initial guess, out = result.  The context keeps
 * isDegenerate across calls like the reference's member (:67). */
int  lisreg_align(lisreg_ctx* ctx,
                  const void* src_corner, int n_corner, const void* src_surf, int n_surf,
                  int stride_bytes, int fmt,
                  const lisreg_params* params, const lisreg_imu* imu /* may be NULL */,
                  float T[6], lisreg_stats* stats /* may be NULL */);

/* Batch of independent registrations (same semantics per item as lisreg_align with its own isDegenerate).
 * T: n_items x 6 floats in/out; stats: n_items entries (may be NULL).  Returns 0 if the batch ran; per-item
 * outcomes are in stats[i].status. */
int  lisreg_align_batch(lisreg_ctx* ctx, int n_items, const lisreg_item* items,
                        const lisreg_params* params, float* T, lisreg_stats* stats);

/* Asynchronous form for device-resident batches (all items LISREG_FMT_DEVICE): enqueue on the context's
 * stream with poses/stats left on the device; lisreg_batch_fetch synchronises and copies them out.
 * `prepare` does all per-batch allocation and the item-table upload (one pinned staging buffer, asynchronous copies);
 * `run` only launches kernels — optional index rebuild, then `bound` x {correspondence kernel, solve kernel}, finalize:
 * eager launches, no host synchronisation, registrations that have converged make their workgroups exit at once — so
 * it can be timed with events on lisreg_get_stream().  T_init is copied at prepare time and re-applied on the device at
 * the start of every run.  (The synchronous entry points lisreg_align / lisreg_align_batch additionally read a 4-byte
 * finished-counter every few iterations and stop launching once every item is done.)
 * Life time of a target's search structures: the grid index of a target slot is built by lisreg_set_target* and stays valid until that
 * slot is set again; its k-NN graph (front-end 3) is built on first use — at lisreg_set_target when "search_mode" is 3, else at the first
 * lisreg_batch_prepare whose auto choice falls on the graph — and is then REUSED by every later prepare / run against that slot: a node
 * that registers many batches against one submap pays for index and graph once.  Only with "rebuild_targets_each_run" = 1 (bench.py: the
 * reference rebuilds both kd-trees inside every scan2SubMapOptimization, :602-603) does every lisreg_batch_run rebuild the indexes, and
 * the graphs with them, of all targets of the prepared batch. */
int  lisreg_batch_prepare(lisreg_ctx* ctx, int n_items, const lisreg_item* items,
                          const lisreg_params* params, const float* T_init);
int  lisreg_batch_run(lisreg_ctx* ctx);
int  lisreg_batch_fetch(lisreg_ctx* ctx, float* T, lisreg_stats* stats);
/* Host feeder for streams of batches (lisreg_api_feed.hip).  Packs the HOST clouds of `items` (the reference's PCL structs) to 16-byte
 * device records with "feeder_threads" host threads (option, default 8) into pinned staging and uploads them asynchronously, on a copy
 * stream of the context, into one of two device buffers the context owns; items_out[i] is items[i] with device pointers and
 * LISREG_FMT_DEVICE.  The caller's clouds are not referenced after the call returns.  The next lisreg_batch_prepare waits for the
 * uploads ON THE DEVICE.  Buffers alternate between calls, so the loop
 *     stage(k+1); fetch(k); prepare(k+1); run(k+1);          (or: prepare(k); run(k); stage(k+1); fetch(k); ...)
 * uploads batch k+1 underneath the kernels of batch k — the reference has no counterpart (its clouds never leave the host);
 * lisreg_align_batch uses it internally for batches of >= 262144 source points. */
int  lisreg_stage_host_items(lisreg_ctx* ctx, int n_items, const lisreg_item* items, lisreg_item* items_out);

/* Device pointer to the prepared batch's result block: n_items x 12 floats
 * {T[6], iters, deltaR, deltaT, degenerate, n_corr_last, status} — what a multi-GPU host all-gathers. */
void* lisreg_batch_result_device(const lisreg_ctx* ctx);

/* Options outside the reference's parameter surface: "rebuild_targets_each_run" (0/1: re-run the target index
 * build inside every lisreg_batch_run, as the reference rebuilds both kd-trees per registration, :602-603),
 * "trace_cap" (per-item trace records kept on the device for batches; 0 = off), "search_mode" (the exact 5-NN front-end
 * that stands in for pcl::KdTreeFLANN::nearestKSearch: 0 LDS-staged workgroup box, 1 per-lane grid walk,
 * 3 k-NN graph scan with the walk as its fall-back — costs 1 KB of device memory per target point for the
 * neighbour rows —, 5 cell rows: the same certified list scan, but the list belongs to the grid cell (or the octant of it) the query
 * falls into instead of to last iteration's nearest neighbour, so nothing is carried between Gauss-Newton iterations and the first
 * iterations of a batch — queries a pose error away from every surface — cost the same scan as the last ones; ~3-5 KB of device memory
 * per target point, built in ~3 ns per target point —, 4 auto [default]: 5 when the prepared batch asks at least "cell_min_ratio"
 * [default 170] query-iterations per target point and its targets' rows fit "cell_rows_max_mb" [default 16384], else 3 from
 * "graph_min_ratio" [default 60] query-iterations per target point on, else 1 — all return the same neighbours; the one exception is two candidates at exactly equal float distance from a query
 * competing for the fifth place: the first one met wins, and the front-ends meet them in different orders),
 * "sort_sources" (0 caller order, 1 column sort, 2 auto [default]: probe the order when a batch is prepared — one pass over the sources
 * and a 4-byte read-back, skipped for batches of the same shape (items, points) as the one last probed, whose verdict is reused; every 32nd
 * such batch is probed again; the verdict decides speed only, the search is exact on any order),
 * "index_build" (how "rebuild_targets_each_run" rebuilds the target grids of a batch: 0 bucket sort with one global atomic per
 * point, 1 strip form — LDS histograms, one workgroup per strip of cells; an error if a grid does not fit its LDS tables —,
 * 2 [default] strip form whenever the grids fit; both produce the same index bit for bit), "index_strip_cells" (cells per strip aimed at; 0 [default]: 1024 for a batch of one or two
 * targets, 2048 beyond — a shared submap is a handful of strips and wants them smaller), "index_strip_cap",
 * "xcd_order" (dispatch order of the correspondence workgroups of a graph-front-end batch: 0 block order, 1 by target sector so that
 * each of the 8 XCDs — each with its own L2 — works on one eighth of the target, 2 auto [default]: 1 for batches of >= 32 registrations;
 * results do not depend on it, bit for bit),
 * "exact_arithmetic" (0 [default] the production arithmetic of the correspondence kernel: FMA contraction, 1-ulp hardware reciprocal /
 * square root in the line and plane fits, fp32 sums inside a wavefront — poses within 1e-3 of the reference, a handful of
 * threshold-straddling correspondences per thousand may differ; 1 the reference's arithmetic operation for operation — IEEE division
 * and sqrt, cv::eigen's pivoted Jacobi, fp64 sums, no contraction, the pose's sines / cosines from the host's libm (one small
 * synchronous round trip per iteration: a run of this build blocks the caller and cannot be stream-captured): accept flags,
 * correspondence counts, iteration counts and — bit for bit on the 100-configuration sweep — poses EQUAL the CPU restatement's
 * (tests/test_exact.py, tests/exact_sweep.py), at roughly 0.5x the throughput),
 * "canonical_ties" (0 [default]: of two target points at EXACTLY equal float distance from a query the first one met is kept, and
 * the search front-ends meet them in different orders — about one query in 10^5-10^6 on scan data, like FLANN's own traversal order;
 * 1: such ties are noted during the search and resolved by (distance, original index), so the five neighbours, their order and every
 * bit computed from them are the same in every front-end and batch shape — a frame registers identically alone and inside any batch —
 * at +7 % per correspondence launch; implied by "exact_arithmetic"),
 * "cell_anchor_until" (graph front-end: during GN iterations 1 .. this [default 1] a query also tries an anchor out of its own grid column
 * and takes the nearer of the two — the pose moves by decimetres in the first step, last iteration's nearest neighbour is a poor start;
 * results do not depend on it),
 * "feeder_threads" (host threads of lisreg_stage_host_items, default 8; 0 = structs uploaded as they are and packed on the device),
 * "feeder_numa" (1 [default]: the packing threads are bound to those CPUs of the device's NUMA node that the process may run on — the
 * pinned staging buffers are on that node already; read-only "feeder_numa_node" / "feeder_numa_cpus" say what was found),
 * "feeder_copy_engine" (1 [default]: while the next packed chunk is not ready and the copy engine is idle, the engine takes the last free
 * chunk of a pinned cloud as it is and a kernel packs it on the device; 0: never; 2: whenever a packed chunk is not ready — for tests;
 * 3: the engine takes the FIRST chunk whatever the packing threads do — for tests),
 * "interleave" (0 [default]: one launch per Gauss-Newton iteration for the whole batch; 1 / 2: a fixed-iteration run of at least
 * "interleave_min_blocks" [8192] workgroups is cut at an item boundary near its middle and the halves iterate on two streams, so that each
 * half's 6x6 solves run underneath the other half's correspondence launch — 1: the halves' launches alternate, 2: free-running; same
 * kernels on the same data in the same order per registration, results identical to the bit; measured +1.7 % (mode 2) on
 * BASELINE configs[1], +5.8 % on configs[4], at the price of per-kernel durations that are no longer a launch's own — off),
 * "row_reach" (1 [default]: a batch that takes the cell rows and rebuilds its targets inside every run ("rebuild_targets_each_run") builds
 * rows only for the grid cells its queries come within a metre of under their INITIAL poses — lisreg_batch_prepare makes one pass over
 * the batch's source points for that —; a query that reaches a cell without rows takes the cell walk, so results do not depend on it,
 * bit for bit.  A run counts such queries; more than one query-iteration in a thousand and the prepared batch's later runs, and the next
 * 32 batches prepared on the context, build all rows.  0: all rows always),
 * "graph_min_ratio", "cell_min_ratio" (auto takes the cell rows from this many query-iterations per target point: 110), "cell_rows_max_mb",
 * "first_pass_mm", "count_searches", "early_stop_chunk". */
int  lisreg_set_option(lisreg_ctx* ctx, const char* name, int value);
/* Read back an option, or "front_end" = the search front-end the prepared batch actually runs (auto resolved), or
 * "index_build_now" = 1 if the prepared batch rebuilds its targets in strip form, "xcd_order_now" = 1 if the last run used the
 * sector dispatch order, "interleaved_now" = 1 if it ran as two halves; size of the prepared batch's search index: "index_kib_grid"
 * (sorted records + cell tables of its targets), "index_kib_front_end" (k-NN graph rows or cell rows + their tables; 0 for the cell
 * walk), "index_target_points" (points in those targets) — bench.py's roofline.index_bytes_per_target_point —, "index_kib_front_end_built" (the
 * cell rows the last run really built, read back from the device: synchronous); "row_reach_now" = 1 if the last
 * run built its cell rows for the cells the query marks reach only, "row_reach_misses" = query-iterations of the last FETCHED run that
 * found their cell without rows. */
int  lisreg_get_option(const lisreg_ctx* ctx, const char* name, int* value);

/* Diagnostics of the motion certificate (enable with option "count_searches" = 1; accumulates until re-enabled):
 * out[2*k] = queries re-searched at GN iteration k, out[2*k+1] = queries processed at iteration k (k < 32).
 * Graph front-end, n > 64: out[64+k] = (wavefronts with a walking lane) << 32 | wavefronts, of iteration k. */
int  lisreg_get_counters(lisreg_ctx* ctx, unsigned long long* out, int n);

/* Test hook (option "dump_neighbors" = 1 before the batch is prepared; search modes 1 and 3): the five neighbours the LAST
 * executed GN iteration used for every source point, as ORIGINAL indices into the target cloud of the point's kind, nearest
 * first, -1 where fewer than five lie inside sqrt(knn_sq_thresh) — i.e. what nearestKSearch(5) + the `sqDist[4] < tau` test
 * of :657 / :776 select; row 5 = 1 where the point passed every accept test and contributed a row to the normal equations
 * (laserCloudOriFlag, :741 / :820).  out: host int[6][n_elems], n_elems = all source points of the batch in item order (corner
 * then surf of each item). */
int  lisreg_get_neighbors(lisreg_ctx* ctx, int* out, int n_elems);

/* Test hook: the DEVICE functions of the two residual models, one case per thread, on caller-given neighbourhoods — what the
 * correspondence kernel computes for a query once its five neighbours are known.  kind 1 = surfOptimization's body
 * (odomEstimationNode.cpp:776-821: plane through the five neighbours, |n.p + d| <= plane_tol test, point-to-plane residual, robust
 * weight, accept test; in the production arithmetic the plane comes from the closed form `plane5_closed` with its hand-over to the
 * column-pivoted QR), kind 0 = cornerOptimization's (:664-739).  neighbours: host float[n][5][3] (nearest first), queries: host
 * float[n][3] (the TRANSFORMED query point), label weight 1.  exact != 0 runs the exact-arithmetic build of the same source.
 * out: host float[n][10]: kind 1: pa, pb, pc, pd (NaN = plane rejected), 1 if the closed form produced it / 0 if the QR did,
 * coeff x, y, z, intensity (:808-811), 1 if accepted (s > accept_s, :813); kind 0: centroid x, y, z and two components of the line
 * direction (NaN = lambda-ratio test failed), coeff x, y, z, intensity (:729-732), accepted (:734). */
int  lisreg_test_fit_models(lisreg_ctx* ctx, int kind, int n, const float* neighbours, const float* queries,
                            const lisreg_params* params, int exact, float* out);

/* Diagnostics (tests): the search index of target `slot`, kind 0 = corner / 1 = surf, as it stands in HBM after the work queued on
 * the context's stream: dims = {n, nx, ny, nz, n_cells}, geom = {ox, oy, oz, cell edge}, the cell-sorted records
 * (x, y, z, bit-cast original index; ordered by cell (ix * ny + iy) * nz + iz, then original index) and cell_start[n_cells + 1].
 * Any output pointer may be NULL. */
int  lisreg_get_target_index(lisreg_ctx* ctx, int slot, int kind, int* dims, float* geom,
                             float* sorted_out, int sorted_capacity, int* cell_start_out, int cell_capacity);

/* Diagnostics: the k-NN graph of a target's search index (search front-end 3; built now if it was not yet).  *k = entries per
 * row; rows_out[p * k + j] = (x, y, z, sorted position as int bits) of the j-th nearest other point of sorted point p, ascending
 * by distance (to 2^-16 relative: the build sorts quantised keys, the scan's stop test carries a millimetre of slack for it), padded with (p's own coordinates, -1); meta_out[p] = (rho^2, count as int bits): every point closer to p than rho
 * is in its row.  Either output may be NULL.  capacity_points >= the target's point count. */
int  lisreg_get_target_graph(lisreg_ctx* ctx, int slot, int kind, int* k, float* rows_out, float* meta_out, int capacity_points);

/* Diagnostics: the cell rows of a target's search index (search front-end 5; built now if they were not yet).  *n_rows = rows in all,
 * *k = entries per row.  table_out[cell] (the index's n_cells entries, cell = (ix * ny + iy) * nz + iz): -2 = no target point within two
 * cells of this one, -1 = no row, else (first row << 8) | octant mask — the cell's rows are [the row at its centre, then one row per
 * octant of the mask in ascending octant order, octant = (x upper half) + 2 (y upper half) + 4 (z upper half)]; rows_out[r * k + j] =
 * (x, y, z, sorted position as int bits) of the j-th nearest target point of the row's centre, ascending (to 2^-16 relative), padded with
 * (the centre's coordinates, -1); meta_out[r] = (rho^2, count as int bits): every point closer to the centre than rho is in the row.
 * Outputs may be NULL; capacity_rows >= *n_rows when rows_out / meta_out are given (call once with NULLs to learn it).  Building the rows
 * re-makes the target's grid two cells wider than the cloud on every side (once per target): read lisreg_get_target_index after this.
 * When that happens a batch prepared on this context (lisreg_batch_prepare) is no longer valid — its copy of the grid geometry is stale —
 * and lisreg_batch_run refuses it (LISREG_ERR_ARG, "no prepared batch") until it is prepared again. */
int  lisreg_get_target_cell_rows(lisreg_ctx* ctx, int slot, int kind, int* n_rows, int* k, int* table_out, int capacity_cells,
                                 float* rows_out, float* meta_out, int capacity_rows);

/* Trace of the LAST lisreg_align call: copies min(n_iters_run, max_iters) records; returns the count. */
int  lisreg_get_trace(lisreg_ctx* ctx, float* buf, int max_iters);

/* Per-kernel timing of the last align/batch_run when enabled (HIP events on the context's stream):
 * out[0] = total ms in the correspondence+normal-equation kernel, out[1] = its launch count,
 * out[2] = total ms in the solve/update kernel, out[3] = its launch count, out[4] = index build ms. */
int  lisreg_set_profiling(lisreg_ctx* ctx, int enable);
int  lisreg_get_timing(lisreg_ctx* ctx, double out[5]);

/* ---- the step before the registration --------------------------------------------------- */
/* Replaces pcl::VoxelGrid<PointT>::filter with the reference's default settings — downSizeFilterCorner/Surf on the
 * incoming feature clouds (odomEstimationNode.cpp:272-277) and on the assembled local map (:196-201), and
 * voxel_downsample_pcl on submap class clouds (src/include/subMap.h:1207-1249): one centroid per occupied voxel of
 * edge `leaf` (xyz and intensity averaged, label = most frequent), output in ascending PCL voxel index.
 * in/out: LISREG_FMT_XYZI / _XYZIL host clouds (out has the input's stride), or LISREG_FMT_DEVICE (in and out are device
 * lisreg_dpoint arrays; payload label majority-voted) so the result can feed lisreg_align / lisreg_set_target directly.
 * *n_out receives the voxel count; if it exceeds out_capacity nothing is written and LISREG_ERR_ARG is returned.
 * Returns LISREG_LEAF_TOO_SMALL (and copies the input, as PCL does) when the voxel index would overflow int32. */
int  lisreg_voxel_downsample(lisreg_ctx* ctx, const void* in, int n, int stride_bytes, int fmt, float leaf,
                             void* out, int out_capacity, int* n_out);
/* The same for K clouds of device records (LISREG_FMT_DEVICE: label vote, or LISREG_FMT_DEVICE_XYZI: intensity average) in ONE launch
 * sequence with three host round trips in all (the K bounding boxes, the K counts, completion) — the five class grids of a key frame or of the local
 * map are bound by their ~12 launches and 3 round trips apiece, not by bandwidth.  Results equal K single calls bit for bit; clouds
 * that do not fit the joint sort (a leaf too small for its cloud, > 8 clouds) are done one by one. */
int  lisreg_voxel_downsample_multi(lisreg_ctx* ctx, int k, const void* const* in, const int* n, const float* leaf, int fmt,
                                   void* const* out, const int* out_capacity, int* n_out);
/* Replaces transformPointCloud(cloud, &pose6D) (src/core/common.cpp:112-173 and the PointXYZIL overload; callers
 * odomEstimationNode.cpp:457-458, 574-575): p' = R(T) p + t with T = {roll,pitch,yaw,x,y,z}; other fields copied.
 * Same formats as above; in == out is allowed. */
int  lisreg_transform_cloud(lisreg_ctx* ctx, const void* in, int n, int stride_bytes, int fmt, const float T[6], void* out);

/* ---- producer of cloud_info: range-image projection + LOAM feature extraction ----------------- */
/* Host point layout of the raw scan: PointXYZIRT (src/include/common.h:12-23): x@0 y@4 z@8 intensity@16 uint16 ring@20
 * float time@24, 32-byte stride.  Device layout: lisreg_dpoint with the ring in the low 16 bits of the payload. */
#define LISREG_FMT_XYZIRT  3
typedef struct lisreg_feature_params {
    int   n_scan;             /* N_SCAN        (config/params.yaml:68 = 64)   */
    int   horizon_scan;       /* Horizon_SCAN  (config/params.yaml:69 = 1800) */
    int   downsample_rate;    /* downsampleRate(config/params.yaml:72 = 2): rows with ring % rate != 0 are dropped */
    float min_range;          /* lidarMinRange (config/params.yaml:73 = 0.0)  */
    float max_range;          /* lidarMaxRange (config/params.yaml:74 = 70.0) */
    float edge_threshold;     /* edgeThreshold (config/params.yaml:117 = 1.0) */
    float surf_threshold;     /* surfThreshold (config/params.yaml:118 = 0.1) */
} lisreg_feature_params;
/* The five clouds of cloud_info (msg/cloud_info.msg:20-25).  Buffers are caller-allocated in the INPUT's layout with the
 * stated capacities (points); counts are written back.  n_scan*horizon_scan is always enough for each. */
typedef struct lisreg_feature_out {
    void* deskewed;      int cap_deskewed,      n_deskewed;       /* extractedCloud    -> cloud_deskewed        */
    void* corner;        int cap_corner,        n_corner;         /* cornerCloud       -> cloud_corner          */
    void* surface;       int cap_surface,       n_surface;        /* surfaceCloud      -> cloud_surface         */
    void* corner_sharp;  int cap_corner_sharp,  n_corner_sharp;   /* sharpCornerCloud  -> cloud_corner_sharp    */
    void* surface_sharp; int cap_surface_sharp, n_surface_sharp;  /* SharpSurfaceCloud -> cloud_surface_sharp   */
} lisreg_feature_out;
/* Replaces LaserProcessing::projectPointCloud, cloudExtraction, calculateSmoothness, markOccludedPoints and
 * extractFeatures (src/core/laserProcessing.cpp:467-510, 515-539, 544-563, 568-605, 610-713) for one scan, without the
 * IMU de-skew (deskewPoint returns the point unchanged when no IMU data is available, :429; lisreg_extract_features_deskew
 * below applies it).  Output order is the
 * reference's: rings ascending, six sectors per ring, corners in pick order (largest curvature first). */
int  lisreg_extract_features(lisreg_ctx* ctx, const void* cloud, int n, int stride_bytes, int fmt,
                             const lisreg_feature_params* params, lisreg_feature_out* out);
int  lisreg_default_feature_params(lisreg_feature_params* p);
/* The IMU de-skew inside projectPointCloud (deskewPoint / findRotation, laserProcessing.cpp:368-399, 427-462, called at :501):
 * every point that wins a range-image pixel is rotated into the frame of the first such point, with the rotation integrated from
 * the IMU (imuDeskewInfo, :222-266 — a ~50-term prefix sum that stays on the host) interpolated at the point's time stamp.
 * findPosition returns zeros in the reference (:405-421), so there is no positional part.  Ranges, columns and the feature
 * selection use the raw points, hence only the coordinates of the five output clouds change. */
typedef struct lisreg_deskew {
    int    enabled;               /* deskewFlag == 1 && cloudInfo.imuAvailable (:429) */
    int    imu_pointer_cur;       /* last valid index of the tables (imuPointerCur after :261) */
    const double* imu_time;       /* host arrays [imu_pointer_cur + 1] */
    const double* imu_rot_x;
    const double* imu_rot_y;
    const double* imu_rot_z;
    double time_scan_cur;         /* timeScanCur; a point's time = time_scan_cur + its `time` field */
    const float* time_device;     /* LISREG_FMT_DEVICE only: per-point `time` (device array [n]); host structs carry it at offset 24 */
} lisreg_deskew;
/* lisreg_extract_features with the de-skew applied to the output coordinates; deskew == NULL or enabled == 0 is the plain call */
int  lisreg_extract_features_deskew(lisreg_ctx* ctx, const void* cloud, int n, int stride_bytes, int fmt,
                                    const lisreg_feature_params* params, const lisreg_deskew* deskew, lisreg_feature_out* out);

/* lisreg_extract_features for n_sweeps sweeps at once — the throughput form: the sweeps are stacked into one range image (the
 * selection kernel's grid is sweeps x rings) and every pass of the pipeline runs once.  sweeps[s] / n[s]: DEVICE lisreg_dpoint
 * records (ring in the low 16 bits of the payload); outs[s]: device buffers as in lisreg_extract_features.  No IMU de-skew.
 * The five clouds of every sweep are identical to what a single call on that sweep returns.  At most 256 sweeps and
 * n_sweeps x n_scan <= 32768 per call. */
int  lisreg_extract_features_batch(lisreg_ctx* ctx, int n_sweeps, const void* const* sweeps, const int* n,
                                   const lisreg_feature_params* params, lisreg_feature_out* outs);

/* The "semantic mask": SemanticFusionNode::categoryMapping (src/node/semanticFusionNode.cpp:173-189) splits the labelled
 * cloud, preserving order, by UsingLableMap[label] (config/label.yaml:177-196): 10 -> dynamic, 40 -> ground, 50 -> building,
 * 81 -> pole, anything else (label 0 has no entry) -> outlier.  These five clouds are what semantic_info carries and what
 * selects the edge (pole) and planar (ground + building + dynamic) feature sets of copies #2/#3.
 * using_label[32]: the map indexed by label & 31, or NULL for the reference's label.yaml.  Clouds: LISREG_FMT_XYZIL host
 * structs or LISREG_FMT_DEVICE records (label in the payload); outputs in the input's layout. */
typedef struct lisreg_semantic_out {
    void* cloud[5];          /* dynamic, ground, building, pole, outlier — the order of the reference's if-chain */
    int   cap[5];
    int   n[5];
} lisreg_semantic_out;
int  lisreg_semantic_split(lisreg_ctx* ctx, const void* cloud, int n, int stride_bytes, int fmt,
                           const uint32_t* using_label /* [32] or NULL */, lisreg_semantic_out* out);

/* `*a += *b` for device records: the K (<= 8) clouds end to end into `out` (capacity >= the sum), one launch on the context's stream;
 * the call does not wait, every later call of this context is ordered behind it.  *n_out = the total (may be NULL). */
int  lisreg_concat_device(lisreg_ctx* ctx, int k, const void* const* in, const int* n, void* out, int* n_out);

/* One host cloud (the reference's PCL structs: LISREG_FMT_XYZI, or LISREG_FMT_XYZIL whose uint16 at byte 20 — label, or the ring of a
 * raw sweep — becomes the payload) into a caller-owned device buffer of n 16-byte lisreg_dpoint records, through a pinned staging buffer
 * packed by the feeder threads; returns when the records are in HBM.  What a node's callback does first with a sensor_msgs cloud
 * (pcl::fromROSMsg in laserCloudInfoHandler, odomEstimationNode.cpp:164-175), for the *_DEVICE forms of the entry points. */
int  lisreg_upload_cloud(lisreg_ctx* ctx, const void* cloud, int n, int stride_bytes, int fmt, void* dev_out);

/* ---- §8 f-3: local-map maintenance (src/include/subMap.h) -------------------------------------------------- */
/* A k = 1 search index over one cloud, kept in HBM under `slot` (its own slot space, separate from the registration
 * targets).  Replaces pcl::search::KdTree<PointT>::setInputCloud (subMap.h:889 `local_map->tree_dynamic`,
 * subMapOptmizationNode.cpp:2779 icp.setInputTarget).  LISREG_FMT_DEVICE clouds are referenced, not copied. */
int  lisreg_map_index_set(lisreg_ctx* ctx, int slot, const void* cloud, int n, int stride_bytes, int fmt);
/* The same for n clouds at once — setInputTarget of every candidate of a loop-closure batch (subMapOptmizationNode.cpp:2793 inside the
 * candidate loop of :2776-2840): one bounding-box launch, one read-back and ONE sort launch sequence for all of them.  slots[k] receives
 * clouds[k] (counts[k] points; one stride / format for the batch; a slot may not be named twice).  Each index equals what
 * lisreg_map_index_set builds for that cloud, bit for bit.  Two or more clouds whose grids fit it take the strip form of the build
 * (option "index_build": 0 keeps the batched bucket sort; 256 clouds of 200 k points: 1.2 instead of 4.9 ms). */
int  lisreg_map_index_set_batch(lisreg_ctx* ctx, int n_maps, const int* slots, const void* const* clouds, const int* counts,
                                int stride_bytes, int fmt);
/* nearestKSearch(query, k = 1) for a whole cloud: idx_out[i] = index into the map cloud of the nearest point, or -1 when it
 * is farther than max_dist (pass a huge value for the reference's unbounded search); sqd_out[i] = squared distance in float,
 * accumulated x, y, z like FLANN's L2_Simple.  Equidistant candidates resolve to the smallest index.  idx_out / sqd_out are
 * host arrays, or device arrays when fmt is LISREG_FMT_DEVICE. */
int  lisreg_nearest(lisreg_ctx* ctx, int slot, const void* query, int n, int stride_bytes, int fmt, float max_dist,
                    int* idx_out, float* sqd_out);
/* SubMapManager::map_scan_feature_pts_distance_removal (subMap.h:1064-1100): drop the points of `cloud` within
 * center_radius (x, y) of the sensor whose nearest point of the indexed map lies at distance d with d <= near_dist_thre or
 * dist_thre_min <= d <= dist_thre_max; order preserved.  `out` has room for n points of the input layout.  Returns
 * LISREG_NOT_ENOUGH_FEATURES (and out = in) where the reference returns false (n <= 10). */
int  lisreg_dynamic_filter(lisreg_ctx* ctx, int slot, const void* cloud, int n, int stride_bytes, int fmt, float center_radius,
                           float dist_thre_min, float dist_thre_max, float near_dist_thre, void* out, int* n_out);
/* SubMapManager::bbx_filter (subMap.h:1124-1152): bounds = {min_x, min_y, min_z, max_x, max_y, max_z} (bounds_t); keeps the
 * points strictly inside, or the others when delete_box is set; order preserved. */
int  lisreg_bbx_filter(lisreg_ctx* ctx, const void* cloud, int n, int stride_bytes, int fmt, const double bounds[6],
                       int delete_box, void* out, int* n_out);
/* SubMapManager::get_cloud_bbx (subMap.h:131-163); an empty cloud yields {DBL_MAX x3, -DBL_MAX x3}. */
int  lisreg_cloud_bounds(lisreg_ctx* ctx, const void* cloud, int n, int stride_bytes, int fmt, double bounds[6]);

/* The sliding local map as ONE device-resident object (localMap_t, subMap.h:679-777): five class clouds in the map frame kept in
 * HBM as 16-byte records, class order of append_feature / merge_feature_points: 0 dynamic, 1 pole, 2 ground, 3 building,
 * 4 outlier.  lisreg_localmap_insert = SubMapManager::insert_local_map (subMap.h:979-1059) as makeSubMapThread calls it
 * (subMapOptmizationNode.cpp:640-646, 708-714); lisreg_localmap_extract = extractSlidingCloud (:1369-1432) followed by the two
 * kdtree setInputCloud calls of scan2SubMapOptimization (:1517-1518): it leaves the registration target in `target_slot`
 * (pass -1 to skip that), so a frame loop is  extract -> lisreg_align / batch -> insert  with no cloud crossing PCIe
 * except the incoming frame. */
typedef struct lisreg_localmap_params {
    int   max_num_pts;                    /* 80000: dynamic removal starts once feature_point_num > max_num_pts / 5 (:605, subMap.h:1007) */
    int   dynamic_removal_on;             /* map_based_dynamic_removal_on = true (:608)                                         */
    float dynamic_removal_center_radius;  /* 30.0 (:609) — compared with x^2 + y^2 of the MAP-frame point, as the reference does */
    float dynamic_dist_thre_min;          /* 0.3  (:610)                                                                        */
    float dynamic_dist_thre_max;          /* 3.0  (:611); widened to at least min + 0.1 like subMap.h:1006                      */
    float near_dist_thre;                 /* 0.03 (:612)                                                                        */
    float leaf[5];                        /* in-place voxel grids of extractSlidingCloud: 0.1, 0.05, 0.4, 0.2, 0.6 (:1385-1389)  */
    float crop_box[6];                    /* cur_bbx = {-70,-70,-10, 70,70,20} (:1377-1379), moved by the current pose           */
    float crop_pad;                       /* 2.0: bbx_boundary_pad of get_intersection_bbx (:1384)                               */
} lisreg_localmap_params;
typedef struct lisreg_localmap_info {
    int    n[5];                          /* points per class now                                   */
    int    feature_point_num;             /* as of the last insert (subMap.h:1039-1043)             */
    double bound[6];                      /* localMap->bound as of the last insert (:1047-1049)     */
    double crop[6];                       /* bbx_intersection of the last extract                   */
    int    n_target_corner, n_target_surf;/* laserCloud{Corner,Surf}FromSubMap sizes of the last extract */
} lisreg_localmap_info;
int  lisreg_localmap_default_params(lisreg_localmap_params* p);
int  lisreg_localmap_reset(lisreg_ctx* ctx, int map_id);
/* clouds[5] / n[5]: the key frame's UN-downsampled class clouds in the sensor frame (semantic_dynamic, _pole, _ground, _building,
 * _outlier); host PointXYZIL structs or LISREG_FMT_DEVICE records.  pose = the frame's optimized_pose {roll,pitch,yaw,x,y,z}.
 * The outlier cloud is accepted and ignored, exactly as the reference leaves its transform commented out (subMap.h:1003). */
int  lisreg_localmap_insert(lisreg_ctx* ctx, int map_id, const void* const clouds[5], const int n[5], int stride_bytes, int fmt,
                            const float pose[6], const lisreg_localmap_params* params, lisreg_localmap_info* info);
int  lisreg_localmap_extract(lisreg_ctx* ctx, int map_id, const float cur_pose[6], const lisreg_localmap_params* params,
                             int target_slot, lisreg_localmap_info* info);
/* Copy one cloud out as 16-byte records (host or device destination): cls 0-4 = the class clouds, 5 / 6 = the corner / surf
 * target of the last extract.  *n_out is always set; LISREG_ERR_ARG if capacity is too small. */
int  lisreg_localmap_get(lisreg_ctx* ctx, int map_id, int cls, void* out, int capacity, int* n_out);
/* Copy #3's side of the same row: submap_t + SubMapManager::insert_submap (src/include/subMap.h:835-978) and
 * SubMapOptmizationNode::extractSubMapCloud (src/node/subMapOptmizationNode.cpp:3976-4081), device-resident.  Submaps share the id space and
 * the storage of the local maps (lisreg_localmap_reset / _get work on them: classes 0-4, 5 / 6 = the clouds of the last extract).
 *   _insert  = insert_submap: the key frame's DOWN-sampled class clouds (semantic_*_down, all FIVE — the outlier class is appended here,
 *              :882) are transformPointCloud'ed by the frame's relative_pose into the submap's own frame, the dynamic class goes through
 *              the map-based removal once feature_point_num > max_num_pts / 5 (:887-892), append_feature, feature_point_num, local_bound
 *              over all classes (:961-963) and bound = transform_bbx(local_bound, local_cp, submap_pose_6D_optimized) (:968-969);
 *              relative_pose = NULL is fisrt_submap (:785-830): the first key frame of a submap, appended as it is;
 *   _extract = extractSubMapCloud for (previous submap, current submap): bound boxes under the two poses, their intersection padded by
 *              `pad` (10 m, :3995); TARGET = previous submap's pole | ground + building + dynamic transformed into the map frame by pre_pose
 *              and cropped (:3997-4020), installed as registration target `target_slot` (-1: assembled only); SOURCES = current submap's
 *              pole | dynamic + ground + building in the submap's own frame, cropped by the intersection box moved into that frame
 *              (tran_map.inverse(), :4056-4061) and voxel-downsampled (0.2 / 0.5, :4066-4067): device records owned by the context, valid
 *              until the next call on that submap — hand them to lisreg_align (LISREG_FMT_DEVICE, LISREG_VARIANT_SUBMAP) with cur_pose as
 *              the guess = subMap2SubMapOptimization (:4485-4540);
 *   _crop_boxes = the host arithmetic of the two boxes alone (double boxes, float matrices, Eigen's cofactor Affine3f::inverse). */
typedef struct lisreg_submap_info {
    int    n[5];                          /* points per class now (dynamic, pole, ground, building, outlier)       */
    int    feature_point_num;
    double local_bound[6];                /* local_bound: {min x, y, z, max x, y, z} in the submap's own frame       */
    double bound[6];                      /* bound: local_bound moved by submap_pose (transform_bbx)                 */
} lisreg_submap_info;
typedef struct lisreg_submap_extract_out {
    double isect[6];                      /* bbx_intersection in the map frame                                       */
    double isect_local[6];                /* the same box moved into the current submap's frame                      */
    int    n_target_corner, n_target_surf;/* laserCloud{Corner,Surf}FromSubMap                                       */
    const void* src_corner; int n_src_corner;     /* laserCloudCornerLastDS (device records)                         */
    const void* src_surf;   int n_src_surf;       /* laserCloudSurfLastDS                                            */
} lisreg_submap_extract_out;
int  lisreg_submap_insert(lisreg_ctx* ctx, int map_id, const void* const clouds[5], const int n[5], int stride_bytes, int fmt,
                          const float relative_pose[6], const float submap_pose[6], const lisreg_localmap_params* params,
                          lisreg_submap_info* info);
int  lisreg_submap_extract(lisreg_ctx* ctx, int pre_id, int cur_id, const float pre_pose[6], const float cur_pose[6], float pad,
                           float corner_leaf, float surf_leaf, int target_slot, lisreg_submap_extract_out* out);
void lisreg_submap_crop_boxes(const double pre_local_bound[6], const float pre_pose[6], const double cur_local_bound[6],
                              const float cur_pose[6], float pad, double isect[6], double isect_local[6]);

/* The odometry node's target (odomEstimationNode.cpp, USING_MULTI_FRAME_TARGET), device-resident:
 *   _push   = saveKeyFrames (:421-468): the frame's FULL corner / surf feature clouds (host PCL structs or LISREG_FMT_DEVICE records,
 *             sensor frame) are transformPointCloud'ed by `pose` into the map frame and kept; the oldest frames are dropped until at
 *             most max_keep remain (the reference keeps fewer than 20: max_keep = 19);
 *   _target = laserCloudInfoHandler (:185-207) + the two kd-tree builds (:602-603): the kept frames concatenated NEWEST FIRST, voxel
 *             grids with the two leaf sizes (mappingCornerLeafSize 0.2, mappingSurfLeafSize 0.4), installed as target `target_slot`
 *             (-1: assembled only).  The target clouds live in the ring until the next _target / _reset of that ring.  The reference
 *             rebuilds clouds and trees for every sweep although they only change with a key frame; here a call finds its work done —
 *             and returns at once — while no frame was pushed since the last call, the leaf sizes are the same and `target_slot` still
 *             holds what that call installed (any lisreg_set_target* on the slot in between makes the next call rebuild). */
typedef struct lisreg_keyframes_info {
    int n_keyframes;                      /* frames kept now                                                 */
    int n_target_corner, n_target_surf;   /* laserCloud{Corner,Surf}FromMapDS sizes of the last _target call */
} lisreg_keyframes_info;
int  lisreg_keyframes_reset(lisreg_ctx* ctx, int ring_id);
int  lisreg_keyframes_push(lisreg_ctx* ctx, int ring_id, const void* corner, int n_corner, const void* surf, int n_surf, int stride_bytes,
                           int fmt, const float pose[6], int max_keep, lisreg_keyframes_info* info);
int  lisreg_keyframes_target(lisreg_ctx* ctx, int ring_id, float corner_leaf, float surf_leaf, int target_slot, lisreg_keyframes_info* info);
/* updateInitialGuess with neither IMU nor odometry (odomEstimationNode.cpp:351-392; subMapOptmizationNode.cpp:984-1020): the
 * constant-velocity guess T_cur * (T_last^-1 * T_cur). */
void lisreg_predict_pose(const float T_last[6], const float T_cur[6], float T_guess[6]);
/* updateInitialGuess as a whole (odomEstimationNode.cpp:297-419 = variant 0; subMapOptmizationNode.cpp:896-1032 = variant 1): the first
 * call takes the IMU attitude (yaw 0 unless useImuHeadingInitialization); with odometry (cloudInfo.odomAvailable, the IMU pre-integration
 * guess initialGuess*) the increment between consecutive guesses is applied to the pose; with the IMU alone the increment of its attitude;
 * with neither the constant-velocity guess above.  The copies differ where the reference's do: #1 tests `odomAvailable == false` alone for
 * the constant-velocity branch and lets the FIRST odometry message fall through to the IMU increment; #2 saves the IMU attitude in the
 * odometry branch only when there is an IMU and takes the constant-velocity branch only with neither input.  The function-local statics
 * of the reference live in lisreg_guess_state (zero-initialise; one per node).  T = transformTobeMapped / transformTobeSubMapped
 * {roll,pitch,yaw,x,y,z}, updated in place; T_prediction (may be NULL) receives transPredictionMapped where the reference sets it.
 * Host arithmetic, Eigen's Affine3f operations step by step in float. */
typedef struct lisreg_guess_input {
    int   odom_available, imu_available;                                  /* cloudInfo.odomAvailable, .imuAvailable           */
    float imu_roll_init, imu_pitch_init, imu_yaw_init;                    /* cloudInfo.imuRollInit, ...                       */
    float initial_guess_x, initial_guess_y, initial_guess_z;              /* cloudInfo.initialGuessX, ...                     */
    float initial_guess_roll, initial_guess_pitch, initial_guess_yaw;
} lisreg_guess_input;
typedef struct lisreg_guess_state {
    int   first_trans_available;                 /* firstTransAvailable                                  */
    int   last_imu_pre_trans_available;          /* lastImuPreTransAvailable                             */
    int   first;                                 /* `first` of the constant-velocity branch              */
    int   reserved;
    float last_imu_transformation[12];           /* lastImuTransformation, row-major 3x4                 */
    float last_imu_pre_transformation[12];       /* lastImuPreTransformation                             */
    float last_transform_tobe_mapped[6];         /* lastTransformTobeMapped / lastTransformTobeSubMapped */
} lisreg_guess_state;
void lisreg_guess_state_init(lisreg_guess_state* st);
void lisreg_update_initial_guess(int variant, int use_imu_heading_initialization, const lisreg_guess_input* in, lisreg_guess_state* st,
                                 float T[6], float T_prediction[6]);

/* ---- §8 f-4: pcl::IterativeClosestPoint as the loop-closure / relocalisation code drives it ------------------- */
/* Call sites: src/node/subMapOptmizationNode.cpp:2763-2833 (loop closure: 10 m, 30 iterations, 1e-4, 1e-4),
 * :1444-1465 and :4400-4420 (0.2 m, 50 iterations, 1e-5, 1e-5); RANSAC iterations 0 everywhere, no rejectors.
 * Point-to-point ICP: per iteration k = 1 correspondences within max_corr_dist -> closed-form rigid transform
 * (TransformationEstimationSVD = Umeyama without scale) -> DefaultConvergenceCriteria. */
enum {                       /* pcl::registration::DefaultConvergenceCriteria::ConvergenceState */
    LISREG_ICP_NOT_CONVERGED      = 0,
    LISREG_ICP_ITERATIONS         = 1,
    LISREG_ICP_TRANSFORM          = 2,
    LISREG_ICP_ABS_MSE            = 3,
    LISREG_ICP_REL_MSE            = 4,
    LISREG_ICP_NO_CORRESPONDENCES = 5
};
typedef struct lisreg_icp_params {
    double max_corr_dist;               /* setMaxCorrespondenceDistance (compared squared, in double, like PCL) */
    int    max_iters;                   /* setMaximumIterations */
    int    reserved;
    double transformation_epsilon;      /* setTransformationEpsilon: squared-translation bound, and 1 - eps bounds cos(angle) */
    double euclidean_fitness_epsilon;   /* setEuclideanFitnessEpsilon -> relative MSE threshold */
    double prev_mse;                    /* correspondences_prev_mse_: DBL_MAX on a fresh object; the reference's `static`
                                           ICP objects carry it from one align() to the next (feed lisreg_icp_result.prev_mse back) */
} lisreg_icp_params;
typedef struct lisreg_icp_result {
    float  final_transform[16];         /* getFinalTransformation(), row-major 4x4 */
    int    converged;                   /* hasConverged() */
    int    iters;                       /* nr_iterations_ */
    int    state;                       /* LISREG_ICP_* */
    int    n_corr_last;                 /* correspondences of the last iteration */
    double fitness;                     /* getFitnessScore(): mean squared k = 1 distance of the aligned source, unbounded */
    double prev_mse;
} lisreg_icp_result;
/* {10, 30, 1e-4, 1e-4} for kind 0 (loop closure), {0.2, 50, 1e-5, 1e-5} for kind 1; prev_mse = DBL_MAX */
int  lisreg_icp_default_params(int kind, lisreg_icp_params* p);
/* align(): target = the map index in `slot` (setInputTarget), source = `source` (setInputSource), guess = row-major 4x4 or
 * NULL for identity.  aligned_out: NULL, or room for n points of the input layout (the `output` cloud of align()). */
int  lisreg_icp_align(lisreg_ctx* ctx, int slot, const void* source, int n, int stride_bytes, int fmt,
                      const lisreg_icp_params* params, const float* guess, lisreg_icp_result* result, void* aligned_out);

/* The candidate loop of loop-closure verification as one call (detectLoopClosureForSubMap, subMapOptmizationNode.cpp:2776-2840; BASELINE
 * configs[3]: per candidate setInputTarget(:2793) / setInputSource(:2823) / align(:2831) / getFitnessScore(:2835) / hasConverged(:2836) /
 * getFinalTransformation(:2841)).  Item k aligns ITS source against the map index of ITS slot from ITS guess; every ICP iteration is one
 * launch sequence over all items, an item that has converged drops out of the launches that follow, results[k] is what
 * lisreg_icp_align returns for item k alone — to the bit (the sums are formed by one tree whatever the batch size).
 * chain_prev_mse: 0 = every item starts from params->prev_mse (fresh ICP objects); 1 = the reference's `static` object (:2763), whose
 * DefaultConvergenceCriteria carries correspondences_prev_mse_ from one align() to the next: item 0 starts from params->prev_mse, item
 * k from results[k - 1].prev_mse, in array order (results as if the items had been aligned one after the other).
 * Sources: host PCL structs or LISREG_FMT_DEVICE records (one stride / format for the batch); n = 0 items are allowed. */
typedef struct lisreg_icp_item {
    const void*  source;                /* setInputSource */
    int          n;
    int          slot;                  /* setInputTarget: lisreg_map_index_set(slot) */
    const float* guess;                 /* row-major 4x4, NULL = identity */
} lisreg_icp_item;
int  lisreg_icp_align_batch(lisreg_ctx* ctx, const lisreg_icp_item* items, int n_items, int stride_bytes, int fmt,
                            const lisreg_icp_params* params, int chain_prev_mse, lisreg_icp_result* results);

/* OptimizedICPGN (src/core/registration.cpp:19-115, src/include/registration.h:44-70): Gauss-Newton point-to-point ICP — per
 * iteration k = 1 correspondences, J = [I, -R hat(p)], H += J^T J, B -= J^T e, delta = H^-1 B, t += delta[0:3],
 * R = R exp(delta[3:6]) (Sophus SO3).  The class is never instantiated in the reference (commented-out call sites only); it is
 * built because SURVEY.md §8 f-4 lists it.  Quirk kept: the squared k = 1 distance is compared with the un-squared
 * max_correspond_distance (:50).  No convergence test (HasConverged() returns true). */
typedef struct lisreg_icpgn_result {
    float final_transform[16];          /* result_pose, row-major 4x4 */
    int   steps_applied;                /* iterations whose Hessian had a non-zero determinant */
    int   n_corr_last;
    float fitness;                      /* GetFitnessScore() */
    int   reserved;
} lisreg_icpgn_result;
/* SetTargetCloud = lisreg_map_index_set(slot); Match(source, predict_pose, transformed_out, result) + GetFitnessScore() */
int  lisreg_icp_gn_match(lisreg_ctx* ctx, int slot, const void* source, int n, int stride_bytes, int fmt,
                         unsigned max_iterations, float max_correspond_distance, const float predict_pose[16],
                         lisreg_icpgn_result* result, void* transformed_out);

/* ---- helpers that mirror src/core/common.cpp ------------------------------------------------------------- */
/* trans2Affine3f (common.cpp:54-57): row-major 3x4 [R|t]. */
void lisreg_pose_to_matrix(const float T[6], float M[12]);
/* transformUpdate (odomEstimationNode.cpp:976-1006): IMU roll/pitch blend + constraintTransformation clamps. */
void lisreg_transform_update(const lisreg_params* p, const lisreg_imu* imu, float T[6]);

/* ---- multi-GPU pose gather (one process per GPU) ------------------------------------------ */
/* RCCL bootstrap without MPI: rank 0 calls lisreg_comm_unique_id, ships the 128 bytes to the other ranks by
 * any means (file, env, torch.distributed store), then every rank calls lisreg_comm_init.  lisreg_gather_results
 * all-gathers each rank's n_local x 12-float result block (device memory) into `out_device`
 * (nranks x n_local x 12 floats) on the context's stream. */
int  lisreg_comm_unique_id(unsigned char id[128]);
int  lisreg_comm_init(lisreg_ctx* ctx, int rank, int nranks, const unsigned char id[128]);
int  lisreg_gather_results(lisreg_ctx* ctx, const void* local_device, int n_local, void* out_device);
void lisreg_comm_destroy(lisreg_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* LISREG_H_ */
