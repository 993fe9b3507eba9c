// lisreg_internal.hpp — device-side data contract shared by the HIP kernels and the C-ABI host layer.
// gfx950 only (wave64, 160 KiB LDS/CU, 256 CUs in 8 XCDs). Not a public header.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/lisreg.h"

namespace lisreg {

constexpr int kBlockQ     = 256;   // queries per workgroup in the correspondence kernel (4 waves)
constexpr int kStageCap   = 1024;  // target points staged in LDS per chunk (16 KiB)
constexpr int kNumAcc     = 28;    // 21 upper-tri AtA + 6 AtB + 1 count
constexpr int kResultSize = 12;    // floats per item in the result block
constexpr int kTraceStride = LISREG_TRACE_STRIDE;
constexpr int kGraphK     = 64;    // neighbour-list length of the target's k-NN graph (search_mode 3)

// Uniform-grid index over one target cloud (replaces one pcl::KdTreeFLANN, odomEstimationNode.cpp:602-603).
// Points are bucket-sorted by cell; linear cell id = (ix*ny + iy)*nz + iz (z fastest), so a z-range of one
// (ix,iy) column is one contiguous run of the sorted array.
struct GridIndex {
    const float4* pts;         // [n] sorted; .w = bit-cast ORIGINAL index in the caller's cloud
    const int*    cell_start;  // [nx*ny*nz + 1]
    int   n;
    float ox, oy, oz;          // grid origin (min corner)
    float cell, inv_cell;
    int   nx, ny, nz;
    // k-NN graph over the sorted points (search_mode 3; null otherwise): nbr[s * kGraphK + j] = (x, y, z, sorted position as int
    // bits) of the j-th nearest other point of sorted point s, ascending by distance — the row carries the neighbours' COORDINATES,
    // so a scan reads consecutive bytes of one row instead of gathering a point per id; padded entries = (s's own coordinates, -1).
    // nbr_meta[s] = (rho^2, count bits) where every point closer to s than rho is in the list.
    const float4* nbr;
    const float2* nbr_meta;
    // cell rows (search_mode 5; null otherwise): the same kind of row — kGraphK entries (x, y, z, sorted position), ascending by distance,
    // (rho^2, count) beside it — but anchored at a LOCATION, not at a point: the centre of a grid cell ("coarse" row) or of one of its eight
    // octants (those with a point within a quarter cell of their box).  crow_tab[cell] = -2: no target point within two cells
    // of this one; -1: no row (capacity); else (first row << 8) | octant mask, rows = [centre, the mask's octants in ascending order].  A query reads the row of the cell / octant it falls into, whatever
    // its distance from the surface, so the certificate radius c5 + |q - centre| is bounded by the cell size in every Gauss-Newton iteration.
    const float4* crow;
    const float2* crow_meta;
    const int*    crow_tab;
    // query marks (round 6; null: every populated cell gets its rows): one bit per cell — column (ix * ny + iy) owns qmark_w consecutive
    // 32-bit words, bit iz & 31 of word iz >> 5 — set by k_query_marks for the cells the batch's queries fall into under their INITIAL
    // poses.  The row build of a run leaves out the cells no query of the batch comes within a metre of (crow_tab -1: a query that gets
    // there all the same takes the cell walk — results never depend on the marks).
    unsigned*     qmark;
    int           qmark_w;
};

// Cell-row entries carry plain ids (the sorted position of the listed point, -1 in padded entries) — the position tags of round 5 went with the
// LDS-staged scan that read them (profiles/r05_xp_mfma_stage_select.patch); cell-row targets are limited only by their 32-bit record offsets.

// centre coordinate of cell h (off = 0.5f) or of one of its halves (0.25f / 0.75f) along one axis: the build and the scan must agree to
// the bit (the scan re-derives the list distances of a row from this centre) although their translation units are compiled with different
// contraction settings: ONE fused multiply-add, whatever the flags (hipcc's __fmul_rn / __fadd_rn are plain operators and do get contracted)
__device__ __forceinline__ float crow_centre(float origin, float cell, int h, float off) { return __builtin_fmaf((float)h + off, cell, origin); }

// One (item, kind) source segment; kind 0 = edge features vs corner target, 1 = planar features vs surf target.
struct Segment {
    const float4* src;         // caller's device records (x,y,z,payload), original order
    int   n;
    int   item;
    int   kind;
    int   target;              // index into the GridIndex array (slot*2 + kind)
    int   flat_base;           // first position of this segment in the batch-wide sorted arrays
    int   bucket_base;         // first sort bucket of this segment
    int   tnx, tny, tnz;       // sort-tile grid dims
    float tox, toy, toz;       // sort-tile grid origin
    float inv_tile;
};

// One target cloud of a batched index build (many submaps at once: loop-closure candidate batches).
struct TargetSeg {
    const float4* raw;         // caller's records
    float4*       sorted_out;  // [n]
    int*          cell_start_out;  // [n_cells + 1]
    int   n, n_cells;
    int   flat_base;           // first element in the batch-wide numbering
    int   bucket_base;         // first bucket
    float ox, oy, oz, inv_cell;
    int   nx, ny, nz;
    int   grid_id;             // entry of the device GridIndex array this target fills (slot * 2 + kind)
    int   strip_base;          // strip form of the build (lisreg_index.hip): first strip of this target in the batch-wide tables,
    int   ystrip, nstrips;     // iy cells per strip, strips per ix
};

// One workgroup of the correspondence kernel (also the unit of the source-key kernel).
struct BlockDesc {
    int seg;                   // segment id
    int start;                 // first query of the block, relative to the segment
    int count;                 // <= kBlockQ
    int item;
};

// Literals of lisreg_params in the form the kernels want (passed by value as a kernel argument).
struct DevParams {
    float tau, line_ratio, plane_tol, accept_s, conv_deg, conv_cm, eig_thresh;
    int   min_corr, use_label, emulate_shadow, skip_empty, fixed_iters, bound, edge_min, surf_min, use_imu;
    float imu_w, rot_tol, z_tol;
    int   ties;                // "canonical_ties": equal distances resolved by (distance, original index) in every front-end
    int   exact;               // "exact_arithmetic" (the launches pick launch_assoc_exact; the host supplies the pose caches' sin / cos)
    float wtab[32];            // w = (float)(2.0 - LabelSorce[label]) precomputed on the host
    int   cell_anchor_until;   // graph front-end: GN iterations 1 .. this also try an anchor out of the query's own grid column
    int   n_guard_failed;      // registrations of the batch that fail the feature-count guard (known on the host): the done counter's value after a reset
    int*  reach_miss;          // cell rows built under "row_reach": queries that found their cell without rows (-1) are counted here (null: not counted)
    int   freeze_pose;         // timing experiments only (env LISREG_XP_FREEZE_POSE): the solve leaves T as it is, so that every launch of a run
                               // sees the same queries whatever a variant under test writes into the normal equations
};

// Mutable per-registration state (device resident for the whole GN loop — no host sync per iteration).
struct ItemState {
    float T[6];                // transformTobeMapped
    float T_init[6];
    float P[36];               // matP
    float M[12];               // trans2Affine3f(T) for the NEXT correspondence launch (uniform -> scalar loads)
    float sc[6];               // srx, crx, sry, cry, srz, crz of LMOptimization (:862-867) for the same T
    float jk[21];              // the pose-only factors of LMOptimization's arx / ary / arz (:898-907), see jacobian_row
    int   iter;                // iterations started so far
    int   done;                // converged / exhausted: all later launches skip this item
    int   iters_out;           // iterCount as the reference reports it
    float deltaR, deltaT;
    int   degenerate;          // isDegenerate
    int   degenerate_in;
    int   n_corr;
    int   any_solved;
    int   guard_failed;        // NOT_ENOUGH_FEATURES
    int   blk_begin, blk_count;// this item's range in the partials array (corner blocks first, then surf)
    int   n_sc, n_ss;          // source sizes (guard)
    lisreg_imu imu;
};

// pcl::VoxelGrid geometry (filters/impl/voxel_grid.hpp): computed on the host from the bounding box
struct VoxelDesc {
    float    inv_leaf;             // inverse_leaf_size_ = 1 / leaf (float)
    int      min_b0, min_b1, min_b2;
    int      mul1, mul2;           // divb_mul_ = (1, div0, div0*div1)
    uint32_t span;                 // consecutive voxel indices per sort bucket
};

// K clouds voxel-gridded by one sort (lisreg_voxel_downsample_multi): point i of the concatenation belongs to cloud s with off[s] <= i
constexpr int kVoxelMultiMax = 8;
struct VoxelMulti {
    int       k;
    int       off[kVoxelMultiMax + 1];
    int       bucket_base[kVoxelMultiMax + 1];
    uint32_t  idx_base[kVoxelMultiMax + 1];    // first joint voxel index of cloud s (the clouds' index spaces laid end to end, < 2^32 in all)
    VoxelDesc d[kVoxelMultiMax];               // own geometry and own bucket span per cloud
};

// ---- launchers (lisreg_kernels.hip); all enqueue on `st`, none synchronise --------------------------------
struct SortBuffers {           // scratch for one deterministic bucket sort
    int*      hist;            // [n_buckets] counts (consumed by the scatter)
    int*      bucket_start;    // [n_buckets + 1]
    int*      scan_tmp;        // [ceil(n_buckets / 2048) + 1]
    uint32_t* elem_bucket;     // [n_elems] bucket of each ORIGINAL element
    uint32_t* elem_sub;        // [n_elems] sub-key of each original element
    uint32_t* tmp_bucket;      // [n_elems] by scattered position
    uint32_t* tmp_sub;
    int*      tmp_idx;
    float4*   tmp_pts;         // [n_elems] batched target build: the points themselves by scattered position
};

// bounding box of device records -> bbox6 = {minx,miny,minz,maxx,maxy,maxz} (device)
void launch_bbox(const float4* pts, int n, float* bbox6, float* scratch /* >= 6*256 floats */, hipStream_t st);
// target index: sorted_out / cell_start_out for the grid described by `g` (g.pts / g.cell_start ignored)
void launch_build_target(const float4* pts, int n, GridIndex g, float4* sorted_out, int* cell_start_out,
                         int n_cells, SortBuffers sb, hipStream_t st);
// several target indexes in one launch sequence (blocks: seg = TargetSeg id, start/count = point chunk)
void launch_build_targets_batched(const BlockDesc* blocks, int n_blocks, const TargetSeg* tsegs, int n_tsegs,
                                  int n_elems, int n_buckets, SortBuffers sb, hipStream_t st);
// the same build in strip form (lisreg_index.hip): partition into strips (one ix, a run of iy), then one workgroup per
// (target, strip) in LDS.  chunks: seg = TargetSeg id, start/count = point chunk of at most kPartChunkHost points.
constexpr int    kPartChunkHost = 4096;
constexpr int    kMaxStrips     = 8192;          // strips of one target the partition histogram holds
constexpr size_t kStripLdsLarge = 150 * 1024;    // LDS budget of the one-per-CU variant: cell table of a strip + 6 bytes per point
struct StripBuffers {
    int*      cnt;             // [n_strips + 1] populations       } one allocation, zeroed per build
    int*      fill;            // [n_strips + 1] scatter cursors   }
    int*      start;           // [n_strips + 1] first flat position of every strip
    int*      scan_tmp;        // [n_strips / 2048 + 4]
    float4*   tmp_pts;         // [n_elems] records by strip
    uint32_t* slot_idx;        // [n_elems] } slot tables of strips too big for LDS
    uint32_t* slot_pos;        // [n_elems] }
    hipStream_t side;          // optional: stream + events for running the big-strip variant underneath the small-strip one
    hipEvent_t  ev_fork, ev_join;
};
int  launch_build_targets_strips(const BlockDesc* chunks, int n_chunks, const TargetSeg* tsegs, int n_tsegs, int n_strips, int max_units,
                                 int max_strip_cells, int cap_small, StripBuffers sb, hipStream_t st, int* zero_ints_known = nullptr);
// k-NN graph of every target in `tsegs` (grids[t.grid_id] must describe the finished index and carry nbr / nbr_meta)
void launch_build_graph(const BlockDesc* blocks, int n_blocks, const TargetSeg* tsegs, const GridIndex* grids, hipStream_t st);
void launch_build_graph_one(GridIndex g, hipStream_t st);
// cell rows of one target (search_mode 5).  classify: need[cell] = rows the cell wants (0 / 1 / 8), scan[cell] = first row, scan[n_cells] = rows
// in all; build: crow_tab, then one wave per row (at most cap_rows of them: cells past the capacity get no row and their queries walk)
struct CrowBuffers { int* need; int* omask; int* scan; int* scan_tmp; int cap_rows;
                     const unsigned* reach = nullptr; };       // reach: the query marks grown by a metre (same layout as GridIndex::qmark), or null
// Query marks (lisreg_index.hip): launch_query_marks sets, for every source point of the batch under its item's CURRENT pose cache (the
// initial pose right after launch_reset_items), the bit of the grid cell it falls into (clamped into the grid: what the cell-row scan
// does with a query outside); launch_reach_dilate ORs the marks of the (2 D + 1)^3 block around every cell into `reach`, D = ceil(1 m / cell) >= 2.
void launch_query_marks(const BlockDesc* blocks, int n_blocks, const Segment* segs, const GridIndex* grids, const ItemState* items, hipStream_t st);
void launch_reach_dilate(GridIndex g, unsigned* reach, hipStream_t st);
// The octant masks (cb.omask) are accumulated by atomic ORs and must start at zero: launch_crow_build hands every cell's mask back as zero
// once it has read it, so a classify that FOLLOWS a build on the same buffer needs no memset (two fill launches and their gaps on the
// critical path of every step).  *omask_zero_cells = cells of the buffer's head known to be zero (0: unknown); both launchers keep it.
void launch_crow_classify(GridIndex g, int n_cells, CrowBuffers cb, hipStream_t st, int* omask_zero_cells = nullptr);
void launch_crow_build(GridIndex g, int n_cells, CrowBuffers cb, hipStream_t st, int* omask_zero_cells = nullptr);
// the same for the corner ([0]) and the surf ([1]) target of one slot in one launch sequence on one stream (round 6: no side stream, no event hops)
void launch_crow_rows_pair(const GridIndex g[2], const int n_cells[2], const CrowBuffers cb[2], hipStream_t st, int* omask_zero_cells[2]);
// sources of a whole batch: tile-sort every segment under its item's initial pose
void launch_sort_sources(const BlockDesc* blocks, int n_blocks, const Segment* segs, int n_segs,
                         const ItemState* items, int n_elems, int n_buckets, SortBuffers sb, float4* sorted_all, int* order_all,
                         hipStream_t st);
void launch_reset_items(ItemState* items, int n_items, DevParams prm, int* done_counter, hipStream_t st, int* zero_too = nullptr);
// (launch_finalize with done_counter != null also leaves every registration reset for the NEXT run of the prepared batch: see run_impl)
void launch_assoc(const BlockDesc* blocks, int n_blocks, const Segment* segs, const GridIndex* grids,
                  const ItemState* items, DevParams prm, const float4* sorted_all, double* partials,
                  int mode /* 0 LDS-staged workgroup box, 1 per-lane grid walk, 3 k-NN graph scan (walk without a certificate) */,
                  int* nn, int n_elems, float first_pass_r2,
                  bool wide /* centre-first walk for the early iterations whose seeds / anchors are stale */,
                  int graph_hops /* mode 3: neighbour lists scanned per query before the cell walk takes over */,
                  unsigned long long* counters /* may be null */,
                  int* dbg_nn /* may be null; modes 1 and 3: [6][n_elems] original indices of each query's neighbours + accept flag */,
                  int lanes_q /* mode 1: 1, or 8 lanes per query (small batches) */,
                  const BlockDesc* blocks_q, int n_blocks_q /* lanes_q = 8: descriptors of kBlockQ / 8 queries for the search */,
                  float4* coef, int* coef_ok /* lanes_q = 8: per-query coefficients handed to k_rows_reduce */,
                  const int* xcd_order /* mode 3: dispatch position -> block id (launch_xcd_order), or null */, hipStream_t st);
// the same launch with the reference's arithmetic (lisreg_assoc.hip built with -DLISREG_EXACT=1 -ffp-contract=off): the parity anchor
void launch_assoc_exact(const BlockDesc* blocks, int n_blocks, const Segment* segs, const GridIndex* grids,
                        const ItemState* items, DevParams prm, const float4* sorted_all, double* partials,
                        int mode, int* nn, int n_elems, float first_pass_r2, bool wide, int graph_hops,
                        unsigned long long* counters, int* dbg_nn, int lanes_q, const BlockDesc* blocks_q, int n_blocks_q,
                        float4* coef, int* coef_ok, const int* xcd_order, hipStream_t st);
// test hook: the device functions of the residual models on caller-given neighbourhoods (lisreg_test_fit_models)
void launch_test_fit(int kind, int n, const float* nb15, const float* q3, DevParams prm, float* out, hipStream_t st);
void launch_test_fit_exact(int kind, int n, const float* nb15, const float* q3, DevParams prm, float* out, hipStream_t st);
// XCD-aware dispatch order of a shared-target batch: blocks ranked by the azimuth of their middle query around the target centre
// (by_target: a batch with more than one target — blocks ranked by target slot instead, so that every XCD works on whole targets)
void launch_xcd_order(const BlockDesc* blocks, int n_blocks, const Segment* segs, const GridIndex* grids, const ItemState* items,
                      const float4* sorted_all, bool by_target, int* keys, int* order, hipStream_t st);
// exact build: poses to a dense [n][6] array / pose caches (M, sin-cos, Jacobian factors) rebuilt from host-computed trig values [n][6]
void launch_pose_gather(const ItemState* items, int n_items, float* T_out, hipStream_t st);
void launch_pose_cache_from_trig(ItemState* items, int n_items, const float* trig, hipStream_t st);
void launch_solve(ItemState* items, int n_items, DevParams prm, const double* partials, float* trace,
                  int trace_cap, int* done_counter, hipStream_t st);
void launch_finalize(ItemState* items, int n_items, DevParams prm, float* results, hipStream_t st, int* reset_done_counter = nullptr);

// §8 f-1 (lisreg_index.hip)
void launch_voxel_sort(const float4* pts, int n, VoxelDesc d, int n_buckets, SortBuffers sb, int* order, uint32_t* sidx,
                       int* head, int* slot /* [n+1], slot[n] = number of voxels */, hipStream_t st);
struct VoxelHandOut { int k; int vo[kVoxelMultiMax + 1]; float4* out[kVoxelMultiMax]; };
void launch_bbox_multi(const float4* cat, const VoxelMulti& m, float* bbox_out /* [6 * k] */, float* scratch /* >= 6 * 64 * k floats */, hipStream_t st);
struct BboxJobs { int k; int n[kVoxelMultiMax]; const float4* pts[kVoxelMultiMax]; };     // n == 0: the empty box (3e38, -3e38)
void launch_bbox_jobs(const BboxJobs& j, float* bbox_out /* [6 * k] */, float* scratch /* >= 6 * 64 * k floats */, hipStream_t st);
struct CloudRef { const float4* pts; int n; int pad_; };
void launch_bbox_refs(const CloudRef* refs_dev, int k, float* bbox_out /* [6 * k] */, float* scratch /* >= 6 * 64 * k floats */, hipStream_t st);
void launch_concat_jobs(const BboxJobs& j, const VoxelMulti& m, float4* cat, hipStream_t st);
void launch_multi_bounds(const int* slot, const VoxelMulti& m, int* out /* [k + 1] */, hipStream_t st);
void launch_hand_out(const float4* src, const VoxelHandOut& h, hipStream_t st);
void launch_voxel_sort_multi(const float4* pts, int n, const VoxelMulti& m, int n_buckets, SortBuffers sb, int* order, uint32_t* sidx,
                             int* head, int* slot, hipStream_t st);
void launch_voxel_centroids(int n, int n_vox, const float4* pts, const uint32_t* labels, int w_mode, const int* order,
                            const int* head, const int* slot, int* vstart /* [n_vox+1] */, float4* out_pts,
                            uint32_t* out_labels /* may be null */, hipStream_t st);
// raw PCL structs already in device memory (n records of `stride` bytes) -> 16-byte records
void launch_pack_cloud(const void* raw_dev, size_t n, int stride, int has_label, float4* out, hipStream_t st);
void launch_transform_cloud(const float4* in, int n, const float* M12_dev, float4* out, hipStream_t st);
struct Mat12 { float m[12]; };
void launch_transform_cloud_m(const float4* in, int n, const float M12_host[12], float4* out, hipStream_t st);   // matrix by value
// number of consecutive source points further apart than thr (coherence probe for sort_sources = auto)
void launch_count_jumps(const BlockDesc* blocks, int n_blocks, const Segment* segs, float thr, int* jumps, hipStream_t st);
// out[n+1] = exclusive scan of in[n] (out[n] = total); tmp: n/2048 + 4 ints
void launch_exclusive_scan(const int* in, int* out, int* tmp, int n, hipStream_t st);

// §8 f-2 (lisreg_features.hip): range-image projection + LOAM feature extraction for one scan
struct FeatureBuffers {
    int*   owner;      // [HW]      input index owning each range-image pixel (INT_MAX = empty)
    int*   flag;       // [HW+16]   scratch flags for the scans
    int*   pos;        // [2*(HW+17)] exclusive scans: [0, HW] pixel -> extracted position; upper half: surface scan
    int*   scan_tmp;   // [HW/2048 + 4]
    int*   col;        // [HW+16]   pointColInd
    float* range;      // [HW+16]   pointRange
    int*   src;        // [HW+16]   extracted position -> input index   (list 0: deskewed)
    float* curv;       // [HW+16]   cloudCurvature
    int*   picked;     // [HW+16]   cloudNeighborPicked
    int*   label;      // [HW+16]   cloudLabel
    int*   ring_lists; // [H][3][128] per-ring corner / sharp-corner / sharp-surface picks (extracted positions)
    int*   ring_counts;// [H][4]
    int*   lists;      // [4][HW+16] corner, surface, corner_sharp, surface_sharp as INPUT indices
    int*   counts;     // [8]       deskewed, corner, surface, corner_sharp, surface_sharp
};
void launch_extract_features(const float4* pts, const uint32_t* rings /* null: ring = payload & 0xffff */, int n,
                             lisreg_feature_params P, FeatureBuffers fb, hipStream_t st, int n_sweeps = 1);
// batched extraction (S sweeps stacked into one range image of S x H rows; ring offsets need [H_total + 1][3] ints in fb.flag)
void launch_feature_batch_rows(const float4* cat, int n, const int* offsets /* host [n_sweeps + 1] */, int n_sweeps, int H, int rate,
                               uint32_t* rows, hipStream_t st);
void launch_feature_batch_bounds(int n_sweeps, int H, int hw_sweep, FeatureBuffers fb, int hw_total, int* B /* [n_sweeps + 1][5] */, hipStream_t st);
void launch_feature_batch_gather(const float4* cat, const int* idx, const void* jobs_dev /* {float4* dst; int begin, count} per sweep */,
                                 int n_sweeps, int max_count, hipStream_t st);
// IMU de-skew tables in device memory (lisreg_deskew, laserProcessing.cpp:222-266)
struct DeskewTables {
    const double* time; const double* rx; const double* ry; const double* rz;
    int    imu_pointer_cur;
    double time_scan_cur;
};
// rotate every pixel-owning point of pts_copy (a private copy of the sweep) into the first owner's frame
void launch_deskew(const int* owner, int hw, const float* times_dev, DeskewTables T, int* first_dev, float* rsi_dev, float4* pts_copy,
                   hipStream_t st);
void launch_gather_points(const float4* pts, const int* idx, int n, float4* out, hipStream_t st);
// §8 f-3 / f-4 building blocks (lisreg_nn1.hip): exact k = 1 queries on a GridIndex that lives in device memory
void launch_nn1(const float4* q, int n, const GridIndex* grid_dev, float max_dist, int* idx_out, float* d2_out, hipStream_t st);
void launch_dynamic_flags(const float4* pts, int n, const GridIndex* grid_dev, float center_radius, float near_thre, float dmin,
                          float dmax, int* flag, hipStream_t st);
// §8 f-4: device-side state of one pcl::IterativeClosestPoint::align
struct IcpState {
    float  F[16];            // final_transformation_ (row-major 4x4)
    float  Tm[16];           // transformation_ of the last iteration
    int    iters, done, state, converged, n_corr, fit_n;
    double prev_mse, cur_mse, fit_sum;
    double first_mse;        // MSE of the first iteration, < 0 while its MSE test has not been reached
    int    defer_first, pad_;// chained batch: the first MSE comparison is left to the host (lisreg_icp_align_batch)
};
// one alignment of an ICP batch (device): blocks blk0 .. blk0 + nblk of the launch work on it, one partial row per block
struct IcpItem {
    const float4*    src;    // source records
    float4*          cur;    // working copy (input_transformed)
    int*             nn;     // last iteration's neighbour of every source point (position in the sorted target, -1 none): the next search's seed
    const GridIndex* grid;   // its target's k = 1 index
    int n, blk0, nblk;
    int pad_;
};
int  icp_batch_lanes(long long total_points);
int  icp_batch_blocks(int n, int q);
int  icp_blocks(int n);
// one ICP iteration on the working copy `cur` (input_transformed): apply st->Tm in place, correspondences + sums
// (partials: icp_blocks(n) * 17 doubles), then transform estimate + convergence (st->Tm = the new transformation_)
void launch_icp_assoc(const IcpItem* items, int n_items, int total_blocks, int q, IcpState* states, float cap2, double* partials,
                      hipStream_t stream);
void launch_icp_solve(const IcpItem* items, int n_items, int q, IcpState* states, const double* partials,
                      int max_iters, double eps_t, double eps_mse, int* n_done, hipStream_t stream);
void launch_icp_fitness_batch(const IcpItem* items, int n_items, int total_blocks, int q, IcpState* states, double* partials, hipStream_t stream);
// one OptimizedICPGN iteration (partials: icp_blocks(n) * 22 doubles); st->F is T, st->iters counts the applied steps
void launch_icpgn_iteration(const float4* src, int n, const GridIndex* grid_dev, IcpState* st, float cap2, double* partials,
                            hipStream_t stream);
void launch_icp_fitness(const float4* src, int n, const GridIndex* grid_dev, IcpState* st, double* partials, hipStream_t stream);
void launch_bbx_flags(const float4* pts, int n, const double b[6], int delete_box, int* flag, hipStream_t st);
// stable compaction: idx_out[0..count) = indices i with flag[i] != 0, ascending; pos [n+1]
void launch_compact(int n, const int* flag, int* pos, int* scan_tmp, int* idx_out, int* count_out, hipStream_t st);
struct SemanticGather { float4* out[5]; int count[5]; };     // count 0: that class is not wanted
void launch_semantic_gather(const float4* pts, const int* idx /* [5][n] */, int n, const SemanticGather& g, hipStream_t st);
void launch_semantic_split(const float4* pts, const uint32_t* labels /* null: payload */, int n, const uint32_t map[32],
                           int* flag /* [n] */, int* pos /* [n+1] */, int* scan_tmp, int* idx_out /* [5*n] */, int* counts /* [5] */,
                           hipStream_t st);

}  // namespace lisreg
