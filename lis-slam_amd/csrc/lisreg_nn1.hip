// lisreg_nn1.hip — exact k = 1 nearest-neighbour queries on the uniform-grid index, and the two order-preserving
// filters of SURVEY.md §8 f-3 (submap / local-map maintenance) that are actual compute:
//   * SubMapManager::map_scan_feature_pts_distance_removal (/root/reference/src/include/subMap.h:1062-1100): keep a
//     point iff it lies outside center_radius (x,y) or its nearest map point is at squared distance in
//     (near^2, dmin^2) or beyond dmax^2 — pcl::search::KdTree::nearestKSearch(k = 1) replaced by the grid walk;
//   * SubMapManager::bbx_filter (:1124-1152): strict-inequality box crop, inside or outside.
// The k = 1 walk is the k = 5 walk of lisreg_assoc.hip with one register of state: passes of doubling radius
// (0.5, 1, 2, ... m) each visit every cell that can hold a point closer than min(best so far, pass radius), so the
// usual sub-metre neighbour costs one tight pass while the caps (3 m here, 10 m for ICP) stay exact.
#include "lisreg_internal.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace lisreg {

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const v4f* gptr_f4;
typedef __attribute__((address_space(1))) const int* gptr_i32;

__device__ __forceinline__ int gcoord(float v, float origin, float inv_cell) { return (int)floorf((v - origin) * inv_cell); }

// nearest target point with squared distance <= max_d2 (index into g.pts, -1 if none); *d2_out = its squared distance.
// Q lanes (1, 4 or 8 neighbouring lanes of a wave) share one query: the (x, y) columns of every pass are dealt round-robin
// to the Q lanes and the per-lane winners are merged with Q-wide butterflies after each pass.  A single scan is only ~1 800
// waves of queries — a fifth of the chip's wave slots — so one-lane-per-query leaves the walk latency-bound; splitting a
// query's columns over Q lanes shortens every dependent-load chain by Q and fills the machine (launchers pick Q from n).
// All Q lanes of a query must call this together (same q, g, max_d2) and all of them get the merged result.
// seed (optional): the position of a target point that is probably near — last ICP iteration's neighbour.  It enters as an ordinary
// candidate (same distance expression, same tie rule), so the result does not depend on it; what it buys is a first pass whose radius is
// the distance to that point instead of 0.5 m.
template <int Q>
__device__ __forceinline__ int nn1_search(float qx, float qy, float qz, const GridIndex& g, float max_d2, float* d2_out, int seed = -1)
{
    constexpr float kEps = 1e-3f;
    const gptr_f4 pts = (gptr_f4)g.pts;
    const gptr_i32 cells = (gptr_i32)g.cell_start;
    float best = __uint_as_float(__float_as_uint(max_d2) + 1u);      // strictly-less test below must admit d2 == max_d2
    int bi = -1, bw = 0x7fffffff;                                    // bw = original index of the best (ties: smallest wins)
    if (g.n <= 0) { *d2_out = best; return -1; }
    const int sub = Q > 1 ? (int)(threadIdx.x & (Q - 1)) : 0;
    if (seed >= 0 && seed < g.n) {
        const v4f c = pts[seed];
        const float ex = qx - c.x, ey = qy - c.y, ez = qz - c.z;
        const float d2 = ex * ex + ey * ey + ez * ez;
        if (d2 < best) { best = d2; bi = seed; bw = __float_as_int(c.w); }
    }
#define LISREG_NN1_TRY(c_, j_) do { \
        const float ex_ = qx - (c_).x, ey_ = qy - (c_).y, ez_ = qz - (c_).z; \
        const float d2_ = ex_ * ex_ + ey_ * ey_ + ez_ * ez_;        /* flann::L2_Simple order */ \
        const int w_ = __float_as_int((c_).w); \
        if (d2_ < best || (d2_ == best && w_ < bw)) { best = d2_; bi = (j_); bw = w_; } } while (0)
    for (float r = 0.5f;; r *= 2.f) {
        const float lim = fminf(r * r, best);
        const float rad = __builtin_amdgcn_sqrtf(lim) * 1.0001f + kEps;
        const int cx0 = max(gcoord(qx - rad, g.ox, g.inv_cell), 0), cx1 = min(gcoord(qx + rad, g.ox, g.inv_cell), g.nx - 1);
        const int cy0 = max(gcoord(qy - rad, g.oy, g.inv_cell), 0), cy1 = min(gcoord(qy + rad, g.oy, g.inv_cell), g.ny - 1);
        const int cz0 = max(gcoord(qz - rad, g.oz, g.inv_cell), 0), cz1 = min(gcoord(qz + rad, g.oz, g.inv_cell), g.nz - 1);
        if (cz0 <= cz1 && cx0 <= cx1 && cy0 <= cy1) {
            const int ncy = cy1 - cy0 + 1;
            int ix = cx0, iy = cy0 + sub;                            // lane `sub` takes columns sub, sub + Q, ... in row-major order
            while (iy > cy1) { iy -= ncy; ++ix; }
            while (ix <= cx1) {
                const float xl = g.ox + (float)ix * g.cell, yl = g.oy + (float)iy * g.cell;
                const float dx = fmaxf(fmaxf(xl - qx, qx - (xl + g.cell)) - kEps, 0.f);
                const float dy = fmaxf(fmaxf(yl - qy, qy - (yl + g.cell)) - kEps, 0.f);
                if (dx * dx + dy * dy < fminf(best, lim)) {
                    const int base = (ix * g.ny + iy) * g.nz;
                    const int js = cells[base + cz0], je = cells[base + cz1 + 1];
                    for (int j = js; j < je; j += 4) {
                        const int l = je - 1;
                        const int j1 = min(j + 1, l), j2 = min(j + 2, l), j3 = min(j + 3, l);
                        const v4f c0 = pts[j], c1 = pts[j1], c2 = pts[j2], c3 = pts[j3];     // 4 loads in flight; a clamped
                        LISREG_NN1_TRY(c0, j); LISREG_NN1_TRY(c1, j1);                     // tail repeats the last point,
                        LISREG_NN1_TRY(c2, j2); LISREG_NN1_TRY(c3, j3);                    // which cannot beat itself
                    }
                }
                iy += Q;
                while (iy > cy1) { iy -= ncy; ++ix; }
            }
        }
        if (Q > 1) {
#pragma unroll
            for (int o = 1; o < Q; o <<= 1) {
                const float ob = __shfl_xor(best, o, 64);
                const int oi = __shfl_xor(bi, o, 64), ow = __shfl_xor(bw, o, 64);
                if (ob < best || (ob == best && ow < bw)) { best = ob; bi = oi; bw = ow; }
            }
        }
        if (best <= r * r || !(r * r < max_d2)) break;     // exact: everything within min(best, r) was visited (a NaN cap ends the search too)
    }
#undef LISREG_NN1_TRY
    *d2_out = best;
    return bi;
}

// The same search for ONE lane per query with the candidate loop flattened (round 5; the ICP kernels' batches).  nn1_search above costs a
// wavefront, per pass, the SUM over column steps of the longest run any lane has at that step, and every candidate the full tie rule
// (lane use 0.41-0.50, ~1 200 vector instructions per wavefront in the seeded iterations of configs[3]).  Here a lane first collects the
// non-empty runs of its columns (LDS, kNn1Cap per lane), then streams through them in ONE loop four candidates at a time — a wavefront pays
// its longest TOTAL — and the common case of a group (a unique closest candidate, not at the distance of the best so far) is a strict
// insertion of that one candidate; a group whose minimum EQUALS the best so far, or is attained twice, takes the full rule
// (distance, then original index) for its four.  The result is the lexicographic minimum over the same candidates either way.
// The seed enters one ulp ABOVE its distance: the walk meets the seed point itself (its cell is inside the final box) and inserts it at
// its true distance by the strict rule — so a real tie with it is an equality like any other, and meeting it is not.
constexpr int kNn1Cap = 8;
#ifndef LISREG_ICP_FLAT
#define LISREG_ICP_FLAT 1
#endif
constexpr bool kIcpFlat = LISREG_ICP_FLAT != 0;
__device__ __forceinline__ int nn1_search_flat(float qx, float qy, float qz, const GridIndex& g, float max_d2, float* d2_out, int seed,
                                               int2 (*s_runs)[256])
{
    constexpr float kEps = 1e-3f;
    // records and table entries are addressed base (scalar registers) + 32-bit byte offset: a target of the ICP kernels has fewer than 2^28
    // points (icp_run checks) and a grid at most 2^24 cells (make_grid)
    typedef __attribute__((address_space(1))) const char* gptr_c;
    const unsigned long long pa_ = (unsigned long long)g.pts, ca_ = (unsigned long long)g.cell_start;
    // (the builtin returns a signed int: each half goes through `unsigned` before it is widened)
    const gptr_c pts_b = (gptr_c)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(pa_ >> 32)) << 32) |
                                  (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)pa_));
    const gptr_c cells_b = (gptr_c)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(ca_ >> 32)) << 32) |
                                    (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ca_));
#define LISREG_NN1_PT(i_) (*(gptr_f4)(pts_b + ((unsigned)(i_) << 4)))
#define LISREG_NN1_CELL(i_) (*(gptr_i32)(cells_b + ((unsigned)(i_) << 2)))
    float best = __uint_as_float(__float_as_uint(max_d2) + 1u);      // strictly-less test below must admit d2 == max_d2
    int bi = -1;
    if (g.n <= 0) { *d2_out = best; return -1; }
    const int tid = threadIdx.x;
    float seed_d2 = 0.f;
    int seed_i = -2;
    if (seed >= 0 && seed < g.n) {
        const v4f c = LISREG_NN1_PT(seed);
        const float ex = qx - c.x, ey = qy - c.y, ez = qz - c.z;
        const float d2 = ex * ex + ey * ey + ez * ez;
        if (d2 < best) { seed_d2 = d2; seed_i = seed; best = __uint_as_float(__float_as_uint(d2) + 1u); bi = seed; }
    }
    for (float r = 0.5f;; r *= 2.f) {
        const float lim = fminf(r * r, best);
        const float rad = __builtin_amdgcn_sqrtf(lim) * 1.0001f + kEps;
        const int cx0 = max(gcoord(qx - rad, g.ox, g.inv_cell), 0), cx1 = min(gcoord(qx + rad, g.ox, g.inv_cell), g.nx - 1);
        const int cy0 = max(gcoord(qy - rad, g.oy, g.inv_cell), 0), cy1 = min(gcoord(qy + rad, g.oy, g.inv_cell), g.ny - 1);
        const int cz0 = max(gcoord(qz - rad, g.oz, g.inv_cell), 0), cz1 = min(gcoord(qz + rad, g.oz, g.inv_cell), g.nz - 1);
        int ix = cx0, iy = cy0;
        if (cz0 > cz1 || cx0 > cx1 || cy0 > cy1) ix = cx1 + 1;          // an empty box (a non-finite query has one): nothing to step through
        while (ix <= cx1) {
            int cnt = 0, grp = 0;
            // phase 1: up to kNn1Cap non-empty runs (grp: their groups of four candidates)
            while (cnt < kNn1Cap && ix <= cx1) {
                const float xl = g.ox + (float)ix * g.cell, yl = g.oy + (float)iy * g.cell;
                const float dx = fmaxf(fmaxf(xl - qx, qx - (xl + g.cell)) - kEps, 0.f);
                const float dy = fmaxf(fmaxf(yl - qy, qy - (yl + g.cell)) - kEps, 0.f);
                if (dx * dx + dy * dy < fminf(best, lim)) {
                    const int base = (ix * g.ny + iy) * g.nz;
                    const int js = LISREG_NN1_CELL(base + cz0), je = LISREG_NN1_CELL(base + cz1 + 1);
                    if (js < je) { s_runs[cnt][tid] = make_int2(js, je); ++cnt; grp += (je - js + 3) >> 2; }
                }
                if (++iy > cy1) { iy = cy0; ++ix; }
            }
            // phase 2: one loop over the collected candidates
            int rr = 0, j = 0, e = 0;
#pragma unroll 1
            for (int gi = 0; gi < grp; ++gi) {       // (a counted loop that leaves at its head only: no copies of the loop-carried values)
                if (j >= e) { const int2 t = s_runs[rr][tid]; j = t.x; e = t.y; ++rr; }
                const int l = e - 1;
                const int j1 = min(j + 1, l), j2 = min(j + 2, l), j3 = min(j + 3, l);
                const v4f c0 = LISREG_NN1_PT(j), c1 = LISREG_NN1_PT(j1), c2 = LISREG_NN1_PT(j2), c3 = LISREG_NN1_PT(j3);      // (a clamped tail repeats the run's last point)
                const float ax = qx - c0.x, ay = qy - c0.y, az = qz - c0.z, bx = qx - c1.x, by = qy - c1.y, bz = qz - c1.z;
                const float ux = qx - c2.x, uy = qy - c2.y, uz = qz - c2.z, vx = qx - c3.x, vy = qy - c3.y, vz = qz - c3.z;
                const float e0 = ax * ax + ay * ay + az * az, e1 = bx * bx + by * by + bz * bz;        // flann::L2_Simple order
                const float e2 = ux * ux + uy * uy + uz * uz, e3 = vx * vx + vy * vy + vz * vz;
                const float m = fminf(fminf(e0, e1), fminf(e2, e3));
                if (m <= best) {
                    const bool q0 = e0 == m, q1 = e1 == m, q2 = e2 == m, q3 = e3 == m;
                    const bool twice = (q0 && q1) || (q0 && q2) || (q0 && q3) || (q1 && q2) || (q1 && q3) || (q2 && q3);     // (lane masks: scalar work)
                    if (m == best || twice) {
                        // the full rule for the four: closer, or as close with the smaller original index
                        int bw = bi >= 0 ? __float_as_int(LISREG_NN1_PT(bi).w) : (int)0x80000000;     // (nothing held yet: equality with the cap's bound must not insert)
#define LISREG_NN1_FULL(e_, c_, j_) do { const int w_ = __float_as_int((c_).w); \
                            if ((e_) < best || ((e_) == best && w_ < bw)) { best = (e_); bi = (j_); bw = w_; } } while (0)
                        LISREG_NN1_FULL(e0, c0, j); LISREG_NN1_FULL(e1, c1, j1); LISREG_NN1_FULL(e2, c2, j2); LISREG_NN1_FULL(e3, c3, j3);
#undef LISREG_NN1_FULL
                    } else {
                        best = m;
                        bi = q0 ? j : (q1 ? j1 : (q2 ? j2 : j3));
                    }
                }
                j += 4;
            }
        }
        if (best <= r * r || !(r * r < max_d2)) break;     // exact: everything within min(best, r) was visited (a NaN cap ends the search too)
    }
    if (bi == seed_i && best > seed_d2) best = seed_d2;    // (cannot happen — the seed's cell is inside the last box — but the distance returned must be the point's)
    *d2_out = best;
    return bi;
#undef LISREG_NN1_PT
#undef LISREG_NN1_CELL
}

// A seed for a query that has none (the first ICP iteration): the nearest of up to four points out of the z-window [hz - 1, hz + 1] of the
// query's own (x, y) column of the grid — one contiguous run of the cell-sorted array, two table reads and four records.  Any point will do
// (nn1_search treats the seed as an ordinary candidate); a near one makes the first pass's radius the distance to it instead of a 0.5 m box
// that doubles until something is inside.  -1: the column has nothing there.
__device__ __forceinline__ int nn1_cell_seed(float qx, float qy, float qz, const GridIndex& g)
{
    if (g.n <= 0) return -1;
    const int hx = gcoord(qx, g.ox, g.inv_cell), hy = gcoord(qy, g.oy, g.inv_cell), hz = gcoord(qz, g.oz, g.inv_cell);
    if (hx < 0 || hx >= g.nx || hy < 0 || hy >= g.ny || hz < -1 || hz > g.nz) return -1;          // (a non-finite query saturates: out)
    const int z0 = min(max(hz - 1, 0), g.nz - 1), z1 = min(max(hz + 1, 0), g.nz - 1);
    const gptr_i32 cells = (gptr_i32)g.cell_start;
    const gptr_f4 pts = (gptr_f4)g.pts;
    const int base = (hx * g.ny + hy) * g.nz;
    const int js = cells[base + z0], je = cells[base + z1 + 1];
    if (js >= je) return -1;
    const int n = je - js;
    const int p1 = js + (n >> 2), p2 = js + (n >> 1), p3 = je - 1;
    const v4f c0 = pts[js], c1 = pts[p1], c2 = pts[p2], c3 = pts[p3];
    const float ax = qx - c0.x, ay = qy - c0.y, az = qz - c0.z, bx = qx - c1.x, by = qy - c1.y, bz = qz - c1.z;
    const float ex = qx - c2.x, ey = qy - c2.y, ez = qz - c2.z, fx = qx - c3.x, fy = qy - c3.y, fz = qz - c3.z;
    const float d0 = ax * ax + ay * ay + az * az, d1 = bx * bx + by * by + bz * bz;
    const float d2 = ex * ex + ey * ey + ez * ez, d3 = fx * fx + fy * fy + fz * fz;
    int a = js; float d = d0;
    if (d1 < d) { d = d1; a = p1; }
    if (d2 < d) { d = d2; a = p2; }
    if (d3 < d) { a = p3; }
    return a;
}

// map_scan_feature_pts_distance_removal (subMap.h:1076-1087).  The reference's unbounded search is cut at `cap2`, the largest
// FINITE threshold (thresholds default to FLT_MAX, whose square is +inf): beyond it the predicate no longer depends on the
// distance and its value is `keep_far`.
template <int Q>
__global__ __launch_bounds__(256) void k_dynamic_flags(const float4* __restrict__ pts, int n, const GridIndex* __restrict__ gp,
                                                       float center_r2, float near2, float dmin2, float dmax2, float cap2,
                                                       int keep_far, int* __restrict__ flag)
{
    const int i = (blockIdx.x * 256 + threadIdx.x) / Q;
    if (i >= n) return;
    const float4 p = pts[i];
    int keep;
    if (p.x * p.x + p.y * p.y > center_r2) keep = 1;
    else {
        float d2;
        const GridIndex g = *gp;
        const int bi = nn1_search<Q>(p.x, p.y, p.z, g, cap2, &d2);
        keep = (bi < 0) ? keep_far : (((d2 > near2 && d2 < dmin2) || d2 > dmax2) ? 1 : 0);
    }
    if ((threadIdx.x & (Q - 1)) == 0) flag[i] = keep;
}

// bbx_filter (subMap.h:1131-1144): float coordinate against double bounds
__global__ __launch_bounds__(256) void k_bbx_flags(const float4* __restrict__ pts, int n, double x0, double y0, double z0,
                                                   double x1, double y1, double z1, int delete_box, int* __restrict__ flag)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 p = pts[i];
    const bool in = (double)p.x > x0 && (double)p.x < x1 && (double)p.y > y0 && (double)p.y < y1 && (double)p.z > z0 && (double)p.z < z1;
    flag[i] = (in != (delete_box != 0)) ? 1 : 0;
}

__global__ __launch_bounds__(256) void k_compact_write(int n, const int* __restrict__ flag, const int* __restrict__ pos,
                                                       int* __restrict__ idx_out, int* __restrict__ count_out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) *count_out = pos[n];
    if (i < n && flag[i]) idx_out[pos[i]] = i;
}

// plain k = 1 query for a batch of points (exposed for tests and for the ICP row)
template <int Q>
__global__ __launch_bounds__(256) void k_nn1(const float4* __restrict__ q, int n, const GridIndex* __restrict__ gp, float max_d2,
                                             int* __restrict__ idx_out, float* __restrict__ d2_out)
{
    const int i = (blockIdx.x * 256 + threadIdx.x) / Q;
    if (i >= n) return;
    const GridIndex g = *gp;
    float d2;
    const int bi = nn1_search<Q>(q[i].x, q[i].y, q[i].z, g, max_d2, &d2);
    if ((threadIdx.x & (Q - 1)) != 0) return;
    idx_out[i] = bi < 0 ? -1 : __float_as_int(g.pts[bi].w);      // ORIGINAL index in the caller's map cloud
    d2_out[i] = d2;
}

// ---- §8 f-4: point-to-point ICP (pcl::IterativeClosestPoint, subMapOptmizationNode.cpp:2763-2833) ----------------------------
constexpr int kIcpAcc = 17;       // count, sum p (3), sum q (3), sum q_r * p_c (9), sum d2

__device__ __forceinline__ double wave_sum(double v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// The sums of pcl::IterativeClosestPoint are formed by ONE tree whatever the number of lanes per query, so that an alignment gives the
// same bits alone (four lanes per query: a scan-sized cloud does not fill the chip otherwise) and inside a batch of candidates (one
// lane per query).  Leaves = queries in index order (the non-lead lanes of a query hold exact zeros); levels = bit 0, 1, 2, ... of the
// query index: inside a wavefront by an ascending butterfly, across the four wavefronts of a workgroup as (w0 + w1) + (w2 + w3); a
// workgroup row covers 256 / Q queries, and the summing kernels first fold Q consecutive rows the same way (icp_row256) — from there
// on both forms hold one value per 256 queries.
// (the partner's value for the masks 1, 2 and 8 comes through DPP moves — quad permutes and a row rotation, vector-ALU work — instead of the
// LDS crossbar: a launch of the ICP correspondence kernel made 204 ds_bpermute per wavefront for its 17 double sums, half of its
// instructions and the busiest pipe; the masks 4, 16 and 32 stay there.  Same pairs, same order: the same bits.)
template <int M> __device__ __forceinline__ double lane_xor_d(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    if constexpr (M == 1)      { lo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true);  hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true); }    // quad_perm [1,0,3,2]
    else if constexpr (M == 2) { lo = __builtin_amdgcn_mov_dpp(lo, 0x4E, 0xF, 0xF, true);  hi = __builtin_amdgcn_mov_dpp(hi, 0x4E, 0xF, 0xF, true); }    // quad_perm [2,3,0,1]
    else if constexpr (M == 8) { lo = __builtin_amdgcn_mov_dpp(lo, 0x128, 0xF, 0xF, true); hi = __builtin_amdgcn_mov_dpp(hi, 0x128, 0xF, 0xF, true); }   // row_ror:8
    else if constexpr (M == 4) {                                                                                  // row_half_mirror (xor 7), then quad_perm [3,2,1,0] (xor 3)
        lo = __builtin_amdgcn_mov_dpp(__builtin_amdgcn_mov_dpp(lo, 0x141, 0xF, 0xF, true), 0x1B, 0xF, 0xF, true);
        hi = __builtin_amdgcn_mov_dpp(__builtin_amdgcn_mov_dpp(hi, 0x141, 0xF, 0xF, true), 0x1B, 0xF, 0xF, true); }
    else                       { lo = __shfl_xor(lo, M, 64); hi = __shfl_xor(hi, M, 64); }
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_up(double v)
{
    v += lane_xor_d<1>(v); v += lane_xor_d<2>(v); v += lane_xor_d<4>(v);
    v += lane_xor_d<8>(v); v += lane_xor_d<16>(v); v += lane_xor_d<32>(v);
    return v;
}

// the block -> item table of a batch: items[k].blk0 ascending, items[n_items].blk0 = total
__device__ __forceinline__ int icp_find_item(const IcpItem* __restrict__ items, int n_items, int blk)
{
    int lo = 0, hi = n_items - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (items[mid].blk0 <= blk) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// Which alignment, and which of its blocks, a workgroup of a batch launch works on: workgroup p = block p of the batch.  (An XCD-aware order —
// XCD x working through the alignments x, x + 8, ... so that an L2 holds the one target its CUs are searching — measured 4 % SLOWER on
// configs[3] in round 5: the search is not bound by L2 misses; profiles/r05_kernel_experiments.md section 6, r05_xp_icp_xcd_order.patch.)
__device__ __forceinline__ int icp_locate(const IcpItem* __restrict__ items, int n_items, int* b)
{
    const int item = icp_find_item(items, n_items, blockIdx.x);
    *b = (int)blockIdx.x - items[item].blk0;
    return item;
}

// value `k` of the b-th 256-query row of an item (rows of 256 / Q queries are folded pairwise: Q = 4 -> (r0 + r1) + (r2 + r3))
template <int kAcc>
__device__ __forceinline__ double icp_row256(const double* __restrict__ rows, int n_rows, int q, int b, int k)
{
    if (q == 1) return rows[(size_t)b * kAcc + k];
    const int r = 4 * b;
    const double r0 = rows[(size_t)r * kAcc + k];
    const double r1 = r + 1 < n_rows ? rows[(size_t)(r + 1) * kAcc + k] : 0.0;
    const double r2 = r + 2 < n_rows ? rows[(size_t)(r + 2) * kAcc + k] : 0.0;
    const double r3 = r + 3 < n_rows ? rows[(size_t)(r + 3) * kAcc + k] : 0.0;
    return (r0 + r1) + (r2 + r3);
}

__device__ __forceinline__ void apply4(const float* F, float x, float y, float z, float& ox, float& oy, float& oz)
{
    ox = ((F[0] * x + F[1] * y) + F[2] * z) + F[3];
    oy = ((F[4] * x + F[5] * y) + F[6] * z) + F[7];
    oz = ((F[8] * x + F[9] * y) + F[10] * z) + F[11];
}

// transformCloud(input_transformed, transformation_) of the previous iteration (icp.hpp applies it in place, in float, once per
// iteration — kept that way rather than re-deriving the points from the cumulative transform), then determineCorrespondences
// and the sums TransformationEstimationSVD needs.
template <int Q>
__global__ __launch_bounds__(256) void k_icp_assoc(const IcpItem* __restrict__ items, int n_items, const IcpState* __restrict__ states,
                                                   float cap2, double* __restrict__ partials)
{
    __shared__ double red[4][kIcpAcc];
    __shared__ int2 s_runs[Q == 1 ? kNn1Cap : 1][256];      // run lists of the flattened search (one lane per query)
    int blk;
    int item = icp_locate(items, n_items, &blk);
    // (both are the same in every lane; said so, the item's table entry, its grid and the target's base pointers stay in scalar registers)
    item = __builtin_amdgcn_readfirstlane(item); blk = __builtin_amdgcn_readfirstlane(blk);
    if (item < 0) return;
    const IcpState* stp = &states[item];
    if (stp->done) return;
    const IcpItem I = items[item];
    const int i = (int)(((long long)blk * 256 + threadIdx.x) / Q);
    const bool lead = (threadIdx.x & (Q - 1)) == 0;
    // The 16 sums other than the squared distance are the 4 x 4 outer product (1, q) x (1, p) summed over the correspondences: count = 1*1,
    // sum p = 1*p, sum q = q*1, sum q_r p_c.  They are reduced by a HALVING butterfly — 8 + 4 + 2 + 1 (+ 2) additions per lane instead of
    // 16 x 6 — over exactly the pairs, and in exactly the order (lane ^ 1, 2, 4, 8, 16, 32), of wave_sum_up: the same tree, the same bits as
    // the full butterflies this replaces (each addition has the same two operands; a + b = b + a to the bit).  What makes a halving
    // butterfly cheap here is the placement: lane l keeps value (s ^ (l & 15)) in slot s, so at every level the value a lane keeps and the
    // value its partner sends are the same slots for every lane — no selects; the placement itself is two conditional swaps per factor
    // vector, made on the float factors before the products are formed.
    float fa[4] = { 0.f, 0.f, 0.f, 0.f }, fb[4] = { 0.f, 0.f, 0.f, 0.f };
    double dsum = 0.0;
    if (i < I.n) {
        const GridIndex g = *I.grid;
        // the first pass reads the source itself (and moves it by the guess), later ones the working copy: no copy up front
        const float4 s = stp->iters == 0 ? I.src[i] : I.cur[i];
        float px, py, pz, d2;
        apply4(stp->Tm, s.x, s.y, s.z, px, py, pz);
        // last iteration's neighbour (position in the sorted target); the first iteration takes a point out of the query's own grid column
        const int seed = stp->iters == 0 ? nn1_cell_seed(px, py, pz, g) : I.nn[i];
        int bi;
        if (Q == 1 && kIcpFlat) bi = nn1_search_flat(px, py, pz, g, cap2, &d2, seed, s_runs);
        else bi = nn1_search<Q>(px, py, pz, g, cap2, &d2, seed);          // (all Q lanes have read their record before the lead lane writes)
        if (lead) { I.cur[i] = make_float4(px, py, pz, s.w); I.nn[i] = bi; }
        if (bi >= 0 && lead) {
            const float4 q = g.pts[bi];
            fa[0] = 1.f; fa[1] = q.x; fa[2] = q.y; fa[3] = q.z;
            fb[0] = 1.f; fb[1] = px;  fb[2] = py;  fb[3] = pz;
            dsum = d2;
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    {
        // slot s <- value s ^ (lane & 15): value index = 4 * (row of fa) + (column of fb)
#define LISREG_CSWAP(c_, x_, y_) do { const float tx_ = (x_), ty_ = (y_); (x_) = (c_) ? ty_ : tx_; (y_) = (c_) ? tx_ : ty_; } while (0)
        const bool a1 = (lane & 4) != 0, a2 = (lane & 8) != 0, b1 = (lane & 1) != 0, b2 = (lane & 2) != 0;
        LISREG_CSWAP(a1, fa[0], fa[1]); LISREG_CSWAP(a1, fa[2], fa[3]); LISREG_CSWAP(a2, fa[0], fa[2]); LISREG_CSWAP(a2, fa[1], fa[3]);
        LISREG_CSWAP(b1, fb[0], fb[1]); LISREG_CSWAP(b1, fb[2], fb[3]); LISREG_CSWAP(b2, fb[0], fb[2]); LISREG_CSWAP(b2, fb[1], fb[3]);
#undef LISREG_CSWAP
    }
    double v16[16];
#pragma unroll
    for (int sl = 0; sl < 16; ++sl) v16[sl] = (double)fa[sl >> 2] * (double)fb[sl & 3];
    double v8[8], v4[4], v2[2];
#pragma unroll
    for (int k = 0; k < 8; ++k) v8[k] = v16[2 * k] + lane_xor_d<1>(v16[2 * k + 1]);
#pragma unroll
    for (int k = 0; k < 4; ++k) v4[k] = v8[2 * k] + lane_xor_d<2>(v8[2 * k + 1]);
#pragma unroll
    for (int k = 0; k < 2; ++k) v2[k] = v4[2 * k] + lane_xor_d<4>(v4[2 * k + 1]);
    double v1 = v2[0] + lane_xor_d<8>(v2[1]);
    v1 += lane_xor_d<16>(v1); v1 += lane_xor_d<32>(v1);              // lane l now holds the wavefront's total of value l & 15
    dsum = wave_sum_up(dsum);
    if (lane < 16) {
        // back to the row layout the summing kernels read: 0 count, 1-3 sum p, 4-6 sum q, 7-15 sum q_r p_c (row-major), 16 sum d2
        const int ra = lane >> 2, cb = lane & 3;
        const int k = ra == 0 ? cb : (cb == 0 ? 3 + ra : 7 + 3 * (ra - 1) + (cb - 1));
        red[wave][k] = v1;
    }
    if (lane == 0) red[wave][16] = dsum;
    __syncthreads();
    if (threadIdx.x < kIcpAcc)
        partials[(size_t)(I.blk0 + blk) * kIcpAcc + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// 3x3 SVD by one-sided Jacobi (double); U S V^T = A, singular values descending, U completed to an orthogonal matrix
__device__ void svd3(const double* Ain, double* U, double* S, double* V)
{
    double A[9];
    for (int i = 0; i < 9; ++i) { A[i] = Ain[i]; V[i] = (i % 4 == 0) ? 1.0 : 0.0; }
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double a = 0, b = 0, c = 0;
                for (int r = 0; r < 3; ++r) { a += A[3 * r + p] * A[3 * r + p]; b += A[3 * r + q] * A[3 * r + q]; c += A[3 * r + p] * A[3 * r + q]; }
                if (fabs(c) <= 1e-300 || fabs(c) <= 1e-15 * sqrt(a * b)) continue;
                off += fabs(c);
                const double zeta = (b - a) / (2.0 * c);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
                for (int r = 0; r < 3; ++r) {
                    double x = A[3 * r + p], y = A[3 * r + q];
                    A[3 * r + p] = cs * x - sn * y; A[3 * r + q] = sn * x + cs * y;
                    x = V[3 * r + p]; y = V[3 * r + q];
                    V[3 * r + p] = cs * x - sn * y; V[3 * r + q] = sn * x + cs * y;
                }
            }
        if (off == 0) break;
    }
    double nrm[3];
    int ord[3] = { 0, 1, 2 };
    for (int j = 0; j < 3; ++j) nrm[j] = sqrt(A[j] * A[j] + A[3 + j] * A[3 + j] + A[6 + j] * A[6 + j]);
    for (int i = 0; i < 2; ++i) for (int j = i + 1; j < 3; ++j) if (nrm[ord[j]] > nrm[ord[i]]) { const int t = ord[i]; ord[i] = ord[j]; ord[j] = t; }
    double Vs[9];
    for (int j = 0; j < 3; ++j) {
        S[j] = nrm[ord[j]];
        for (int r = 0; r < 3; ++r) { Vs[3 * r + j] = V[3 * r + ord[j]]; U[3 * r + j] = S[j] > 0 ? A[3 * r + ord[j]] / S[j] : 0.0; }
    }
    for (int i = 0; i < 9; ++i) V[i] = Vs[i];
    const double tiny = 1e-12 * (S[0] > 0 ? S[0] : 1.0);
    if (S[1] <= tiny) {
        if (S[0] <= 0) { U[0] = 1; U[3] = 0; U[6] = 0; }
        const double u0[3] = { U[0], U[3], U[6] };
        const int k = fabs(u0[0]) < fabs(u0[1]) ? (fabs(u0[0]) < fabs(u0[2]) ? 0 : 2) : (fabs(u0[1]) < fabs(u0[2]) ? 1 : 2);
        double e[3] = { 0, 0, 0 }; e[k] = 1;
        const double d = u0[k];
        double u1[3] = { e[0] - d * u0[0], e[1] - d * u0[1], e[2] - d * u0[2] };
        const double n1 = sqrt(u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2]);
        for (int r = 0; r < 3; ++r) U[3 * r + 1] = u1[r] / n1;
    }
    if (S[2] <= tiny) {
        U[2] = U[3] * U[7] - U[6] * U[4];
        U[5] = U[6] * U[1] - U[0] * U[7];
        U[8] = U[0] * U[4] - U[3] * U[1];
    }
}

__device__ __forceinline__ double det3(const double* M)
{
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}

// estimateRigidTransformation (Umeyama, no scale) + final_transformation_ update + DefaultConvergenceCriteria::hasConverged
__device__ void icp_solve_step(const double* tot, IcpState* st, int max_iters, double eps_t, double eps_mse)
{
    const double cnt = tot[0];
    st->n_corr = (int)cnt;
    if (cnt < 3.0) { st->state = LISREG_ICP_NO_CORRESPONDENCES; st->converged = 0; st->done = 1; return; }    // icp.hpp: min_number_correspondences_
    const double inv = 1.0 / cnt;
    const double ms[3] = { tot[1] * inv, tot[2] * inv, tot[3] * inv }, md[3] = { tot[4] * inv, tot[5] * inv, tot[6] * inv };
    double sg[9];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) sg[3 * r + c] = tot[7 + 3 * r + c] * inv - md[r] * ms[c];
    double U[9], S[3], V[9], R[9];
    svd3(sg, U, S, V);
    const double s2 = det3(U) * det3(V) < 0 ? -1.0 : 1.0;
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c)
        R[3 * r + c] = U[3 * r] * V[3 * c] + U[3 * r + 1] * V[3 * c + 1] + s2 * U[3 * r + 2] * V[3 * c + 2];
    float Tm[16];
    for (int i = 0; i < 16; ++i) Tm[i] = (i % 5 == 0) ? 1.f : 0.f;
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) Tm[4 * r + c] = (float)R[3 * r + c];
        Tm[4 * r + 3] = (float)(md[r] - (R[3 * r] * ms[0] + R[3 * r + 1] * ms[1] + R[3 * r + 2] * ms[2]));
    }
    float Fn[16];
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c)
        Fn[4 * r + c] = ((Tm[4 * r] * st->F[c] + Tm[4 * r + 1] * st->F[4 + c]) + Tm[4 * r + 2] * st->F[8 + c]) + Tm[4 * r + 3] * st->F[12 + c];
    for (int i = 0; i < 16; ++i) { st->F[i] = Fn[i]; st->Tm[i] = Tm[i]; }
    const int iters = ++st->iters;
    st->state = LISREG_ICP_NOT_CONVERGED;
    if (iters >= max_iters) { st->state = LISREG_ICP_ITERATIONS; st->converged = 1; st->done = 1; return; }
    const double cos_angle = 0.5 * ((double)Tm[0] + (double)Tm[5] + (double)Tm[10] - 1.0);
    const double tr2 = (double)Tm[3] * Tm[3] + (double)Tm[7] * Tm[7] + (double)Tm[11] * Tm[11];
    if (cos_angle >= 1.0 - eps_t && tr2 <= eps_t) { st->state = LISREG_ICP_TRANSFORM; st->converged = 1; st->done = 1; return; }
    const double cur = tot[16] * inv, prev = st->prev_mse;
    st->cur_mse = cur;
    if (iters == 1) st->first_mse = cur;                 // the MSE test of the first iteration was reached (chained batches)
    // chained batch (the reference's `static` ICP object, subMapOptmizationNode.cpp:2763): this item's correspondences_prev_mse_ is the
    // value the item before it leaves behind, unknown while both run together — the first comparison is made by the host afterwards
    if (iters == 1 && st->defer_first) { st->prev_mse = cur; return; }
    if (fabs(cur - prev) < 1e-12) { st->state = LISREG_ICP_ABS_MSE; st->converged = 1; st->done = 1; return; }
    if (fabs(cur - prev) / prev < eps_mse) { st->state = LISREG_ICP_REL_MSE; st->converged = 1; st->done = 1; return; }
    st->prev_mse = cur;
}


// estimateRigidTransformation (Umeyama, no scale) + final_transformation_ update + DefaultConvergenceCriteria::hasConverged
__global__ __launch_bounds__(1024) void k_icp_solve(const double* __restrict__ all_partials, const IcpItem* __restrict__ items,
                                                    IcpState* __restrict__ states, int q, int max_iters, double eps_t, double eps_mse,
                                                    int* __restrict__ n_done)
{
    __shared__ double red[32][32];
    __shared__ double tot[kIcpAcc];
    IcpState* st = &states[blockIdx.x];
    if (st->done) return;
    const IcpItem I = items[blockIdx.x];
    const double* partials = all_partials + (size_t)I.blk0 * kIcpAcc;
    const int n_rows = I.nblk, n_blocks = q == 1 ? n_rows : (n_rows + 3) / 4;      // rows of 256 queries
    // fixed-order sum of the 256-query rows: 32 row groups x 17 columns in parallel, 4 independent loads in flight per thread
    // (a scan at 4 lanes per query is ~1 800 rows; summing them 8 ways with one load in flight took 45 us per iteration)
    const int k = threadIdx.x & 31, grp = threadIdx.x >> 5;
    double v0 = 0, v1 = 0, v2 = 0, v3 = 0;
    if (k < kIcpAcc) {
        int b = grp;
        for (; b + 96 < n_blocks; b += 128) {
            v0 += icp_row256<kIcpAcc>(partials, n_rows, q, b, k);      v1 += icp_row256<kIcpAcc>(partials, n_rows, q, b + 32, k);
            v2 += icp_row256<kIcpAcc>(partials, n_rows, q, b + 64, k); v3 += icp_row256<kIcpAcc>(partials, n_rows, q, b + 96, k);
        }
        for (; b < n_blocks; b += 32) v0 += icp_row256<kIcpAcc>(partials, n_rows, q, b, k);
    }
    red[grp][k] = (v0 + v1) + (v2 + v3);
    __syncthreads();
    if (threadIdx.x < kIcpAcc) {
        double s = 0;
        for (int g = 0; g < 32; ++g) s += red[g][threadIdx.x];
        tot[threadIdx.x] = s;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    icp_solve_step(tot, st, max_iters, eps_t, eps_mse);
    if (st->done && n_done) atomicAdd(n_done, 1);
}

// getFitnessScore(): unbounded k = 1 of the source under the final transformation
template <int Q>
__global__ __launch_bounds__(256) void k_icp_fitness(const float4* __restrict__ src, int n, const GridIndex* __restrict__ gp,
                                                     const IcpState* __restrict__ stp, double* __restrict__ partials)
{
    __shared__ double red[4][2];
    const int i = (blockIdx.x * 256 + threadIdx.x) / Q;
    double sum = 0, cnt = 0;
    if (i < n) {
        const GridIndex g = *gp;
        const float4 s = src[i];
        float px, py, pz, d2;
        apply4(stp->F, s.x, s.y, s.z, px, py, pz);
        if (nn1_search<Q>(px, py, pz, g, 3.0e38f, &d2) >= 0 && (threadIdx.x & (Q - 1)) == 0) { sum = d2; cnt = 1; }
    }
    sum = wave_sum(sum); cnt = wave_sum(cnt);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[wave][0] = sum; red[wave][1] = cnt; }
    __syncthreads();
    if (threadIdx.x < 2) partials[(size_t)blockIdx.x * 2 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// the batch forms of the two kernels above: block -> item table, rows of 256 / Q queries summed by the Q-independent tree
template <int Q>
__global__ __launch_bounds__(256) void k_icp_fitness_b(const IcpItem* __restrict__ items, int n_items, const IcpState* __restrict__ states,
                                                       double* __restrict__ partials)
{
    __shared__ double red[4][2];
    __shared__ int2 s_runs[Q == 1 ? kNn1Cap : 1][256];
    int blk;
    int item = icp_locate(items, n_items, &blk);
    item = __builtin_amdgcn_readfirstlane(item); blk = __builtin_amdgcn_readfirstlane(blk);
    if (item < 0) return;
    const IcpItem I = items[item];
    const int i = (int)(((long long)blk * 256 + threadIdx.x) / Q);
    double sum = 0, cnt = 0;
    if (i < I.n) {
        const GridIndex g = *I.grid;
        const float4 s = I.src[i];
        float px, py, pz, d2;
        apply4(states[item].F, s.x, s.y, s.z, px, py, pz);
        // seed: the neighbour of the last ICP iteration (every item has run at least one; -1 where nothing was within reach)
        const int bi = (Q == 1 && kIcpFlat) ? nn1_search_flat(px, py, pz, g, 3.0e38f, &d2, I.nn[i], s_runs) : nn1_search<Q>(px, py, pz, g, 3.0e38f, &d2, I.nn[i]);
        if (bi >= 0 && (threadIdx.x & (Q - 1)) == 0) { sum = d2; cnt = 1; }
    }
    sum = wave_sum_up(sum); cnt = wave_sum_up(cnt);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[wave][0] = sum; red[wave][1] = cnt; }
    __syncthreads();
    if (threadIdx.x < 2) partials[(size_t)(I.blk0 + blk) * 2 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

__global__ __launch_bounds__(256) void k_icp_fit_reduce_b(const double* __restrict__ all_partials, const IcpItem* __restrict__ items,
                                                          IcpState* __restrict__ states, int q)
{
    __shared__ double red[256][2];
    const IcpItem I = items[blockIdx.x];
    const double* partials = all_partials + (size_t)I.blk0 * 2;
    const int n_rows = I.nblk, n_blocks = q == 1 ? n_rows : (n_rows + 3) / 4;
    double s = 0, c = 0;
    for (int b = threadIdx.x; b < n_blocks; b += 256) { s += icp_row256<2>(partials, n_rows, q, b, 0); c += icp_row256<2>(partials, n_rows, q, b, 1); }
    red[threadIdx.x][0] = s; red[threadIdx.x][1] = c;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) { red[threadIdx.x][0] += red[threadIdx.x + o][0]; red[threadIdx.x][1] += red[threadIdx.x + o][1]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { states[blockIdx.x].fit_sum = red[0][0]; states[blockIdx.x].fit_n = (int)red[0][1]; }
}

__global__ __launch_bounds__(256) void k_icp_fit_reduce(const double* __restrict__ partials, int n_blocks, IcpState* __restrict__ st)
{
    __shared__ double red[256][2];
    double s = 0, c = 0;
    for (int b = threadIdx.x; b < n_blocks; b += 256) { s += partials[(size_t)b * 2]; c += partials[(size_t)b * 2 + 1]; }
    red[threadIdx.x][0] = s; red[threadIdx.x][1] = c;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) { red[threadIdx.x][0] += red[threadIdx.x + o][0]; red[threadIdx.x][1] += red[threadIdx.x + o][1]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { st->fit_sum = red[0][0]; st->fit_n = (int)red[0][1]; }
}

// ---- §8 f-4, second half: OptimizedICPGN (src/core/registration.cpp:19-115) ------------------------------------------------
// J = [I | A] with A = -R hat(p), so J^T J = [[n I, sum A], [sum A^T, sum A^T A]] and J^T e = [sum e; sum A^T e]:
// count, sum e (3), sum A (9), sum A^T A (6 unique), sum A^T e (3) = 22 sums instead of 21 + 6 + 1 generic ones.
constexpr int kGnAcc = 22;

template <int Q>
__global__ __launch_bounds__(256) void k_icpgn_assoc(const float4* __restrict__ src, int n, const GridIndex* __restrict__ gp,
                                                     const IcpState* __restrict__ stp, float cap2, double* __restrict__ partials)
{
    __shared__ double red[4][kGnAcc];
    const int i = (blockIdx.x * 256 + threadIdx.x) / Q;
    const bool lead = (threadIdx.x & (Q - 1)) == 0;
    double acc[kGnAcc];
#pragma unroll
    for (int k = 0; k < kGnAcc; ++k) acc[k] = 0.0;
    if (i < n) {
        const GridIndex g = *gp;
        const float4 p = src[i];
        const float* F = stp->F;
        if (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) {                      // pcl::isFinite(origin_point), :37
            float tx, ty, tz, d2;
            apply4(F, p.x, p.y, p.z, tx, ty, tz);
            const int bi = nn1_search<Q>(tx, ty, tz, g, cap2, &d2);
            if (bi >= 0 && lead) {
                const float4 q = g.pts[bi];
                const float e[3] = { tx - q.x, ty - q.y, tz - q.z };
                float A[3][3];
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const float R0 = F[4 * r], R1 = F[4 * r + 1], R2 = F[4 * r + 2];
                    A[r][0] = -(R1 * p.z - R2 * p.y);
                    A[r][1] = -(R2 * p.x - R0 * p.z);
                    A[r][2] = -(R0 * p.y - R1 * p.x);
                }
                acc[0] = 1.0;
                acc[1] = e[0]; acc[2] = e[1]; acc[3] = e[2];
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int c = 0; c < 3; ++c) acc[4 + 3 * r + c] = A[r][c];
                int k = 13;
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int b = a; b < 3; ++b) acc[k++] = (A[0][a] * A[0][b] + A[1][a] * A[1][b]) + A[2][a] * A[2][b];
#pragma unroll
                for (int a = 0; a < 3; ++a) acc[19 + a] = (A[0][a] * e[0] + A[1][a] * e[1]) + A[2][a] * e[2];
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < kGnAcc; ++k) {
        const double v = wave_sum(acc[k]);
        if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < kGnAcc)
        partials[(size_t)blockIdx.x * kGnAcc + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// Hessian.determinant() == 0 -> skip; delta = Hessian.inverse() * B (Eigen PartialPivLU in float); t += delta[0:3];
// R = R * Sophus::SO3f::exp(delta[3:6]).matrix()  (src/sophus/so3.hpp:279-312, small-angle branch as written there)
__global__ __launch_bounds__(1024) void k_icpgn_solve(const double* __restrict__ partials, int n_blocks, IcpState* __restrict__ st)
{
    __shared__ double red[32][32];
    __shared__ double tot[kGnAcc];
    const int k = threadIdx.x & 31, grp = threadIdx.x >> 5;
    double v0 = 0, v1 = 0, v2 = 0, v3 = 0;
    if (k < kGnAcc) {
        int b = grp;
        for (; b + 96 < n_blocks; b += 128) {
            v0 += partials[(size_t)b * kGnAcc + k];        v1 += partials[(size_t)(b + 32) * kGnAcc + k];
            v2 += partials[(size_t)(b + 64) * kGnAcc + k]; v3 += partials[(size_t)(b + 96) * kGnAcc + k];
        }
        for (; b < n_blocks; b += 32) v0 += partials[(size_t)b * kGnAcc + k];
    }
    red[grp][k] = (v0 + v1) + (v2 + v3);
    __syncthreads();
    if (threadIdx.x < kGnAcc) {
        double s = 0;
        for (int g = 0; g < 32; ++g) s += red[g][threadIdx.x];
        tot[threadIdx.x] = s;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    st->n_corr = (int)tot[0];
    float H[36], B[6];
    for (int i = 0; i < 36; ++i) H[i] = 0.f;
    for (int r = 0; r < 3; ++r) {
        H[6 * r + r] = (float)tot[0];
        B[r] = (float)(-tot[1 + r]);
        B[3 + r] = (float)(-tot[19 + r]);
        for (int c = 0; c < 3; ++c) { const float a = (float)tot[4 + 3 * r + c]; H[6 * r + 3 + c] = a; H[6 * (3 + c) + r] = a; }
    }
    { int q = 13; for (int a = 0; a < 3; ++a) for (int b = a; b < 3; ++b) { const float h = (float)tot[q++]; H[6 * (3 + a) + 3 + b] = h; H[6 * (3 + b) + 3 + a] = h; } }
    // PartialPivLU
    float A[36]; int perm[6]; float sign = 1.f;
    for (int i = 0; i < 36; ++i) A[i] = H[i];
    for (int i = 0; i < 6; ++i) perm[i] = i;
    for (int c = 0; c < 6; ++c) {
        int piv = c; float big = fabsf(A[6 * c + c]);
        for (int r = c + 1; r < 6; ++r) if (fabsf(A[6 * r + c]) > big) { big = fabsf(A[6 * r + c]); piv = r; }
        if (piv != c) {
            for (int j = 0; j < 6; ++j) { const float t = A[6 * c + j]; A[6 * c + j] = A[6 * piv + j]; A[6 * piv + j] = t; }
            const int t = perm[c]; perm[c] = perm[piv]; perm[piv] = t; sign = -sign;
        }
        if (A[6 * c + c] != 0.f)
            for (int r = c + 1; r < 6; ++r) {
                A[6 * r + c] /= A[6 * c + c];
                for (int j = c + 1; j < 6; ++j) A[6 * r + j] -= A[6 * r + c] * A[6 * c + j];
            }
    }
    float det = sign;
    for (int c = 0; c < 6; ++c) det *= A[6 * c + c];
    if (det == 0.f) return;                                               // :74-76
    float inv[36];
    for (int col = 0; col < 6; ++col) {
        float y[6];
        for (int r = 0; r < 6; ++r) { float s = (perm[r] == col) ? 1.f : 0.f; for (int c = 0; c < r; ++c) s -= A[6 * r + c] * y[c]; y[r] = s; }
        for (int r = 5; r >= 0; --r) { float s = y[r]; for (int c = r + 1; c < 6; ++c) s -= A[6 * r + c] * inv[6 * c + col]; inv[6 * r + col] = s / A[6 * r + r]; }
    }
    float dx[6];
    for (int r = 0; r < 6; ++r) { float s = 0.f; for (int c = 0; c < 6; ++c) s += inv[6 * r + c] * B[c]; dx[r] = s; }
    float* T = st->F;
    T[3] += dx[0]; T[7] += dx[1]; T[11] += dx[2];
    const float wx = dx[3], wy = dx[4], wz = dx[5];
    const float theta_sq = (wx * wx + wy * wy) + wz * wz;
    const float theta = sqrtf(theta_sq), half = 0.5f * theta;
    float imag, real;
    if (theta < 1e-5f) {
        const float po4 = theta_sq * theta_sq;
        imag = 0.5f - (float)(1.0 / 48.0) * theta_sq + (float)(1.0 / 3840.0) * po4;
        real = 1.f - 0.5f * theta_sq + (float)(1.0 / 384.0) * po4;
    } else { imag = sinf(half) / theta; real = cosf(half); }
    const float qw = real, qx = imag * wx, qy = imag * wy, qz = imag * wz;
    const float tx = 2.f * qx, ty = 2.f * qy, tz = 2.f * qz;
    const float twx = tx * qw, twy = ty * qw, twz = tz * qw, txx = tx * qx, txy = ty * qx, txz = tz * qx, tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
    const float E[9] = { 1.f - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1.f - (txx + tzz), tyz - twx, txz - twy, tyz + twx, 1.f - (txx + tyy) };
    float Rn[9];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Rn[3 * r + c] = (T[4 * r] * E[c] + T[4 * r + 1] * E[3 + c]) + T[4 * r + 2] * E[6 + c];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) T[4 * r + c] = Rn[3 * r + c];
    st->iters += 1;
}

}  // namespace

// lanes per query: a scan-sized query set (~1e5) fills the chip only when each query is spread over several lanes
int nn1_lanes(int n)
{
    if (const char* e = getenv("LISREG_NN1_Q")) { const int q = atoi(e); if (q == 1 || q == 4 || q == 8) return q; }
    return n >= 500000 ? 1 : (n >= 250000 ? 4 : 8);        // ~8 k waves fill the chip (measured at n = 115 k: 1 / 4 / 8 lanes -> 0.127 / 0.054 / 0.049 ms)
}
// ICP keeps to 4: every workgroup also writes a 17-double partial row that k_icp_solve has to sum
int icp_lanes(int n) { return std::min(nn1_lanes(n), 4); }
int nn1_blocks(int n) { return (int)(((long long)n * nn1_lanes(n) + 255) / 256); }
int icp_blocks(int n) { return (int)(((long long)n * icp_lanes(n) + 255) / 256); }

#define LISREG_DISPATCH_Q(q_, call1, call4, call8) do { if ((q_) == 1) { call1; } else if ((q_) == 4) { call4; } else { call8; } } while (0)

// lanes per query of an ICP batch with `total` source points in all (results do not depend on it: wave_sum_up / icp_row256)
int icp_batch_lanes(long long total) { return icp_lanes((int)std::min<long long>(total, 0x7fffffff)); }
int icp_batch_blocks(int n, int q) { return (int)(((long long)n * q + 255) / 256); }

void launch_icp_assoc(const IcpItem* items, int n_items, int total_blocks, int q, IcpState* states, float cap2, double* partials,
                      hipStream_t stream)
{
    if (n_items <= 0 || total_blocks <= 0) return;
    if (q == 1) k_icp_assoc<1><<<total_blocks, 256, 0, stream>>>(items, n_items, states, cap2, partials);
    else        k_icp_assoc<4><<<total_blocks, 256, 0, stream>>>(items, n_items, states, cap2, partials);
}

void launch_icp_solve(const IcpItem* items, int n_items, int q, IcpState* states, const double* partials,
                      int max_iters, double eps_t, double eps_mse, int* n_done, hipStream_t stream)
{
    if (n_items > 0) k_icp_solve<<<n_items, 1024, 0, stream>>>(partials, items, states, q, max_iters, eps_t, eps_mse, n_done);
}

void launch_icp_fitness_batch(const IcpItem* items, int n_items, int total_blocks, int q, IcpState* states, double* partials, hipStream_t stream)
{
    if (n_items <= 0) return;
    if (total_blocks > 0) {
        if (q == 1) k_icp_fitness_b<1><<<total_blocks, 256, 0, stream>>>(items, n_items, states, partials);
        else        k_icp_fitness_b<4><<<total_blocks, 256, 0, stream>>>(items, n_items, states, partials);
    }
    k_icp_fit_reduce_b<<<n_items, 256, 0, stream>>>(partials, items, states, q);
}

void launch_icpgn_iteration(const float4* src, int n, const GridIndex* grid_dev, IcpState* st, float cap2, double* partials,
                            hipStream_t stream)
{
    const int nb = icp_blocks(n), q = icp_lanes(n);
    if (nb > 0)
        LISREG_DISPATCH_Q(q, (k_icpgn_assoc<1><<<nb, 256, 0, stream>>>(src, n, grid_dev, st, cap2, partials)),
                             (k_icpgn_assoc<4><<<nb, 256, 0, stream>>>(src, n, grid_dev, st, cap2, partials)),
                             (k_icpgn_assoc<8><<<nb, 256, 0, stream>>>(src, n, grid_dev, st, cap2, partials)));
    k_icpgn_solve<<<1, 1024, 0, stream>>>(partials, nb, st);
}

void launch_icp_fitness(const float4* src, int n, const GridIndex* grid_dev, IcpState* st, double* partials, hipStream_t stream)
{
    const int nb = icp_blocks(n), q = icp_lanes(n);
    if (nb > 0)
        LISREG_DISPATCH_Q(q, (k_icp_fitness<1><<<nb, 256, 0, stream>>>(src, n, grid_dev, st, partials)),
                             (k_icp_fitness<4><<<nb, 256, 0, stream>>>(src, n, grid_dev, st, partials)),
                             (k_icp_fitness<8><<<nb, 256, 0, stream>>>(src, n, grid_dev, st, partials)));
    k_icp_fit_reduce<<<1, 256, 0, stream>>>(partials, nb, st);
}

void launch_dynamic_flags(const float4* pts, int n, const GridIndex* grid_dev, float center_radius, float near_thre,
                          float dmin, float dmax, int* flag, hipStream_t st)
{
    if (n <= 0) return;
    const float near2 = near_thre * near_thre, dmin2 = dmin * dmin, dmax2 = dmax * dmax;      // float products, +inf for FLT_MAX
    const bool fmax = std::isfinite(dmax2), fmin = std::isfinite(dmin2);
    float cap2 = near2;                                            // largest finite threshold
    if (fmin) cap2 = std::max(cap2, dmin2);
    if (fmax) cap2 = std::max(cap2, dmax2);
    // farther than every finite threshold: d2 > dmax2 if that is finite; else "near2 < d2 < dmin2" holds only for dmin2 = +inf
    const int keep_far = (fmax || !fmin) ? 1 : 0;
    const int q = nn1_lanes(n), nb = nn1_blocks(n);
    const float cr2 = center_radius * center_radius;
    LISREG_DISPATCH_Q(q, (k_dynamic_flags<1><<<nb, 256, 0, st>>>(pts, n, grid_dev, cr2, near2, dmin2, dmax2, cap2, keep_far, flag)),
                         (k_dynamic_flags<4><<<nb, 256, 0, st>>>(pts, n, grid_dev, cr2, near2, dmin2, dmax2, cap2, keep_far, flag)),
                         (k_dynamic_flags<8><<<nb, 256, 0, st>>>(pts, n, grid_dev, cr2, near2, dmin2, dmax2, cap2, keep_far, flag)));
}

void launch_bbx_flags(const float4* pts, int n, const double b[6], int delete_box, int* flag, hipStream_t st)
{
    if (n > 0) k_bbx_flags<<<(n + 255) / 256, 256, 0, st>>>(pts, n, b[0], b[1], b[2], b[3], b[4], b[5], delete_box, flag);
}

void launch_compact(int n, const int* flag, int* pos, int* scan_tmp, int* idx_out, int* count_out, hipStream_t st)
{
    launch_exclusive_scan(flag, pos, scan_tmp, n, st);
    k_compact_write<<<(n + 255) / 256, 256, 0, st>>>(n, flag, pos, idx_out, count_out);
}

void launch_nn1(const float4* q, int n, const GridIndex* grid_dev, float max_dist, int* idx_out, float* d2_out, hipStream_t st)
{
    if (n <= 0) return;
    const int lanes = nn1_lanes(n), nb = nn1_blocks(n);
    const float m2 = max_dist * max_dist;
    LISREG_DISPATCH_Q(lanes, (k_nn1<1><<<nb, 256, 0, st>>>(q, n, grid_dev, m2, idx_out, d2_out)),
                             (k_nn1<4><<<nb, 256, 0, st>>>(q, n, grid_dev, m2, idx_out, d2_out)),
                             (k_nn1<8><<<nb, 256, 0, st>>>(q, n, grid_dev, m2, idx_out, d2_out)));
}

}  // namespace lisreg
