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

namespace lisreg {

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const v4f* gptr_f4;
typedef __attribute__((address_space(1))) const int* gptr_i32;

__device__ __forceinline__ int gcoord(float v, float origin, float inv_cell) { return (int)floorf((v - origin) * inv_cell); }

// nearest target point with squared distance <= max_d2 (index into g.pts, -1 if none); *d2_out = its squared distance
__device__ __forceinline__ int nn1_search(float qx, float qy, float qz, const GridIndex& g, float max_d2, float* d2_out)
{
    constexpr float kEps = 1e-3f;
    const gptr_f4 pts = (gptr_f4)g.pts;
    const gptr_i32 cells = (gptr_i32)g.cell_start;
    float best = __uint_as_float(__float_as_uint(max_d2) + 1u);      // strictly-less test below must admit d2 == max_d2
    int bi = -1;
    if (g.n <= 0) { *d2_out = best; return -1; }
    for (float r = 0.5f;; r *= 2.f) {
        const float lim = fminf(r * r, best);
        const float rad = __builtin_amdgcn_sqrtf(lim) * 1.0001f + kEps;
        const int cx0 = max(gcoord(qx - rad, g.ox, g.inv_cell), 0), cx1 = min(gcoord(qx + rad, g.ox, g.inv_cell), g.nx - 1);
        const int cy0 = max(gcoord(qy - rad, g.oy, g.inv_cell), 0), cy1 = min(gcoord(qy + rad, g.oy, g.inv_cell), g.ny - 1);
        const int cz0 = max(gcoord(qz - rad, g.oz, g.inv_cell), 0), cz1 = min(gcoord(qz + rad, g.oz, g.inv_cell), g.nz - 1);
        if (cz0 <= cz1)
            for (int ix = cx0; ix <= cx1; ++ix) {
                const float xl = g.ox + (float)ix * g.cell;
                const float dx = fmaxf(fmaxf(xl - qx, qx - (xl + g.cell)) - kEps, 0.f);
                if (dx * dx >= fminf(best, lim)) continue;
                for (int iy = cy0; iy <= cy1; ++iy) {
                    const float yl = g.oy + (float)iy * g.cell;
                    const float dy = fmaxf(fmaxf(yl - qy, qy - (yl + g.cell)) - kEps, 0.f);
                    if (dx * dx + dy * dy >= fminf(best, lim)) continue;
                    const int base = (ix * g.ny + iy) * g.nz;
                    const int js = cells[base + cz0], je = cells[base + cz1 + 1];
                    for (int j = js; j < je; ++j) {
                        const v4f c = pts[j];
                        const float ex = qx - c.x, ey = qy - c.y, ez = qz - c.z;
                        const float d2 = ex * ex + ey * ey + ez * ez;        // flann::L2_Simple order
                        if (d2 < best) { best = d2; bi = j; }
                        else if (d2 == best && bi >= 0 && __float_as_int(c.w) < __float_as_int(pts[bi].w)) bi = j;   // ties: smallest original index
                    }
                }
            }
        if (best <= r * r || r * r >= max_d2) break;       // exact: everything within min(best, r) was visited
    }
    *d2_out = best;
    return bi;
}

// map_scan_feature_pts_distance_removal (subMap.h:1076-1087).  The reference's unbounded search is cut at `cap2`, the largest
// FINITE threshold (thresholds default to FLT_MAX, whose square is +inf): beyond it the predicate no longer depends on the
// distance and its value is `keep_far`.
__global__ __launch_bounds__(256) void k_dynamic_flags(const float4* __restrict__ pts, int n, const GridIndex* __restrict__ gp,
                                                       float center_r2, float near2, float dmin2, float dmax2, float cap2,
                                                       int keep_far, int* __restrict__ flag)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 p = pts[i];
    int keep;
    if (p.x * p.x + p.y * p.y > center_r2) keep = 1;
    else {
        float d2;
        const GridIndex g = *gp;
        const int bi = nn1_search(p.x, p.y, p.z, g, cap2, &d2);
        keep = (bi < 0) ? keep_far : (((d2 > near2 && d2 < dmin2) || d2 > dmax2) ? 1 : 0);
    }
    flag[i] = keep;
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
__global__ __launch_bounds__(256) void k_nn1(const float4* __restrict__ q, int n, const GridIndex* __restrict__ gp, float max_d2,
                                             int* __restrict__ idx_out, float* __restrict__ d2_out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const GridIndex g = *gp;
    float d2;
    const int bi = nn1_search(q[i].x, q[i].y, q[i].z, g, max_d2, &d2);
    idx_out[i] = bi < 0 ? -1 : __float_as_int(g.pts[bi].w);      // ORIGINAL index in the caller's map cloud
    d2_out[i] = d2;
}

}  // namespace

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
    k_dynamic_flags<<<(n + 255) / 256, 256, 0, st>>>(pts, n, grid_dev, center_radius * center_radius, near2, dmin2, dmax2, cap2,
                                                    keep_far, flag);
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
    if (n > 0) k_nn1<<<(n + 255) / 256, 256, 0, st>>>(q, n, grid_dev, max_dist * max_dist, idx_out, d2_out);
}

}  // namespace lisreg
