// lisreg_index.hip — device search-index construction for gfx950.
//
// Replaces the two pcl::KdTreeFLANN::setInputCloud calls of every registration
// (/root/reference/src/node/odomEstimationNode.cpp:602-603; subMapOptmizationNode.cpp:1516-1517, 4496-4497) with a
// uniform-grid bucket sort of the target cloud (one or many targets per launch sequence), offers an optional 2-D
// column sort of the source features for callers whose clouds are not spatially coherent, and implements the row
// before the registration: pcl::VoxelGrid and transformPointCloud.
//
// One deterministic bucket sort serves all of them: histogram (integer atomics, order-independent counts) ->
// exclusive scan -> scatter (atomic cursor, arbitrary order inside a bucket) -> rank pass that orders every
// bucket by (sub-key, original index).  The result is bit-reproducible run to run.  All passes are HBM
// streaming/gather passes with coalesced 16-byte records; nothing here is GEMM-shaped.
#include "lisreg_internal.hpp"

namespace lisreg {

namespace {

constexpr int kScanBlock = 256;
constexpr int kScanItems = 8;
constexpr int kScanTile  = kScanBlock * kScanItems;   // 2048 buckets per workgroup

__device__ __forceinline__ int wave_incl_scan(int v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_up(v, d);
        if (lane >= d) v += t;
    }
    return v;
}

// exclusive scan of one value per thread across a 256-thread workgroup; *total = workgroup sum
__device__ __forceinline__ int block_excl_scan(int v, int* total, int* s_wave /* [4] */)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = wave_incl_scan(v, lane);
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        int s = s_wave[w];
        if (w < wave) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

__device__ __forceinline__ void scan_local_body(const int* __restrict__ in, int* __restrict__ out, int* __restrict__ block_sums, int n, int blk_, int* s_wave)
{
    const int base = blk_ * kScanTile + threadIdx.x * kScanItems;
    int v[kScanItems], sum = 0;
    // full tiles of 16-byte aligned arrays move as two int4 per thread (the cell tables of a batch of targets are tens of
    // millions of entries: this pass and k_scan_add are pure streaming)
    const bool vec = (base + kScanItems <= n) && ((((uintptr_t)in | (uintptr_t)out) & 15u) == 0);
    if (vec) {
        const int4 a = *reinterpret_cast<const int4*>(in + base), b = *reinterpret_cast<const int4*>(in + base + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
#pragma unroll
        for (int i = 0; i < kScanItems; ++i) sum += v[i];
    } else {
#pragma unroll
        for (int i = 0; i < kScanItems; ++i) { v[i] = (base + i < n) ? in[base + i] : 0; sum += v[i]; }
    }
    int total;
    int ex = block_excl_scan(sum, &total, s_wave);
    if (vec) {
        int4 a, b;
        a.x = ex; a.y = a.x + v[0]; a.z = a.y + v[1]; a.w = a.z + v[2];
        b.x = a.w + v[3]; b.y = b.x + v[4]; b.z = b.y + v[5]; b.w = b.z + v[6];
        *reinterpret_cast<int4*>(out + base) = a; *reinterpret_cast<int4*>(out + base + 4) = b;
    } else {
#pragma unroll
        for (int i = 0; i < kScanItems; ++i) { if (base + i < n) out[base + i] = ex; ex += v[i]; }
    }
    if (threadIdx.x == 0) block_sums[blk_] = total;
}

__global__ __launch_bounds__(kScanBlock) void k_scan_local(const int* __restrict__ in, int* __restrict__ out,
                                                           int* __restrict__ block_sums, int n)
{
    __shared__ int s_wave[4];
    scan_local_body(in, out, block_sums, n, blockIdx.x, s_wave);
}

// single workgroup: exclusive scan of the per-tile sums in place, grand total appended at [nb]
__global__ __launch_bounds__(kScanBlock) void k_scan_tops(int* __restrict__ block_sums, int nb)
{
    __shared__ int s_wave[4];
    int carry = 0;
    for (int b0 = 0; b0 < nb; b0 += kScanTile) {              // eight sums per thread and round
        const int base = b0 + threadIdx.x * kScanItems;
        int v[kScanItems], sum = 0;
#pragma unroll
        for (int i = 0; i < kScanItems; ++i) { v[i] = (base + i < nb) ? block_sums[base + i] : 0; sum += v[i]; }
        int total;
        int ex = carry + block_excl_scan(sum, &total, s_wave);
#pragma unroll
        for (int i = 0; i < kScanItems; ++i) { if (base + i < nb) block_sums[base + i] = ex; ex += v[i]; }
        carry += total;
    }
    if (threadIdx.x == 0) block_sums[nb] = carry;
}

__global__ __launch_bounds__(kScanBlock) void k_scan_add(int* __restrict__ out, const int* __restrict__ block_sums,
                                                         int n, int nb)
{
    const int off = block_sums[blockIdx.x];
    const int base = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
    if ((base + kScanItems <= n) && (((uintptr_t)out & 15u) == 0)) {
        int4 a = *reinterpret_cast<int4*>(out + base), b = *reinterpret_cast<int4*>(out + base + 4);
        a.x += off; a.y += off; a.z += off; a.w += off; b.x += off; b.y += off; b.z += off; b.w += off;
        *reinterpret_cast<int4*>(out + base) = a; *reinterpret_cast<int4*>(out + base + 4) = b;
    } else {
#pragma unroll
        for (int i = 0; i < kScanItems; ++i) if (base + i < n) out[base + i] += off;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = block_sums[nb];
}

// Round 6: the second and third launch in one for up to 4 096 tiles (8 M entries): every workgroup sums the tile totals in front of it itself
// (at most 16 loads per thread of an L2-resident table, one wave reduction, a four-entry LDS exchange) instead of waiting for a single-workgroup
// launch to scan them — one dependent launch (~5 us of latency) less wherever a scan sits on a critical path: the row counts of the cell rows
// in every configs[1] step, the voxel grids and cell tables of the frame loops.  block_sums keeps the raw totals.
constexpr int kScanFusedMaxTiles = 4096;
__device__ __forceinline__ void scan_add_sum_body(int* __restrict__ out, const int* __restrict__ block_sums, int n, int nb, int blk_, int* s_part)
{
    const int b = blk_, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // (block 0 also owes the grand total: it sums every tile, the others the tiles in front of them)
    const int upto = b == 0 ? nb : b;
    int acc = 0;
    for (int j = threadIdx.x; j < upto; j += kScanBlock) acc += block_sums[j];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
    if (lane == 0) s_part[wave] = acc;
    __syncthreads();
    const int sum = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
    const int off = b == 0 ? 0 : sum;
    const int base = b * kScanTile + threadIdx.x * kScanItems;
    if (off != 0) {
        if ((base + kScanItems <= n) && (((uintptr_t)out & 15u) == 0)) {
            int4 a = *reinterpret_cast<int4*>(out + base), c = *reinterpret_cast<int4*>(out + base + 4);
            a.x += off; a.y += off; a.z += off; a.w += off; c.x += off; c.y += off; c.z += off; c.w += off;
            *reinterpret_cast<int4*>(out + base) = a; *reinterpret_cast<int4*>(out + base + 4) = c;
        } else {
#pragma unroll
            for (int i = 0; i < kScanItems; ++i) if (base + i < n) out[base + i] += off;
        }
    }
    if (b == 0 && threadIdx.x == 0) out[n] = sum;
}

__global__ __launch_bounds__(kScanBlock) void k_scan_add_sum(int* __restrict__ out, const int* __restrict__ block_sums, int n, int nb)
{
    __shared__ int s_part[4];
    scan_add_sum_body(out, block_sums, n, nb, blockIdx.x, s_part);
}

// two independent scans in one launch pair (the row counts of a slot's corner and surf target): workgroups [0, nb0) work on the first
struct ScanPair { const int* in[2]; int* out[2]; int* tmp[2]; int n[2]; int nb[2]; };
__global__ __launch_bounds__(kScanBlock) void k_scan_local_pair(ScanPair p)
{
    __shared__ int s_wave[4];
    const int k = (int)blockIdx.x < p.nb[0] ? 0 : 1;
    scan_local_body(p.in[k], p.out[k], p.tmp[k], p.n[k], (int)blockIdx.x - (k ? p.nb[0] : 0), s_wave);
}
__global__ __launch_bounds__(kScanBlock) void k_scan_add_sum_pair(ScanPair p)
{
    __shared__ int s_part[4];
    const int k = (int)blockIdx.x < p.nb[0] ? 0 : 1;
    scan_add_sum_body(p.out[k], p.tmp[k], p.n[k], p.nb[k], (int)blockIdx.x - (k ? p.nb[0] : 0), s_part);
}

// short arrays (the counts of a single frame: strips, voxels of a down-sampled cloud): ONE workgroup, one launch instead of three — in a
// frame loop the three launches of a 5 k-entry scan are 13 us of a 1 ms frame, a dozen times per frame.  1024 threads x 8 items per
// round, the carry in a register.  in == out is fine (a round reads its items before it writes them).
constexpr int kScanSmallMax = 16384;
__global__ __launch_bounds__(1024) void k_scan_small(const int* __restrict__ in, int* __restrict__ out, int n)
{
    __shared__ int s_wave[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int carry = 0;
    for (int b0 = 0; b0 < n; b0 += 1024 * kScanItems) {
        const int base = b0 + (int)threadIdx.x * kScanItems;
        int v[kScanItems], sum = 0;
#pragma unroll
        for (int i = 0; i < kScanItems; ++i) { v[i] = (base + i < n) ? in[base + i] : 0; sum += v[i]; }
        const int inc = wave_incl_scan(sum, lane);
        __syncthreads();                                     // (the previous round's readers of s_wave are through)
        if (lane == 63) s_wave[wave] = inc;
        __syncthreads();
        int wbase = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) { const int t = s_wave[w]; if (w < wave) wbase += t; tot += t; }
        int ex = carry + wbase + inc - sum;
#pragma unroll
        for (int i = 0; i < kScanItems; ++i) { if (base + i < n) out[base + i] = ex; ex += v[i]; }
        carry += tot;
    }
    if (threadIdx.x == 0) out[n] = carry;
}

void exclusive_scan(const int* in, int* out /* [n+1] */, int* tmp, int n, hipStream_t st)
{
    if (n <= kScanSmallMax) { k_scan_small<<<1, 1024, 0, st>>>(in, out, n); return; }
    const int nb = (n + kScanTile - 1) / kScanTile;
    k_scan_local<<<nb, kScanBlock, 0, st>>>(in, out, tmp, n);
    if (nb <= kScanFusedMaxTiles) { k_scan_add_sum<<<nb, kScanBlock, 0, st>>>(out, tmp, n, nb); return; }
    k_scan_tops<<<1, kScanBlock, 0, st>>>(tmp, nb);
    k_scan_add<<<nb, kScanBlock, 0, st>>>(out, tmp, n, nb);
}

// ---- bounding box ------------------------------------------------------------------------------------------
__device__ __forceinline__ void k_bbox_block(const float4* __restrict__ pts, int n, float* __restrict__ part)
{
    __shared__ float s[4][6];
    float lo[3] = { 3.0e38f, 3.0e38f, 3.0e38f }, hi[3] = { -3.0e38f, -3.0e38f, -3.0e38f };
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        float4 p = pts[i];
        lo[0] = fminf(lo[0], p.x); lo[1] = fminf(lo[1], p.y); lo[2] = fminf(lo[2], p.z);
        hi[0] = fmaxf(hi[0], p.x); hi[1] = fmaxf(hi[1], p.y); hi[2] = fmaxf(hi[2], p.z);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], d));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], d));
        }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) for (int k = 0; k < 3; ++k) { s[wave][k] = lo[k]; s[wave][3 + k] = hi[k]; }
    __syncthreads();
    if (threadIdx.x < 6) {
        float v = s[0][threadIdx.x];
        for (int w = 1; w < 4; ++w) v = threadIdx.x < 3 ? fminf(v, s[w][threadIdx.x]) : fmaxf(v, s[w][threadIdx.x]);
        part[blockIdx.x * 6 + threadIdx.x] = v;
    }
}

__global__ __launch_bounds__(256) void k_bbox_partial(const float4* __restrict__ pts, int n, float* __restrict__ part) { k_bbox_block(pts, n, part); }

__device__ __forceinline__ void k_bbox_fold(const float* __restrict__ part, int nb, float* __restrict__ bbox6)
{
    // one wave: every lane folds the partial rows lane, lane + 64, ... , then a butterfly (the serial 6-thread version of
    // this kernel took 25 us — a quarter of a map-index build)
    float lo[3] = { 3.0e38f, 3.0e38f, 3.0e38f }, hi[3] = { -3.0e38f, -3.0e38f, -3.0e38f };
    for (int b = threadIdx.x; b < nb; b += 64)
#pragma unroll
        for (int k = 0; k < 3; ++k) { lo[k] = fminf(lo[k], part[b * 6 + k]); hi[k] = fmaxf(hi[k], part[b * 6 + 3 + k]); }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1)
#pragma unroll
        for (int k = 0; k < 3; ++k) { lo[k] = fminf(lo[k], __shfl_xor(lo[k], d)); hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], d)); }
    if (threadIdx.x == 0)
#pragma unroll
        for (int k = 0; k < 3; ++k) { bbox6[k] = lo[k]; bbox6[3 + k] = hi[k]; }
}
__global__ void k_bbox_final(const float* __restrict__ part, int nb, float* __restrict__ bbox6) { k_bbox_fold(part, nb, bbox6); }

// ---- target keys -------------------------------------------------------------------------------------------
__device__ __forceinline__ int cell_coord(float v, float origin, float inv_cell, int n)
{
    int c = (int)floorf((v - origin) * inv_cell);
    return c < 0 ? 0 : (c >= n ? n - 1 : c);
}

__global__ __launch_bounds__(256) void k_target_keys(const float4* __restrict__ pts, int n, GridIndex g,
                                                     uint32_t* __restrict__ elem_bucket,
                                                     uint32_t* __restrict__ elem_sub, int* __restrict__ hist)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 p = pts[i];
    const int ix = cell_coord(p.x, g.ox, g.inv_cell, g.nx);
    const int iy = cell_coord(p.y, g.oy, g.inv_cell, g.ny);
    const int iz = cell_coord(p.z, g.oz, g.inv_cell, g.nz);
    const uint32_t b = (uint32_t)((ix * g.ny + iy) * g.nz + iz);
    // targets have no sub-key: the slot carries the point's arrival rank in its cell instead, which spares the scatter pass
    // its own atomic (the order inside a cell is fixed afterwards by the rank pass, so which rank a point draws is irrelevant)
    elem_sub[i] = (uint32_t)atomicAdd(&hist[b], 1);
}

// single-target versions of the point-carrying scatter / gather-free rank (see k_tseg_scatter_pts below)
__global__ __launch_bounds__(256) void k_target_scatter_pts(const float4* __restrict__ pts, int n, GridIndex g,
                                                            const uint32_t* __restrict__ elem_rank,
                                                            const int* __restrict__ cell_start, float4* __restrict__ tmp_pts)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float4 p = pts[i];
    const int b = (cell_coord(p.x, g.ox, g.inv_cell, g.nx) * g.ny + cell_coord(p.y, g.oy, g.inv_cell, g.ny)) * g.nz + cell_coord(p.z, g.oz, g.inv_cell, g.nz);
    p.w = __int_as_float(i);
    tmp_pts[cell_start[b] + (int)elem_rank[i]] = p;
}

__global__ __launch_bounds__(256) void k_target_rank_pts(int n, GridIndex g, const float4* __restrict__ tmp_pts,
                                                         const int* __restrict__ cell_start, float4* __restrict__ sorted_out)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const float4 v = tmp_pts[p];
    const int b = (cell_coord(v.x, g.ox, g.inv_cell, g.nx) * g.ny + cell_coord(v.y, g.oy, g.inv_cell, g.ny)) * g.nz + cell_coord(v.z, g.oz, g.inv_cell, g.nz);
    const int s = cell_start[b], e = cell_start[b + 1], idx = __float_as_int(v.w);
    int rank = 0;
    for (int j = s; j < e; ++j) rank += (__float_as_int(tmp_pts[j].w) < idx) ? 1 : 0;
    sorted_out[s + rank] = v;
}

// ---- source keys: (x,y) sort column of the point under the item's INITIAL pose (ItemState::M, written by the
// reset kernel = pcl::getTransformation of T_init, common.cpp:54-57) ---------------------------------------------
__global__ __launch_bounds__(kBlockQ) void k_source_keys(const BlockDesc* __restrict__ blocks,
                                                         const Segment* __restrict__ segs,
                                                         const ItemState* __restrict__ items,
                                                         uint32_t* __restrict__ elem_bucket,
                                                         uint32_t* __restrict__ elem_sub, int* __restrict__ hist)
{
    const BlockDesc bd = blocks[blockIdx.x];
    if ((int)threadIdx.x >= bd.count) return;
    const Segment sg = segs[bd.seg];
    const float* M = items[bd.item].M;      // uniform: scalar loads
    const int e = bd.start + threadIdx.x;
    const float4 p = sg.src[e];
    const float x = M[0] * p.x + M[1] * p.y + M[2] * p.z + M[3];
    const float y = M[4] * p.x + M[5] * p.y + M[6] * p.z + M[7];
    const float z = M[8] * p.x + M[9] * p.y + M[10] * p.z + M[11];
    // sort key = (x,y) column of the point under the initial pose; sub-key = height bin.  Keys only order the
    // queries (wave-level locality for the cell walk); clamping is harmless.
    int tx = (int)floorf((x - sg.tox) * sg.inv_tile), ty = (int)floorf((y - sg.toy) * sg.inv_tile);
    int tz = (int)floorf((z - sg.toz) * sg.inv_tile);
    tx = tx < 0 ? 0 : (tx >= sg.tnx ? sg.tnx - 1 : tx);
    ty = ty < 0 ? 0 : (ty >= sg.tny ? sg.tny - 1 : ty);
    tz = tz < 0 ? 0 : (tz > 4095 ? 4095 : tz);
    const uint32_t b = (uint32_t)(sg.bucket_base + tx * sg.tny + ty);
    const int flat = sg.flat_base + e;
    elem_bucket[flat] = b;
    elem_sub[flat] = (uint32_t)tz;
    atomicAdd(&hist[b], 1);
}

// ---- scatter + rank ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_scatter(const uint32_t* __restrict__ elem_bucket,
                                                 const uint32_t* __restrict__ elem_sub, int n,
                                                 const int* __restrict__ bucket_start, int* __restrict__ hist,
                                                 uint32_t* __restrict__ tmp_bucket, uint32_t* __restrict__ tmp_sub,
                                                 int* __restrict__ tmp_idx)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t b = elem_bucket[i];
    const int pos = bucket_start[b] + atomicSub(&hist[b], 1) - 1;
    tmp_bucket[pos] = b;
    tmp_sub[pos] = elem_sub[i];
    tmp_idx[pos] = i;
}

// final position of scattered slot p inside its bucket: rank by (sub-key, original index).  kOneKey: every element of the bucket has the
// same sub-key (a voxel grid whose buckets are single voxels — span 1: the key-frame ring's 1 M points, ~25 per voxel), so the original
// index alone decides and the loop reads one array instead of two
template <bool kOneKey = false>
__device__ __forceinline__ int rank_in_bucket(int p, const uint32_t* __restrict__ tmp_bucket,
                                              const uint32_t* __restrict__ tmp_sub, const int* __restrict__ tmp_idx,
                                              const int* __restrict__ bucket_start, int* idx_out)
{
    const uint32_t b = tmp_bucket[p];
    const int s = bucket_start[b], e = bucket_start[b + 1];
    const uint32_t sub = kOneKey ? 0u : tmp_sub[p];
    const int idx = tmp_idx[p];
    int rank = 0;
#pragma unroll 8
    for (int j = s; j < e; ++j) {
        const int ij = tmp_idx[j];
        if (kOneKey) rank += ij < idx ? 1 : 0;
        else { const uint32_t sj = tmp_sub[j]; rank += (sj < sub || (sj == sub && ij < idx)) ? 1 : 0; }
    }
    *idx_out = idx;
    return s + rank;
}

__global__ __launch_bounds__(256) void k_rank_source(const Segment* __restrict__ segs, int n_segs, int n,
                                                     const uint32_t* __restrict__ tmp_bucket,
                                                     const uint32_t* __restrict__ tmp_sub,
                                                     const int* __restrict__ tmp_idx,
                                                     const int* __restrict__ bucket_start,
                                                     float4* __restrict__ sorted_all, int* __restrict__ order_all)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    int flat;
    const int dst = rank_in_bucket(p, tmp_bucket, tmp_sub, tmp_idx, bucket_start, &flat);
    int lo = 0, hi = n_segs - 1;           // last segment with flat_base <= flat
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (segs[mid].flat_base <= flat) lo = mid; else hi = mid - 1;
    }
    const int e = flat - segs[lo].flat_base;
    sorted_all[dst] = segs[lo].src[e];
    order_all[dst] = e;
}

// ---- batched target builds: many grids, one bucket sort ------------------------------------------------------------
__global__ __launch_bounds__(kBlockQ) void k_tseg_keys(const BlockDesc* __restrict__ blocks,
                                                       const TargetSeg* __restrict__ tsegs,
                                                       uint32_t* __restrict__ elem_bucket,
                                                       uint32_t* __restrict__ elem_sub, int* __restrict__ hist)
{
    const BlockDesc bd = blocks[blockIdx.x];
    if ((int)threadIdx.x >= bd.count) return;
    const TargetSeg t = tsegs[bd.seg];
    const int e = bd.start + threadIdx.x;
    const float4 p = t.raw[e];
    const int ix = cell_coord(p.x, t.ox, t.inv_cell, t.nx);
    const int iy = cell_coord(p.y, t.oy, t.inv_cell, t.ny);
    const int iz = cell_coord(p.z, t.oz, t.inv_cell, t.nz);
    const uint32_t b = (uint32_t)(t.bucket_base + (ix * t.ny + iy) * t.nz + iz);
    elem_sub[t.flat_base + e] = (uint32_t)atomicAdd(&hist[b], 1);       // arrival rank in the cell (see k_target_keys)
}

__device__ __forceinline__ int find_tseg_by_flat(const TargetSeg* __restrict__ tsegs, int n, int flat)
{
    int lo = 0, hi = n - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (tsegs[mid].flat_base <= flat) lo = mid; else hi = mid - 1; }
    return lo;
}

// Batched target build, second half.  The scatter carries the point itself (x, y, z, original index) to the slot its arrival
// rank assigns — one 16-byte write instead of three scattered 4-byte writes — and the rank pass re-derives the cell from the
// coordinates, orders the cell's members by original index (deterministic result) and writes the final record: no gather.
__device__ __forceinline__ uint32_t tseg_bucket(const TargetSeg& t, float x, float y, float z)
{
    const int ix = cell_coord(x, t.ox, t.inv_cell, t.nx), iy = cell_coord(y, t.oy, t.inv_cell, t.ny), iz = cell_coord(z, t.oz, t.inv_cell, t.nz);
    return (uint32_t)(t.bucket_base + (ix * t.ny + iy) * t.nz + iz);
}

__global__ __launch_bounds__(kBlockQ) void k_tseg_scatter_pts(const BlockDesc* __restrict__ blocks, const TargetSeg* __restrict__ tsegs,
                                                              const uint32_t* __restrict__ elem_rank,
                                                              const int* __restrict__ bucket_start, float4* __restrict__ tmp_pts)
{
    const BlockDesc bd = blocks[blockIdx.x];
    if ((int)threadIdx.x >= bd.count) return;
    const TargetSeg t = tsegs[bd.seg];
    const int e = bd.start + threadIdx.x;
    float4 p = t.raw[e];
    const int pos = bucket_start[tseg_bucket(t, p.x, p.y, p.z)] + (int)elem_rank[t.flat_base + e];
    p.w = __int_as_float(e);
    tmp_pts[pos] = p;
}

__global__ __launch_bounds__(256) void k_tseg_rank_pts(const TargetSeg* __restrict__ tsegs, int n_tsegs, int n,
                                                       const float4* __restrict__ tmp_pts, const int* __restrict__ bucket_start)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const float4 v = tmp_pts[p];
    const TargetSeg t = tsegs[find_tseg_by_flat(tsegs, n_tsegs, p)];      // slots of a target = its flat range
    const uint32_t b = tseg_bucket(t, v.x, v.y, v.z);
    const int s = bucket_start[b], e = bucket_start[b + 1], idx = __float_as_int(v.w);
    int rank = 0;
    for (int j = s; j < e; ++j) rank += (__float_as_int(tmp_pts[j].w) < idx) ? 1 : 0;
    t.sorted_out[s + rank - t.flat_base] = v;
}

// per-target cell_start = slice of the global bucket scan, rebased to the target's own sorted array
__global__ __launch_bounds__(256) void k_tseg_cell_starts(const TargetSeg* __restrict__ tsegs, int n_tsegs,
                                                          const int* __restrict__ bucket_start)
{
    const TargetSeg t = tsegs[blockIdx.y];
    for (int c = blockIdx.x * 256 + threadIdx.x; c <= t.n_cells; c += gridDim.x * 256)
        t.cell_start_out[c] = (c < t.n_cells ? bucket_start[t.bucket_base + c] : t.flat_base + t.n) - t.flat_base;
}


// ---- batched target builds, strip form ----------------------------------------------------------------------------------
// The bucket sort above pays one global atomic per point on a random cell: 12.8 M of them take 0.51-0.55 ms on an MI355X
// whatever the table size (0.4 MB or 172 MB, tests/probes/atomic_probe.hip) — the L2 atomic units serve ~24 G scattered
// requests/s — while the same count on consecutive addresses takes 0.036 ms and a 4096-bin LDS histogram per 4096 points
// 0.09 ms.  So the points are first partitioned into STRIPS — the cells of one ix and a run of `ystrip` consecutive iy, i.e. a
// contiguous range of cell ids ((ix * ny + iy) * nz + iz) — with an LDS histogram per 4096-point chunk and one global atomic
// per chunk and non-empty strip; then ONE workgroup per (target, strip) finishes its strip entirely in LDS: cell histogram,
// scan, placement, order inside the cells by original index, the strip's stretch of cell_start, the sorted records.  Strip
// offsets + in-strip offsets ARE the global cell starts: no pass over the cell table of the whole batch (43 M cells for
// 64 x 2 targets), no memset of it, no scan beyond the strip counts.  Same output as the bucket sort, bit for bit.
constexpr int kPartChunk   = kPartChunkHost;   // points per workgroup of the partition passes
constexpr int kPartThreads = 1024;

__device__ __forceinline__ int strip_of(const TargetSeg& t, float x, float y)
{
    return cell_coord(x, t.ox, t.inv_cell, t.nx) * t.nstrips + cell_coord(y, t.oy, t.inv_cell, t.ny) / t.ystrip;
}

// kScatter = false: strip populations (cnt);  true: move the records (x, y, z, original index) to their strip's stretch of tmp_pts
template <int NT> __device__ __forceinline__ int block_excl_scan_nt(int v, int* s_wave);

// kOwnScan (scatter pass, round 6; batches whose strips number at most kMaxStrips): every workgroup scans the strip populations itself —
// `cnt` is complete at the kernel boundary, 8 K entries are eight per thread and one workgroup scan — and workgroup 0 writes `start_out` for
// the strip builds: the single-workgroup scan launch between the two partition passes (5-7 us of a configs[1] step's serial head) is gone.
template <bool kScatter, bool kOwnScan = false>
__global__ __launch_bounds__(kPartThreads) void k_strip_partition(const BlockDesc* __restrict__ chunks, const TargetSeg* __restrict__ tsegs,
                                                                  int* __restrict__ cnt, const int* __restrict__ start,
                                                                  int* __restrict__ fill, float4* __restrict__ tmp_pts,
                                                                  int* __restrict__ start_out = nullptr, int n_strips_all = 0)
{
    __shared__ int s_hist[kMaxStrips];
    __shared__ int s_start[kOwnScan ? kMaxStrips + 1 : 1];
    __shared__ int s_wv[kPartThreads / 64];
    const BlockDesc bd = chunks[blockIdx.x];
    const TargetSeg t = tsegs[bd.seg];
    const int tid = threadIdx.x, n_units = t.nx * t.nstrips;
    if (kScatter && kOwnScan) {
        constexpr int L = (kMaxStrips + kPartThreads - 1) / kPartThreads;
        int v[L], sum = 0;
#pragma unroll
        for (int i = 0; i < L; ++i) { const int e = tid * L + i; v[i] = e < n_strips_all ? cnt[e] : 0; sum += v[i]; }
        int run = block_excl_scan_nt<kPartThreads>(sum, s_wv);
#pragma unroll
        for (int i = 0; i < L; ++i) { const int e = tid * L + i; if (e <= n_strips_all) s_start[e] = run; run += v[i]; }
        __syncthreads();
        if (blockIdx.x == 0) for (int e = tid; e <= n_strips_all; e += kPartThreads) start_out[e] = s_start[e];
    }
    for (int k = tid; k < n_units; k += kPartThreads) s_hist[k] = 0;
    __syncthreads();
    float4 p[kPartChunk / kPartThreads];
    int u[kPartChunk / kPartThreads], r[kPartChunk / kPartThreads];
#pragma unroll
    for (int k = 0; k < kPartChunk / kPartThreads; ++k) {
        const int e = k * kPartThreads + tid;
        u[k] = -1; r[k] = 0;
        if (e < bd.count) {
            p[k] = t.raw[bd.start + e];
            u[k] = strip_of(t, p[k].x, p[k].y);
            if (kScatter) r[k] = atomicAdd(&s_hist[u[k]], 1); else atomicAdd(&s_hist[u[k]], 1);
        }
    }
    __syncthreads();
    for (int k = tid; k < n_units; k += kPartThreads) {
        const int c = s_hist[k];
        if (c) {
            if (kScatter) s_hist[k] = (kOwnScan ? s_start[t.strip_base + k] : start[t.strip_base + k]) + atomicAdd(&fill[t.strip_base + k], c);
            else atomicAdd(&cnt[t.strip_base + k], c);
        }
    }
    if (kScatter) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kPartChunk / kPartThreads; ++k)
            if (u[k] >= 0) {
                float4 v = p[k];
                v.w = __int_as_float(bd.start + k * kPartThreads + tid);
                tmp_pts[s_hist[u[k]] + r[k]] = v;
            }
    }
}

// slot tables of one strip: in LDS (16-bit positions: a strip in LDS holds < 65536 points), or — a strip too big for LDS — in
// global scratch, read back past the L1 (the workgroup reads what other wavefronts of it wrote)
struct SlotsLds {
    uint32_t* idx; uint16_t* pos;
    __device__ __forceinline__ void put(int s, uint32_t i, int p) const { idx[s] = i; pos[s] = (uint16_t)p; }
    __device__ __forceinline__ uint32_t get_idx(int s) const { return idx[s]; }
    __device__ __forceinline__ int get_pos(int s) const { return pos[s]; }
};
struct SlotsGlobal {
    uint32_t* idx; uint32_t* pos;
    __device__ __forceinline__ void put(int s, uint32_t i, int p) const { idx[s] = i; pos[s] = (uint32_t)p; }
    __device__ __forceinline__ uint32_t get_idx(int s) const { return __hip_atomic_load(&idx[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    __device__ __forceinline__ int get_pos(int s) const { return (int)__hip_atomic_load(&pos[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
};

// exclusive scan of one value per thread across an NT-thread workgroup
template <int NT>
__device__ __forceinline__ int block_excl_scan_nt(int v, int* s_wave /* [NT / 64] */)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int inc = wave_incl_scan(v, lane);
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) if (w < wave) base += s_wave[w];
    __syncthreads();
    return base + inc - v;
}

template <int NT, typename Slots>
__device__ __forceinline__ void strip_build_body(const TargetSeg& t, int y0, int cell0, int ucells, int s0, int n_s,
                                                 int* hist /* LDS [ucells + 1] */, const Slots sl, const float4* __restrict__ tmp, int* s_wave)
{
    const int tid = threadIdx.x;
    auto ucell = [&](const float4& v) { return (cell_coord(v.y, t.oy, t.inv_cell, t.ny) - y0) * t.nz + cell_coord(v.z, t.oz, t.inv_cell, t.nz); };
    for (int c = tid; c <= ucells; c += NT) hist[c] = 0;
    __syncthreads();
    for (int p = tid; p < n_s; p += NT) atomicAdd(&hist[ucell(tmp[s0 + p])], 1);
    __syncthreads();
    {   // in-place INCLUSIVE scan (cell ends): every thread owns a run of cells
        const int L = (ucells + NT - 1) / NT, b = tid * L;
        int sum = 0;
        for (int i = 0; i < L; ++i) if (b + i < ucells) sum += hist[b + i];
        int run = block_excl_scan_nt<NT>(sum, s_wave);
        for (int i = 0; i < L; ++i) if (b + i < ucells) { run += hist[b + i]; hist[b + i] = run; }
        if (tid == 0) hist[ucells] = n_s;
    }
    __syncthreads();
    // placement from the cell ends downwards (arrival order inside a cell is arbitrary; the last pass orders it): afterwards
    // hist[c] is the cell's START
    for (int p = tid; p < n_s; p += NT) {
        const float4 v = tmp[s0 + p];
        const int slot = atomicSub(&hist[ucell(v)], 1) - 1;
        sl.put(slot, (uint32_t)__float_as_int(v.w), p);
    }
    __syncthreads();
    const int rel = s0 - t.flat_base;                       // the strip's first record in the target's own sorted array
    {
        int* cs = t.cell_start_out + cell0;
        for (int c = tid; c < ucells; c += NT) cs[c] = rel + hist[c];
        if (cell0 + ucells == t.n_cells && tid == 0) t.cell_start_out[t.n_cells] = t.n;
    }
    for (int s = tid; s < n_s; s += NT) {
        const float4 v = tmp[s0 + sl.get_pos(s)];
        const int c = ucell(v);
        const int a = hist[c], e = hist[c + 1];
        const uint32_t idx = (uint32_t)__float_as_int(v.w);
        int rank = 0;
        for (int j = a; j < e; ++j) rank += sl.get_idx(j) < idx ? 1 : 0;
        t.sorted_out[rel + a + rank] = v;
    }
}

// kLarge = false: strips of at most cap_small points, 256 threads, several workgroups per CU;  true: the rest, 1024 threads and as
// much LDS as a workgroup can have — and global slot tables for a strip that still does not fit
template <bool kLarge>
__global__ __launch_bounds__(kLarge ? 1024 : 256) void k_strip_build(const TargetSeg* __restrict__ tsegs, const int* __restrict__ start,
                                                                     const float4* __restrict__ tmp_pts, int cap_small, int cap_large,
                                                                     uint32_t* __restrict__ g_slot_idx, uint32_t* __restrict__ g_slot_pos,
                                                                     int* __restrict__ zero_buf = nullptr, int zero_n = 0)
{
    constexpr int NT = kLarge ? 1024 : 256;
    extern __shared__ int s_dyn[];
    __shared__ int s_wave[NT / 64];
    // (round 6) the strip populations and scatter cursors go back clean for the next build of this batch: no memset launch in front of it
    if (!kLarge && zero_buf && blockIdx.x == 0 && blockIdx.y == 0) for (int i = threadIdx.x; i < zero_n; i += NT) zero_buf[i] = 0;
    const TargetSeg t = tsegs[blockIdx.y];
    const int unit = blockIdx.x;
    if (t.n <= 0) { if (!kLarge && unit == 0 && threadIdx.x == 0) t.cell_start_out[0] = 0; return; }
    if (unit >= t.nx * t.nstrips) return;
    const int s0 = start[t.strip_base + unit], n_s = start[t.strip_base + unit + 1] - s0;
    if (kLarge ? (n_s <= cap_small) : (n_s > cap_small)) return;
    const int ix = unit / t.nstrips, ys = unit - ix * t.nstrips;
    const int y0 = ys * t.ystrip, y1 = min(t.ny, y0 + t.ystrip);
    const int cell0 = (ix * t.ny + y0) * t.nz, ucells = (y1 - y0) * t.nz;
    const int ucap = t.ystrip * t.nz;                       // table size the launch was dimensioned for
    const int cap = kLarge ? cap_large : cap_small;
    if (!kLarge || n_s <= cap) {
        SlotsLds sl;
        sl.idx = (uint32_t*)(s_dyn + ucap + 1);
        sl.pos = (uint16_t*)(sl.idx + cap);
        strip_build_body<NT>(t, y0, cell0, ucells, s0, n_s, s_dyn, sl, tmp_pts, s_wave);
    } else {
        SlotsGlobal sl;
        sl.idx = g_slot_idx + s0; sl.pos = g_slot_pos + s0;
        strip_build_body<NT>(t, y0, cell0, ucells, s0, n_s, s_dyn, sl, tmp_pts, s_wave);
    }
}

// ---- k-NN graph over a finished target index (search_mode 3) -----------------------------------------------------------
// For every sorted point s: up to kGraphK nearest OTHER points (sorted positions, ascending by distance) and a coverage
// radius rho(s) with the guarantee  |x - s| < rho(s)  =>  x is in the list  (and every listed point is within rho).
// The correspondence kernel turns that into an exact 5-NN certificate by the triangle inequality (lisreg_assoc.hip,
// LISREG_GRAPH_SCAN): with an anchor a at distance d_a from the query and c5 the 5th-best distance found in
// {a} + list(a), every point outside the list is at least rho(a) - d_a away.
//
// One WAVE per point, lanes = candidates: the candidates are the points of the 5 x 5 x 5 cell block around the point's
// cell (25 contiguous z-runs), 64 at a time; each chunk is bitonic-sorted across the wave by (d^2, id) and merged
// into the running 32 best.  rho = min(distance of the 32nd, distance to the nearest face of the block that has
// cells beyond it), so nothing outside the block has to be looked at: sparse neighbourhoods simply get shorter
// lists with rho = the block's inscribed radius (>= 2 cells).  No divergence, one coalesced 128-byte store per point.
// One wavefront per workgroup: the rows are wavefront-level work (no barrier), and a workgroup's slots come free only when its LAST
// wavefront ends — with four wavefronts of unequal work per workgroup the build ran at 2.4 wavefronts per SIMD in flight (round 5,
// profiles/r05_kernel_experiments.md section 19: graph build 947 -> 906 us per launch on configs[4] with 8 points per wavefront, row build below)
#ifndef LISREG_GRAPH_WPB
#define LISREG_GRAPH_WPB 1           // wavefronts per workgroup of the graph build
#endif
#ifndef LISREG_GRAPH_PPW
#define LISREG_GRAPH_PPW 8
#endif
constexpr int kGraphPPW = LISREG_GRAPH_PPW;            // points per wave (sequential)

// value of lane (l ^ M).  The sort below is bound by cross-lane traffic, and ds_bpermute (the LDS crossbar, 4 LDS cycles per
// wave-instruction) was 40 % of the build's wave time; most masks of the network have a pure-VALU form on gfx950: quad
// permutes (1, 2, 3), row mirrors (7, 15), a row rotation (8) — DPP moves.  The masks 4, 16, 31, 32, 63 stay on the
// LDS crossbar: moving them to VALU forms too (row_ror pairs, V_PERMLANE16/32_SWAP) measured slower (0.29 vs 0.256 ms per 200 k
// points) — the two pipes are balanced as it is.
template <int M> __device__ __forceinline__ int lane_xor(int v)
{
    if constexpr (M == 1)       return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true);     // quad_perm [1,0,3,2]
    else if constexpr (M == 2)  return __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true);     // quad_perm [2,3,0,1]
    else if constexpr (M == 3)  return __builtin_amdgcn_mov_dpp(v, 0x1B, 0xF, 0xF, true);     // quad_perm [3,2,1,0]
    else if constexpr (M == 7)  return __builtin_amdgcn_mov_dpp(v, 0x141, 0xF, 0xF, true);    // row_half_mirror
    else if constexpr (M == 15) return __builtin_amdgcn_mov_dpp(v, 0x140, 0xF, 0xF, true);    // row_mirror
    else if constexpr (M == 8)  return __builtin_amdgcn_mov_dpp(v, 0x128, 0xF, 0xF, true);    // row_ror:8
    else return __shfl_xor(v, M);
}

// compare-exchange of 64-bit keys (squared distance bits << 32 | id: positive floats order like unsigned integers, the id breaks
// ties) with the lane M away; lanes whose bit `LowBit` is clear keep the smaller key.  Every stage of the network below has
// this one direction, so a stage is two cross-lane moves, one 64-bit compare and two selects.
template <int M, int LowBit> __device__ __forceinline__ unsigned long long cmpx64(unsigned long long k, int lane)
{
    const unsigned lo = (unsigned)lane_xor<M>((int)(unsigned)k), hi = (unsigned)lane_xor<M>((int)(unsigned)(k >> 32));
    const unsigned long long pk = ((unsigned long long)hi << 32) | lo;
    const bool take_min = (lane & LowBit) == 0;
    return ((pk < k) == take_min) ? pk : k;
}

template <int H> __device__ __forceinline__ unsigned long long half_cleaners(unsigned long long k, int lane)
{
    if constexpr (H > 0) { k = cmpx64<H, H>(k, lane); return half_cleaners<(H >> 1)>(k, lane); }
    else return k;
}
// merge sorted blocks of H into sorted blocks of 2H, in the "mirror" form: the first stage pairs lane i with the lane mirrored
// inside the block, the rest are half-cleaners — the lower lane always keeps the minimum
template <int H> __device__ __forceinline__ unsigned long long merge_blocks(unsigned long long k, int lane)
{
    k = cmpx64<2 * H - 1, H>(k, lane);
    return half_cleaners<(H >> 1)>(k, lane);
}
// ascending sort of one key per lane across the wave
__attribute__((unused)) __device__ __forceinline__ unsigned long long sort64(unsigned long long k, int lane)
{
    k = merge_blocks<1>(k, lane); k = merge_blocks<2>(k, lane); k = merge_blocks<4>(k, lane);
    k = merge_blocks<8>(k, lane); k = merge_blocks<16>(k, lane); k = merge_blocks<32>(k, lane);
    return k;
}

// One row: the kGraphK target points nearest to the location q (home cell hx, hy, hz; `s` = a sorted point to leave out, or -1), ascending,
// with the coverage radius of the 5 x 5 x 5 block — the k-NN graph's rows are anchored at the points themselves (graph_build_wave), the
// cell rows of search_mode 5 at cell and octant centres (k_crow_build).
template <int R>                          // block = (2 R + 1)^3 cells
__device__ __forceinline__ void row_build_wave(const GridIndex& g, const float4 q, int hx, int hy, int hz, int s,
                                               float4* __restrict__ row_out, float2* __restrict__ meta_out, int (*s_off)[64], int (*s_js)[64])
{
    constexpr int W = 2 * R + 1, NR = W * W;
    static_assert(NR <= 64, "one lane per z-run");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr float kEps = 1e-3f;
    const int x0 = max(hx - R, 0), x1 = min(hx + R, g.nx - 1), y0 = max(hy - R, 0), y1 = min(hy + R, g.ny - 1);
    const int z0 = max(hz - R, 0), z1 = min(hz + R, g.nz - 1);
    // inscribed radius: faces of the block that coincide with the grid boundary have nothing beyond them
    float rc = 3.0e18f;
    if (x0 > 0)        rc = fminf(rc, q.x - (g.ox + (float)x0 * g.cell));
    if (x1 < g.nx - 1) rc = fminf(rc, (g.ox + (float)(x1 + 1) * g.cell) - q.x);
    if (y0 > 0)        rc = fminf(rc, q.y - (g.oy + (float)y0 * g.cell));
    if (y1 < g.ny - 1) rc = fminf(rc, (g.oy + (float)(y1 + 1) * g.cell) - q.y);
    if (z0 > 0)        rc = fminf(rc, q.z - (g.oz + (float)z0 * g.cell));
    if (z1 < g.nz - 1) rc = fminf(rc, (g.oz + (float)(z1 + 1) * g.cell) - q.z);
    rc = fmaxf(rc - kEps, 0.f);
    // the 25 z-runs of the block, their exclusive prefix
    int js = 0, len = 0;
    if (lane < NR) {
        const int ix = hx + lane / W - R, iy = hy + lane % W - R;
        if (ix >= x0 && ix <= x1 && iy >= y0 && iy <= y1) {
            const int base = (ix * g.ny + iy) * g.nz;
            js = g.cell_start[base + z0];
            len = g.cell_start[base + z1 + 1] - js;
        }
    }
    int inc = len;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(inc, d); if (lane >= d) inc += t; }
    const int total = __shfl(inc, 63);
    __builtin_amdgcn_wave_barrier();
    s_off[wave][lane] = inc - len; s_js[wave][lane] = js;
    __builtin_amdgcn_wave_barrier();

    constexpr unsigned long long kEmpty = ((unsigned long long)0x7f800000u << 32) | 0xffffffffull;      // (+inf, id -1)
    unsigned long long top = kEmpty;                        // running kGraphK best in lanes 0..kGraphK-1, ascending
#pragma unroll 1
    for (int c0 = 0; c0 < total; c0 += 64) {
        const int t = c0 + lane;
        unsigned long long k = kEmpty;
        if (t < total) {
            int lo = 0, hi = NR - 1;                        // last run whose offset is <= t
#pragma unroll
            for (int it = 0; it < (NR > 32 ? 6 : 5); ++it) { const int mid = (lo + hi + 1) >> 1; if (s_off[wave][mid] <= t) lo = mid; else hi = mid - 1; }
            const int j = s_js[wave][lo] + (t - s_off[wave][lo]);
            const float4 c = g.pts[j];
            const float ex = q.x - c.x, ey = q.y - c.y, ez = q.z - c.z;
            const float d2 = ex * ex + ey * ey + ez * ez;
            if (j != s && d2 < 3.0e38f) k = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)j;   // NaN / Inf points are never neighbours
        }
        k = sort64(k, lane);
        if constexpr (kGraphK == 32) {
            // lanes 32..63 <- the chunk's 32 smallest, reversed; lanes 0..31 keep the running best: one bitonic sequence, merged
            // by the half-cleaners alone
            const unsigned rlo = (unsigned)__shfl((int)(unsigned)k, 63 - lane), rhi = (unsigned)__shfl((int)(unsigned)(k >> 32), 63 - lane);
            unsigned long long m = lane < 32 ? top : (((unsigned long long)rhi << 32) | rlo);
            top = half_cleaners<32>(m, lane);
        } else {
            // running 64 best in all lanes: the lane-wise minimum of the running list and the reversed chunk is the 64 smallest of
            // both as one bitonic sequence; the half-cleaners sort it
            const unsigned rlo = (unsigned)__shfl((int)(unsigned)k, 63 - lane), rhi = (unsigned)__shfl((int)(unsigned)(k >> 32), 63 - lane);
            const unsigned long long r = ((unsigned long long)rhi << 32) | rlo;
            top = half_cleaners<32>(r < top ? r : top, lane);
        }
    }
    const float tk = __uint_as_float((unsigned)(top >> 32));
    const int ti = (int)(unsigned)top;
    const float dK = __shfl(tk, kGraphK - 1);
    const float rho2 = fminf(rc * rc, dK);
    const bool keep = lane < kGraphK && ti >= 0 && tk <= rho2;
    const int cnt = __popcll(__ballot(keep));
    if (lane < kGraphK) {
        // the row carries the neighbour's coordinates next to its id (one coalesced 1-KB store per point): the correspondence
        // kernel then scans a row without an id -> point gather.  Entries that are not kept alias the point itself.
        float4 e = make_float4(q.x, q.y, q.z, __int_as_float(-1));
        if (keep) { const float4 c = g.pts[ti]; e = make_float4(c.x, c.y, c.z, __int_as_float(ti)); }
        row_out[lane] = e;
    }
    if (lane == 0) *meta_out = make_float2(rho2, __int_as_float(cnt));
}

// Round 5: the graph's rows through the sort of the cell rows (32-bit keys: the squared distance's float bits with the low 7 bits replaced by
// the lane, one min / max per network stage instead of a 64-bit compare and four selects; defined with the cell rows below).  The order of a
// row is then exact to 2^-16 relative — far inside the millimetre of slack the scan's stop test carries for exactly this (kEps in
// LISREG_GRAPH_GROUP) — and rho comes from the quantised key of the first point left out, a floor: "closer than rho => listed" holds as before.
// -DLISREG_GRAPH_KEYS64=1 keeps the 64-bit (distance, id) keys of rounds 2-4.
#ifndef LISREG_GRAPH_KEYS64
#define LISREG_GRAPH_KEYS64 0
#endif
__attribute__((unused)) __device__ __forceinline__ void graph_row_q32(const GridIndex& g, const float4 q, int hx, int hy, int hz, int s,
                                              float4* __restrict__ row_out, float2* __restrict__ meta_out, int (*s_off)[64], int (*s_js)[64]);

__device__ __forceinline__ void graph_build_wave(const GridIndex& g, int s, int (*s_off)[64], int (*s_js)[64])
{
    const float4 q = g.pts[s];
    const int hx = cell_coord(q.x, g.ox, g.inv_cell, g.nx), hy = cell_coord(q.y, g.oy, g.inv_cell, g.ny), hz = cell_coord(q.z, g.oz, g.inv_cell, g.nz);
#if LISREG_GRAPH_KEYS64
    row_build_wave<2>(g, q, hx, hy, hz, s, const_cast<float4*>(g.nbr) + (size_t)s * kGraphK, const_cast<float2*>(g.nbr_meta) + s, s_off, s_js);
#else
    graph_row_q32(g, q, hx, hy, hz, s, const_cast<float4*>(g.nbr) + (size_t)s * kGraphK, const_cast<float2*>(g.nbr_meta) + s, s_off, s_js);
#endif
}

__global__ __launch_bounds__(64 * LISREG_GRAPH_WPB) void k_graph_build_one(GridIndex g)
{
    __shared__ int s_off[LISREG_GRAPH_WPB][64], s_js[LISREG_GRAPH_WPB][64];
    const int first = (blockIdx.x * LISREG_GRAPH_WPB + (threadIdx.x >> 6)) * kGraphPPW;
#pragma unroll 1
    for (int i = 0; i < kGraphPPW; ++i) {
        const int s = first + i;
        if (s >= g.n) break;
        graph_build_wave(g, s, s_off, s_js);
    }
}

__global__ __launch_bounds__(64 * LISREG_GRAPH_WPB) void k_graph_build_batched(const BlockDesc* __restrict__ blocks,
                                                             const TargetSeg* __restrict__ tsegs,
                                                             const GridIndex* __restrict__ grids)
{
    __shared__ int s_off[LISREG_GRAPH_WPB][64], s_js[LISREG_GRAPH_WPB][64];
    constexpr int kSub = kBlockQ / (LISREG_GRAPH_WPB * kGraphPPW);         // workgroups per 256-point block descriptor
    static_assert(kSub * LISREG_GRAPH_WPB * kGraphPPW == kBlockQ, "whole workgroups per block descriptor");
    const BlockDesc bd = blocks[blockIdx.x / kSub];
    const int first = ((int)(blockIdx.x % kSub) * LISREG_GRAPH_WPB + (int)(threadIdx.x >> 6)) * kGraphPPW;
    const GridIndex g = grids[tsegs[bd.seg].grid_id];
#pragma unroll 1
    for (int i = 0; i < kGraphPPW; ++i) {
        const int e = first + i;
        if (e >= bd.count) break;
        graph_build_wave(g, bd.start + e, s_off, s_js);
    }
}

// ---- cell rows (search_mode 5) -------------------------------------------------------------------------------------------
// Which cells get rows.  A cell with no target point in the 5 x 5 x 5 block around it gets none (need 0: a query in it has nothing within
// two cells); every other cell a row at its centre; and each of its eight octants that has a target point within `oct_margin` of its box
// (per axis) its own row behind that (omask bits 0..7: the octants the surface runs through or next to — where the queries are once the
// pose has settled, at most 0.43 of a half cell from the octant's centre).  need = 1 + the number of such octants; omask >> 8 = the
// population of the cell's 5 x 5 x 5 block (capped).
// k_crow_mark: one thread per target point ORs the octants it is near into the cells' masks (at most 2 x 2 x 2 octants: the margin is under
// half an octant's edge) — a point-driven pass costs one or two atomics per point, where a cell-driven pass read every point of a 3 x 3 x 3
// block per cell (7 ms of vector work per million cells).
// The points arrive sorted by cell: neighbouring lanes mostly name the same cell with the same octants, and 3 000 wavefronts of device-scope
// atomics in flight at once are what this launch waits for (a wavefront of the 10 k-point corner target lives 3.7 us, one of the 200 k-point
// surf target 21 us).  A lane whose left neighbour (same 16-lane row) names the same cell with at least its bits leaves the atomic to it —
// the left-most lane of such a run always sends, and its mask covers the run (each lane's mask is inside its sender's by induction).
__device__ __forceinline__ void crow_mark_body(const GridIndex& g, float oct_margin, int* __restrict__ omask, int blk_)
{
    const int i = blk_ * 256 + threadIdx.x, lane = threadIdx.x & 63;
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < g.n) p = g.pts[i];
    const bool valid = i < g.n && p.x == p.x && p.y == p.y && p.z == p.z;
    const float inv_h = 2.f * g.inv_cell;
    // octant coordinates (half cells) of p -/+ the margin, clamped to the grid
    const int ax0 = min(max((int)floorf((p.x - oct_margin - g.ox) * inv_h), 0), 2 * g.nx - 1), ax1 = min(max((int)floorf((p.x + oct_margin - g.ox) * inv_h), 0), 2 * g.nx - 1);
    const int ay0 = min(max((int)floorf((p.y - oct_margin - g.oy) * inv_h), 0), 2 * g.ny - 1), ay1 = min(max((int)floorf((p.y + oct_margin - g.oy) * inv_h), 0), 2 * g.ny - 1);
    const int az0 = min(max((int)floorf((p.z - oct_margin - g.oz) * inv_h), 0), 2 * g.nz - 1), az1 = min(max((int)floorf((p.z + oct_margin - g.oz) * inv_h), 0), 2 * g.nz - 1);
    // at most 2 x 2 x 2 cells (the margin is under half an octant's edge): slot s = the cell at offset (s & 1, s >> 1 & 1, s >> 2) from the first
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int cx = (ax0 >> 1) + (s & 1), cy = (ay0 >> 1) + ((s >> 1) & 1), cz = (az0 >> 1) + (s >> 2);
        const bool in = valid && cx <= (ax1 >> 1) && cy <= (ay1 >> 1) && cz <= (az1 >> 1);
        const int bx = (ax0 <= 2 * cx && 2 * cx <= ax1 ? 1 : 0) | (ax0 <= 2 * cx + 1 && 2 * cx + 1 <= ax1 ? 2 : 0);        // lower / upper half in range
        const int by = (ay0 <= 2 * cy && 2 * cy <= ay1 ? 1 : 0) | (ay0 <= 2 * cy + 1 && 2 * cy + 1 <= ay1 ? 2 : 0);
        const int bz = (az0 <= 2 * cz && 2 * cz <= az1 ? 1 : 0) | (az0 <= 2 * cz + 1 && 2 * cz + 1 <= az1 ? 2 : 0);
        // octant o = (x half) + 2 (y half) + 4 (z half): the mask is the outer product of the three two-bit ranges
        const int mxy = ((by & 1) ? bx : 0) | ((by & 2) ? bx << 2 : 0);
        int m = ((bz & 1) ? mxy : 0) | ((bz & 2) ? mxy << 4 : 0);
        const int key = in ? (cx * g.ny + cy) * g.nz + cz : -1 - lane;          // (lanes without this slot: keys of their own)
        if (!in) m = 0;
        // left neighbour in the 16-lane row (row_shr:1; the row's first lane reads itself through bound_ctrl = 0 -> its own value: it never matches
        // because its "neighbour mask" is then taken as 0)
        const int pk = __builtin_amdgcn_update_dpp(-2 - lane, key, 0x111, 0xF, 0xF, false);
        const int pm = __builtin_amdgcn_update_dpp(0, m, 0x111, 0xF, 0xF, false);
        if (in && !(pk == key && (m & ~pm) == 0)) atomicOr(&omask[key], m);
    }
}

__global__ __launch_bounds__(256) void k_crow_mark(GridIndex g, float oct_margin, int* __restrict__ omask) { crow_mark_body(g, oct_margin, omask, blockIdx.x); }

// The same marks into an LDS tile (round 6, the LDS-tiled classification): the octants point p is near, ORed into the masks of those of its
// (at most 2 x 2 x 2) cells that lie inside the tile [tx0, tx0 + kCtX) x [ty0, ty0 + kCtY) x [0, nz) — LDS atomics, no dedupe needed.
// Same arithmetic as crow_mark_point: the same bits in the same cells.
__device__ __forceinline__ void crow_mark_point_tile(const GridIndex& g, const float4 p, float oct_margin, int tx0, int ty0, int ctx, int cty, int* s_mask)
{
    if (!(p.x == p.x && p.y == p.y && p.z == p.z)) return;
    const float inv_h = 2.f * g.inv_cell;
    const int ax0 = min(max((int)floorf((p.x - oct_margin - g.ox) * inv_h), 0), 2 * g.nx - 1), ax1 = min(max((int)floorf((p.x + oct_margin - g.ox) * inv_h), 0), 2 * g.nx - 1);
    const int ay0 = min(max((int)floorf((p.y - oct_margin - g.oy) * inv_h), 0), 2 * g.ny - 1), ay1 = min(max((int)floorf((p.y + oct_margin - g.oy) * inv_h), 0), 2 * g.ny - 1);
    const int az0 = min(max((int)floorf((p.z - oct_margin - g.oz) * inv_h), 0), 2 * g.nz - 1), az1 = min(max((int)floorf((p.z + oct_margin - g.oz) * inv_h), 0), 2 * g.nz - 1);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int cx = (ax0 >> 1) + (s & 1), cy = (ay0 >> 1) + ((s >> 1) & 1), cz = (az0 >> 1) + (s >> 2);
        const bool in = cx <= (ax1 >> 1) && cy <= (ay1 >> 1) && cz <= (az1 >> 1);
        const int bx = (ax0 <= 2 * cx && 2 * cx <= ax1 ? 1 : 0) | (ax0 <= 2 * cx + 1 && 2 * cx + 1 <= ax1 ? 2 : 0);
        const int by = (ay0 <= 2 * cy && 2 * cy <= ay1 ? 1 : 0) | (ay0 <= 2 * cy + 1 && 2 * cy + 1 <= ay1 ? 2 : 0);
        const int bz = (az0 <= 2 * cz && 2 * cz <= az1 ? 1 : 0) | (az0 <= 2 * cz + 1 && 2 * cz + 1 <= az1 ? 2 : 0);
        const int mxy = ((by & 1) ? bx : 0) | ((by & 2) ? bx << 2 : 0);
        const int m = ((bz & 1) ? mxy : 0) | ((bz & 2) ? mxy << 4 : 0);
        const int lx = cx - tx0, ly = cy - ty0;
        if (in && m != 0 && lx >= 0 && lx < ctx && ly >= 0 && ly < cty) atomicOr(&s_mask[(lx * cty + ly) * g.nz + cz], m);
    }
}

// k_crow_classify: one workgroup per tile of kCtX x kCtY columns over the whole z-range: the cell_start rows of the tile and its
// two-column rim are staged in LDS once (two global reads per cell instead of fifty), every thread sums its cells' 5 x 5 columns from there.
#ifndef LISREG_CT_X
#define LISREG_CT_X 4            // tile of 4 x 8 columns (8 x 8 until round 5: 441 workgroups for a 200 k-point target left most of the chip idle; 17.8 -> 14.6-16.5 us)
#endif
#ifndef LISREG_CT_Y
#define LISREG_CT_Y 8
#endif
constexpr int kCtX = LISREG_CT_X, kCtY = LISREG_CT_Y, kCtRim = 2;
// (both classification kernels) reach != null: a populated cell that no query of the batch comes within a metre of gets no rows this
// run — need 0, and bit 30 of its mask tells the build to write -1 ("no row: walk") instead of -2 ("nothing within two cells") into its table entry
constexpr int kCrowUnreached = 1 << 30;
__device__ __forceinline__ bool crow_reached(const GridIndex& g, const unsigned* __restrict__ reach, int ix, int iy, int iz)
{
    return !reach || ((reach[(size_t)(ix * g.ny + iy) * g.qmark_w + (iz >> 5)] >> (iz & 31)) & 1u) != 0u;
}

// Round 6: the octant masks of the tile's cells are made HERE, in LDS (s_mask, behind the staged cell_start rows): the points of the tile's
// columns and of a one-column rim around it (a point marks cells at most one away; columns adjacent in y are adjacent in memory, so that is
// one contiguous run of points per x) each OR their octants into the tile's masks with LDS atomics.  Until then the marks were a launch of
// their own (k_crow_mark: one or two device-scope atomics per point into a global mask array, 26 us per configs[1] step on the stream's
// serial head) whose result this kernel read back; a tile reads ~1.3 x its own points instead.  Same masks (crow_mark_point_tile).
__device__ __forceinline__ void crow_classify_body(const GridIndex& g, int tiles_y, int* __restrict__ need, int* __restrict__ omask,
                                                   const unsigned* __restrict__ reach, int blk_, int* s_cs, float oct_margin)
{
    //                            // [(kCtX + 2 rim) * (kCtY + 2 rim)][nz + 1] cell_start rows, then [kCtX * kCtY][nz] masks
    constexpr int WX = kCtX + 2 * kCtRim, WY = kCtY + 2 * kCtRim;
    const int nz1 = g.nz + 1;
    const int tx0 = (int)(blk_ / tiles_y) * kCtX, ty0 = (int)(blk_ % tiles_y) * kCtY;
    int* s_mask = s_cs + WX * WY * nz1;
    for (int i = threadIdx.x; i < WX * WY * nz1; i += 256) {
        const int c = i / nz1, z = i - c * nz1;
        const int x = tx0 - kCtRim + c / WY, y = ty0 - kCtRim + c % WY;
        // a column outside the grid holds nothing: any constant row will do
        s_cs[i] = (x >= 0 && x < g.nx && y >= 0 && y < g.ny) ? g.cell_start[(x * g.ny + y) * g.nz + z] : 0;
    }
    for (int i = threadIdx.x; i < kCtX * kCtY * g.nz; i += 256) s_mask[i] = 0;
    __syncthreads();
    {
        const int ylo = max(ty0 - 1, 0), yhi = min(ty0 + kCtY, g.ny - 1);
        if (ylo <= yhi)
            for (int x = max(tx0 - 1, 0); x <= min(tx0 + kCtX, g.nx - 1); ++x) {
                const int c0 = (x - (tx0 - kCtRim)) * WY + (ylo - (ty0 - kCtRim)), c1 = (x - (tx0 - kCtRim)) * WY + (yhi - (ty0 - kCtRim));
                const int p0 = s_cs[c0 * nz1], p1 = s_cs[c1 * nz1 + g.nz];
                for (int p = p0 + (int)threadIdx.x; p < p1; p += 256) crow_mark_point_tile(g, g.pts[p], oct_margin, tx0, ty0, kCtX, kCtY, s_mask);
            }
    }
    __syncthreads();
    const int ncell = kCtX * kCtY * g.nz;
    for (int i = threadIdx.x; i < ncell; i += 256) {
        const int col = i / g.nz, iz = i - col * g.nz;
        const int lx = col / kCtY, ly = col % kCtY;
        const int ix = tx0 + lx, iy = ty0 + ly;
        if (ix >= g.nx || iy >= g.ny) continue;
        const int z0 = max(iz - 2, 0), z1 = min(iz + 2, g.nz - 1);
        int cnt5 = 0;
#pragma unroll
        for (int dx = -2; dx <= 2; ++dx)
#pragma unroll
            for (int dy = -2; dy <= 2; ++dy) {
                const int* row = s_cs + ((lx + kCtRim + dx) * WY + (ly + kCtRim + dy)) * nz1;
                cnt5 += row[z1 + 1] - row[z0];
            }
        const int cid = (ix * g.ny + iy) * g.nz + iz;
        const int m = s_mask[i] & 255;
        const bool r = cnt5 != 0 && crow_reached(g, reach, ix, iy, iz);
        need[cid] = r ? 1 + __popc(m) : 0;
        omask[cid] = cnt5 ? (r ? (m | (min(cnt5, 0xffff) << 8)) : kCrowUnreached) : 0;
    }
}

__global__ __launch_bounds__(256) void k_crow_classify(GridIndex g, int tiles_y, int* __restrict__ need, int* __restrict__ omask,
                                                       const unsigned* __restrict__ reach, float oct_margin)
{
    extern __shared__ int s_cs[];
    crow_classify_body(g, tiles_y, need, omask, reach, blockIdx.x, s_cs, oct_margin);
}

// the same per cell, straight from memory: grids whose tile does not fit the LDS (z-ranges beyond ~120 cells)
__device__ __forceinline__ void crow_classify_plain_body(const GridIndex& g, int n_cells, int* __restrict__ need, int* __restrict__ omask,
                                                         const unsigned* __restrict__ reach, int blk_)
{
    const int cid = blk_ * 256 + threadIdx.x;
    if (cid >= n_cells) return;
    const int iz = cid % g.nz, t = cid / g.nz, iy = t % g.ny, ix = t / g.ny;
    const int z0 = max(iz - 2, 0), z1 = min(iz + 2, g.nz - 1);
    int cnt5 = 0;
#pragma unroll 1
    for (int dx = -2; dx <= 2; ++dx) {
        const int x = ix + dx;
        if (x < 0 || x >= g.nx) continue;
#pragma unroll
        for (int dy = -2; dy <= 2; ++dy) {
            const int y = iy + dy;
            if (y < 0 || y >= g.ny) continue;
            const int base = (x * g.ny + y) * g.nz;
            cnt5 += g.cell_start[base + z1 + 1] - g.cell_start[base + z0];
        }
    }
    const int m = omask[cid] & 255;
    const bool r = cnt5 != 0 && crow_reached(g, reach, ix, iy, iz);
    need[cid] = r ? 1 + __popc(m) : 0;
    omask[cid] = cnt5 ? (r ? (m | (min(cnt5, 0xffff) << 8)) : kCrowUnreached) : 0;
}

__global__ __launch_bounds__(256) void k_crow_classify_plain(GridIndex g, int n_cells, int* __restrict__ need, int* __restrict__ omask,
                                                             const unsigned* __restrict__ reach)
{
    crow_classify_plain_body(g, n_cells, need, omask, reach, blockIdx.x);
}

// The sorts of the cell-row build work on 32-bit keys: the squared distance's float bits with the low 7 bits replaced by a payload (the
// lane the candidate sits in, and which of two lists it came from) — one min / max per network stage instead of a 64-bit compare and
// four selects, the ids fetched once per sort through the payload.  Distances are thereby ordered up to 2^-16 relative, far inside the
// millimetre the scan's stop test allows for (kEps in LISREG_GRAPH_GROUP), and the coverage radius is taken from the QUANTISED key of the
// first point left out (a floor: every point left out is at least that far), so "closer than rho => listed" holds exactly.
template <int M, int LowBit> __device__ __forceinline__ unsigned cmpx32(unsigned k, int lane)
{
    const unsigned pk = (unsigned)lane_xor<M>((int)k);
    const unsigned lo = min(k, pk), hi = max(k, pk);
    return (lane & LowBit) == 0 ? lo : hi;
}
template <int H> __device__ __forceinline__ unsigned half_cleaners32(unsigned k, int lane)
{
    if constexpr (H > 0) { k = cmpx32<H, H>(k, lane); return half_cleaners32<(H >> 1)>(k, lane); }
    else return k;
}
template <int H> __device__ __forceinline__ unsigned merge_blocks32(unsigned k, int lane)
{
    k = cmpx32<2 * H - 1, H>(k, lane);
    return half_cleaners32<(H >> 1)>(k, lane);
}
__device__ __forceinline__ unsigned sort64x32(unsigned k, int lane)
{
    k = merge_blocks32<1>(k, lane); k = merge_blocks32<2>(k, lane); k = merge_blocks32<4>(k, lane);
    k = merge_blocks32<8>(k, lane); k = merge_blocks32<16>(k, lane); k = merge_blocks32<32>(k, lane);
    return k;
}

// The rows of one cell (one wave).  crow_runs: the z-runs of the (2 R + 1)^3 block around the cell (shared by all its rows);
// crow_list: the kGraphK block points nearest to a location q, ascending (quantised key + id per lane); crow_emit: one row from such a list.
template <int R>
struct CrowBlock { int x0, x1, y0, y1, z0, z1, total; };

template <int R>
__device__ __forceinline__ CrowBlock<R> crow_runs(const GridIndex& g, int hx, int hy, int hz, int (*s_off)[64], int (*s_js)[64])
{
    constexpr int W = 2 * R + 1, NR = W * W;
    static_assert(NR <= 64, "one lane per z-run");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    CrowBlock<R> b;
    b.x0 = max(hx - R, 0); b.x1 = min(hx + R, g.nx - 1); b.y0 = max(hy - R, 0); b.y1 = min(hy + R, g.ny - 1);
    b.z0 = max(hz - R, 0); b.z1 = min(hz + R, g.nz - 1);
    int js = 0, len = 0;
    if (lane < NR) {
        const int ix = hx + lane / W - R, iy = hy + lane % W - R;
        if (ix >= b.x0 && ix <= b.x1 && iy >= b.y0 && iy <= b.y1) {
            const int base = (ix * g.ny + iy) * g.nz;
            js = g.cell_start[base + b.z0];
            len = g.cell_start[base + b.z1 + 1] - js;
        }
    }
    int inc = len;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(inc, d); if (lane >= d) inc += t; }
    b.total = __shfl(inc, 63);
    __builtin_amdgcn_wave_barrier();
    s_off[wave][lane] = inc - len; s_js[wave][lane] = js;
    __builtin_amdgcn_wave_barrier();
    return b;
}

// distance from q to the nearest face of the block that has cells beyond it (faces on the grid boundary have nothing beyond them)
template <int R>
__device__ __forceinline__ float crow_inscribed(const GridIndex& g, const CrowBlock<R>& b, float qx, float qy, float qz)
{
    float rc = 3.0e18f;
    if (b.x0 > 0)        rc = fminf(rc, qx - (g.ox + (float)b.x0 * g.cell));
    if (b.x1 < g.nx - 1) rc = fminf(rc, (g.ox + (float)(b.x1 + 1) * g.cell) - qx);
    if (b.y0 > 0)        rc = fminf(rc, qy - (g.oy + (float)b.y0 * g.cell));
    if (b.y1 < g.ny - 1) rc = fminf(rc, (g.oy + (float)(b.y1 + 1) * g.cell) - qy);
    if (b.z0 > 0)        rc = fminf(rc, qz - (g.oz + (float)b.z0 * g.cell));
    if (b.z1 < g.nz - 1) rc = fminf(rc, (g.oz + (float)(b.z1 + 1) * g.cell) - qz);
    return fmaxf(rc - 1e-3f, 0.f);
}

constexpr unsigned kInfQ = 0x7f800000u;                     // +inf, payload bits clear

// topc (x, y, z of the listed point, per lane) is filled where it costs no memory access: a block of at most 64 candidates is ONE chunk,
// every candidate's record is in some lane's registers, and three lane permutes put it next to its key — the row is then written without
// gathering the points a second time (have_c; a multi-chunk list leaves it to crow_emit's gather).
// Round 6 (kFilter; the k-NN graph's rows): a block of more than 64 candidates is FILTERED before it is sorted.  A row lists nothing beyond the block's inscribed radius rc
// (crow_emit: rho^2 = min(rc^2, 64th distance), an entry is kept iff its key <= rho^2), and the sphere of that radius holds pi / 4 of a
// surface's points inside the block's square: the ~75 candidates of a floor cell of BASELINE configs[1] become ~59 — ONE 64-key sort
// instead of two sorts and a merge.  The survivors of every 64-candidate round are compacted (ballot + prefix count) into a 128-entry
// queue in LDS; whenever it holds 64 they are sorted and merged as a chunk was before.  The predicate is crow_emit's own (the truncated
// key against rc^2), so the rows are the same entry for entry.
// Measured (profiles/r06_kernel_experiments.md): the graph build of configs[4] (1 M points, 0.25 m cells: 3-6 rounds per point) -2.6 % of a step;
// the cell rows of configs[1] (two rounds per surface cell) LOSE 1.3 % of a step with it — the queue's LDS round trips and a sixth wavefront
// per SIMD that no longer fits the registers cost more than the second sort they save — so the row build keeps the unfiltered loop.
constexpr int kListQueueWaves = 4;                          // wavefronts per workgroup the queue is dimensioned for (both builds run 1)
template <int R, bool kFilter = false>
__device__ __forceinline__ void crow_list(const GridIndex& g, const CrowBlock<R>& b, float qx, float qy, float qz,
                                          int (*s_off)[64], int (*s_js)[64], unsigned& topk, int& topi, float4& topc, bool& have_c,
                                          int exclude = -1 /* a sorted position to leave out: the point itself, for the k-NN graph's rows */,
                                          float rc2 = 3.0e38f /* nothing farther than this (squared) can be listed: crow_emit's rc * rc */)
{
    constexpr int W = 2 * R + 1, NR = W * W;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ unsigned s_qk[kListQueueWaves][128];
    __shared__ int s_qj[kListQueueWaves][128];
    topk = kInfQ; topi = -1;                                // running kGraphK best, ascending: quantised key and id per lane
    topc = make_float4(0.f, 0.f, 0.f, 0.f);
    have_c = b.total <= 64;
    // one 64-candidate round of the block: the lane's candidate (position j, record cc) and its truncated key, kInfQ if it has none
    auto candidate = [&](int t, int& cj, float4& cc) -> unsigned {
        unsigned ck = kInfQ;
        cj = -1; cc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t < b.total) {
            int lo = 0, hi = NR - 1;                        // last run whose offset is <= t
#pragma unroll
            for (int it = 0; it < (NR > 32 ? 6 : 5); ++it) { const int mid = (lo + hi + 1) >> 1; if (s_off[wave][mid] <= t) lo = mid; else hi = mid - 1; }
            const int j = s_js[wave][lo] + (t - s_off[wave][lo]);
            const float4 c = g.pts[j];
            cc = c;
            const float ex = qx - c.x, ey = qy - c.y, ez = qz - c.z;
            const float d2 = ex * ex + ey * ey + ez * ez;
            if (j != exclude && d2 < 3.0e38f) { ck = __float_as_uint(d2) & ~0x7Fu; cj = j; }     // NaN / Inf points are never listed
        }
        return ck;
    };
    // a sorted chunk (ck, cj per lane; kInfQ / -1 where empty) into the running best
    bool first = true;
    auto take = [&](unsigned ck_in, int cj, const float4 cc) {
        unsigned k = ck_in | (unsigned)lane;
        k = sort64x32(k, lane);
        const int cs = __shfl(cj, (int)(k & 63u));
        const unsigned ck = k & ~0x7Fu;
        if (have_c) { const int sl = (int)(k & 63u); topc = make_float4(__shfl(cc.x, sl), __shfl(cc.y, sl), __shfl(cc.z, sl), 0.f); }
        if (first) { topk = ck; topi = cs; first = false; }
        else {
            // the lane-wise minimum of the running list and the reversed chunk is the 64 smallest of both as one bitonic sequence
            const unsigned kt = topk | 64u | (unsigned)lane;
            const unsigned kcr = (unsigned)__shfl((int)(ck | (unsigned)lane), 63 - lane);
            const unsigned m = half_cleaners32<32>(min(kt, kcr), lane);
            const int pos = (int)(m & 63u);
            const int from_top = __shfl(topi, pos), from_chunk = __shfl(cs, pos);
            topi = (m & 64u) ? from_top : from_chunk;
            topk = m & ~0x7Fu;
        }
    };
    if (!kFilter || b.total <= 64) {                        // one chunk: nothing to save, and the records travel in registers (have_c); or no filter
#pragma unroll 1
        for (int c0 = 0; c0 < b.total; c0 += 64) {
            int cj; float4 cc;
            const unsigned ck = candidate(c0 + lane, cj, cc);
            take(ck, cj, cc);
        }
        return;
    }
    int pend = 0;                                           // entries waiting in the queue (wave-uniform)
#pragma unroll 1
    for (int c0 = 0; c0 < b.total; c0 += 64) {
        int cj; float4 cc;
        const unsigned ck = candidate(c0 + lane, cj, cc);
        const bool in = cj >= 0 && __uint_as_float(ck) <= rc2;
        const unsigned long long m = __ballot(in);
        if (in) {
            const int slot = pend + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            s_qk[wave][slot] = ck; s_qj[wave][slot] = cj;
        }
        pend += __popcll(m);
        __builtin_amdgcn_wave_barrier();
        if (pend >= 64) {
            const unsigned qk = s_qk[wave][lane]; const int qj = s_qj[wave][lane];
            const int rest = pend - 64;
            unsigned rk = 0u; int rj = 0;
            if (lane < rest) { rk = s_qk[wave][64 + lane]; rj = s_qj[wave][64 + lane]; }
            __builtin_amdgcn_wave_barrier();
            if (lane < rest) { s_qk[wave][lane] = rk; s_qj[wave][lane] = rj; }
            pend = rest;
            __builtin_amdgcn_wave_barrier();
            take(qk, qj, make_float4(0.f, 0.f, 0.f, 0.f));
        }
    }
    if (pend > 0 || first) {
        unsigned qk = kInfQ; int qj = -1;
        if (lane < pend) { qk = s_qk[wave][lane]; qj = s_qj[wave][lane]; }
        take(qk, qj, make_float4(0.f, 0.f, 0.f, 0.f));
    }
}

// one row: the entries of the list inside rho (the block's inscribed radius, or the first point left out), padded with (q, -1)
__device__ __forceinline__ float crow_emit(const GridIndex& g, float qx, float qy, float qz, float rc, unsigned topk, int topi,
                                           float4* __restrict__ row_out, float2* __restrict__ meta_out, bool& keep, float4& e,
                                           const float4 topc = make_float4(0.f, 0.f, 0.f, 0.f), bool have_c = false)
{
    const int lane = threadIdx.x & 63;
    const float tk = __uint_as_float(topk);
    const float dK = __shfl(tk, kGraphK - 1);
    const float rho2 = fminf(rc * rc, dK);
    keep = topi >= 0 && tk <= rho2;
    const int cnt = __popcll(__ballot(keep));
    e = make_float4(qx, qy, qz, __int_as_float(-1));
    if (keep) {
        float4 c = topc;
        if (!have_c) c = g.pts[topi];
        e = make_float4(c.x, c.y, c.z, __int_as_float(topi));
    }
    row_out[lane] = e;
    if (lane == 0) *meta_out = make_float2(rho2, __int_as_float(cnt));
    return rho2;
}

__device__ __forceinline__ void graph_row_q32(const GridIndex& g, const float4 q, int hx, int hy, int hz, int s,
                                              float4* __restrict__ row_out, float2* __restrict__ meta_out, int (*s_off)[64], int (*s_js)[64])
{
    const CrowBlock<2> b = crow_runs<2>(g, hx, hy, hz, s_off, s_js);
    unsigned topk; int topi; float4 topc; bool have_c;
    const float rc = crow_inscribed<2>(g, b, q.x, q.y, q.z);
    crow_list<2, true>(g, b, q.x, q.y, q.z, s_off, s_js, topk, topi, topc, have_c, s, rc * rc);
    bool keep; float4 e;
    (void)crow_emit(g, q.x, q.y, q.z, rc, topk, topi, row_out, meta_out, keep, e, topc, have_c);
}

// The row at the cell's centre q and, behind it, one row per octant in `mask`.  An octant row is derived from the centre row where that
// leaves it a useful reach: a point closer than rho - |octant centre - cell centre| to an octant centre is closer than rho to the cell
// centre, hence among the 64 listed — one 64-key sort re-orders the SAME entries by their distance from the octant centre, and that
// difference is the octant row's coverage radius (no candidate gather, one sort instead of chunk sorts + merges).  Where the centre row
// is cut short by its 64th entry (a dense spot: rho under 2.5 offsets), the octant row is built from the block like the centre row.
template <int R>
__device__ __forceinline__ void crow_build_wave(const GridIndex& g, const float4 q, int hx, int hy, int hz, unsigned mask,
                                                float4* __restrict__ row_out, float2* __restrict__ meta_out, int (*s_off)[64], int (*s_js)[64])
{
    const int lane = threadIdx.x & 63;
    constexpr float kEps = 1e-3f;
    const CrowBlock<R> b = crow_runs<R>(g, hx, hy, hz, s_off, s_js);
    unsigned topk; int topi; float4 topc; bool have_c;
    const float rc = crow_inscribed<R>(g, b, q.x, q.y, q.z);
    crow_list<R>(g, b, q.x, q.y, q.z, s_off, s_js, topk, topi, topc, have_c, -1, rc * rc);
    bool keep; float4 e;
    const float rho2 = crow_emit(g, q.x, q.y, q.z, rc, topk, topi, row_out, meta_out, keep, e, topc, have_c);
    if (mask) {
        const float off = 0.25f * g.cell * 1.7320508f;                         // |octant centre - cell centre|
        const float rho_o = fmaxf(sqrtf(rho2) - off - 2.f * kEps, 0.f), rho_o2 = rho_o * rho_o;
        const bool derive = rho_o >= 1.5f * off;
        int slot = 1;
#pragma unroll 1
        for (int o = 0; o < 8; ++o) {
            if (!((mask >> o) & 1u)) continue;
            const float mx = crow_centre(g.ox, g.cell, hx, o & 1 ? 0.75f : 0.25f), my = crow_centre(g.oy, g.cell, hy, o & 2 ? 0.75f : 0.25f);
            const float mz = crow_centre(g.oz, g.cell, hz, o & 4 ? 0.75f : 0.25f);
            float4* orow = row_out + (size_t)slot * kGraphK;
            if (derive) {
                unsigned k = kInfQ | (unsigned)lane;
                if (keep) {
                    const float ex = mx - e.x, ey = my - e.y, ez = mz - e.z;
                    k = (__float_as_uint(ex * ex + ey * ey + ez * ez) & ~0x7Fu) | (unsigned)lane;
                }
                k = sort64x32(k, lane);
                const int sl = (int)(k & 63u);
                const int oi = __shfl(keep ? topi : -1, sl);
                // the coordinates travel with the id: they are the centre row's entries, in some lane's registers (no second gather)
                const float ox = __shfl(e.x, sl), oy = __shfl(e.y, sl), oz = __shfl(e.z, sl);
                const bool okeep = oi >= 0 && __uint_as_float(k & ~0x7Fu) <= rho_o2;
                const int ocnt = __popcll(__ballot(okeep));
                float4 oe = make_float4(mx, my, mz, __int_as_float(-1));
                if (okeep) oe = make_float4(ox, oy, oz, __int_as_float(oi));
                orow[lane] = oe;
                if (lane == 0) meta_out[slot] = make_float2(rho_o2, __int_as_float(ocnt));
            } else {
                unsigned ok; int oi; bool okeep; float4 oe, oc; bool ohave;
                const float orc = crow_inscribed<R>(g, b, mx, my, mz);
                crow_list<R>(g, b, mx, my, mz, s_off, s_js, ok, oi, oc, ohave, -1, orc * orc);
                (void)crow_emit(g, mx, my, mz, orc, ok, oi, orow, meta_out + slot, okeep, oe, oc, ohave);
            }
            ++slot;
        }
    }
}

// One wave per kCrowCPW consecutive cells: their table entries written lane-parallel (most cells of a grid have no row: one coalesced
// load and store for all of them), then the cells that have rows one after the other, the whole wave on each.
// crow_tab[cell] = -2: nothing within two cells; -1: no row (its rows did not fit the capacity the buffers were sized for); else
// (first row << 8) | octant mask, rows = [centre, the octants of the mask in ascending order].
#ifndef LISREG_CROW_WPB
#define LISREG_CROW_WPB 1            // wavefronts per workgroup of the row build.  Measured (same box, interleaved): 1 -> 213-236 us per launch, 2 -> 222,
                                     // 4 (rounds 4-5) -> 248-270, 16 -> 370: most cells of a grid have no row, a few have nine — a workgroup of four
                                     // wavefronts holds its slots until the slowest of them is through
#endif
#ifndef LISREG_CROW_CPW
#define LISREG_CROW_CPW 8
#endif
constexpr int kCrowCPW = LISREG_CROW_CPW;
#ifndef LISREG_CROW_WAVES
#define LISREG_CROW_WAVES 6          // waves per SIMD the row build is compiled for (0: the compiler's choice = 86 registers, 5 waves).  The build is a
                                     // chain of dependent loads per cell: measured (profiles/r05_kernel_experiments.md) 6 waves with 3 spilled registers beat 5
                                     // without by 2-3 % of a configs[1] step; 7 and 8 waves (10 / 18 spilled) lose it again
#endif
__device__ __forceinline__ void crow_build_body(const GridIndex& g, int n_cells, const int* __restrict__ need, int* __restrict__ omask,
                                                const int* __restrict__ scan, int cap, int use_r3 /* 0: never the 7^3 block (experiments) */,
                                                int blk_, int (*s_off)[64], int (*s_js)[64])
{
    const int lane = threadIdx.x & 63;
    const int first = __builtin_amdgcn_readfirstlane((blk_ * LISREG_CROW_WPB + (int)(threadIdx.x >> 6)) * kCrowCPW);
    int* tab = const_cast<int*>(g.crow_tab);
    int n_l = 0, b_l = 0, om_l = 0;
    if (lane < kCrowCPW && first + lane < n_cells) {
        n_l = need[first + lane];
        om_l = omask[first + lane];
        omask[first + lane] = 0;                            // handed back clean: the next classification ORs into it without a memset
        if (n_l) b_l = scan[first + lane];
        const bool fits = n_l != 0 && !(b_l + n_l > cap || b_l >= (1 << 23));
        tab[first + lane] = n_l == 0 ? ((om_l & kCrowUnreached) ? -1 : -2) : (fits ? (b_l << 8) | (om_l & 255) : -1);
        if (!fits) n_l = 0;
    }
    unsigned long long live = __ballot(n_l != 0);
#pragma unroll 1
    while (live) {
        const int i = __ffsll((long long)live) - 1;
        live &= live - 1;
        const int cid = first + i;
        const int b = __shfl(b_l, i);
        const unsigned om = (unsigned)__shfl(om_l, i), mask = om & 255u;
        const int hz = cid % g.nz, t = cid / g.nz, hy = t % g.ny, hx = t / g.ny;
        const float4 q = make_float4(crow_centre(g.ox, g.cell, hx, 0.5f), crow_centre(g.oy, g.cell, hy, 0.5f), crow_centre(g.oz, g.cell, hz, 0.5f), 0.f);
        float4* row = const_cast<float4*>(g.crow) + (size_t)b * kGraphK;
        float2* meta = const_cast<float2*>(g.crow_meta) + b;
        // a centre row serves queries up to 0.87 cells from the centre wherever they are relative to the surface; one with fewer than five
        // points inside sqrt(tau) is certified by the coverage radius alone, which then has to reach sqrt(tau) + 0.87 cells: where the
        // 5 x 5 x 5 block does not fill the row (2.5 cells of coverage), the 7 x 7 x 7 block is searched (3.5 cells; few candidates there)
        if (mask != 0u || !use_r3 || (int)(om >> 8) >= kGraphK) crow_build_wave<2>(g, q, hx, hy, hz, mask, row, meta, s_off, s_js);
        else crow_build_wave<3>(g, q, hx, hy, hz, 0u, row, meta, s_off, s_js);
    }
}

#if LISREG_CROW_WAVES
#define LISREG_CROW_BUILD_ATTR __global__ __launch_bounds__(64 * LISREG_CROW_WPB) __attribute__((amdgpu_waves_per_eu(LISREG_CROW_WAVES, LISREG_CROW_WAVES)))
#else
#define LISREG_CROW_BUILD_ATTR __global__ __launch_bounds__(64 * LISREG_CROW_WPB)
#endif
LISREG_CROW_BUILD_ATTR void k_crow_build(GridIndex g, int n_cells, const int* __restrict__ need, int* __restrict__ omask,
                                         const int* __restrict__ scan, int cap, int use_r3)
{
    __shared__ int s_off[LISREG_CROW_WPB][64], s_js[LISREG_CROW_WPB][64];
    crow_build_body(g, n_cells, need, omask, scan, cap, use_r3, blockIdx.x, s_off, s_js);
}

// Round 6: the corner and the surf target of a slot through the row build in ONE launch sequence (marks, classification, two scan launches,
// rows): workgroups [0, nb of job 0) work on job 0, the rest on job 1.  Until round 6 the corner target's (small) launches ran on a side
// stream underneath the surf target's — two event hops on the critical path of every step (a marker in front of the marks, a wait in front
// of the first correspondence launch: 6 us each on an otherwise gap-free stream).
struct CrowJob { GridIndex g; int n_cells; int* need; int* omask; int* scan; int cap; const unsigned* reach; int tiles_y; int plain; int nb; };
struct CrowJobs { CrowJob j[2]; };
__global__ __launch_bounds__(256) void k_crow_mark_pair(CrowJobs J, float margin_cells)
{
    const int k = (int)blockIdx.x < J.j[0].nb ? 0 : 1;
    const CrowJob& j = J.j[k];
    crow_mark_body(j.g, margin_cells * j.g.cell, j.omask, (int)blockIdx.x - (k ? J.j[0].nb : 0));
}
__global__ __launch_bounds__(256) void k_crow_classify_pair(CrowJobs J, float margin_cells)
{
    extern __shared__ int s_cs[];
    const int k = (int)blockIdx.x < J.j[0].nb ? 0 : 1;
    const CrowJob& j = J.j[k];
    const int blk = (int)blockIdx.x - (k ? J.j[0].nb : 0);
    if (j.plain) crow_classify_plain_body(j.g, j.n_cells, j.need, j.omask, j.reach, blk);
    else crow_classify_body(j.g, j.tiles_y, j.need, j.omask, j.reach, blk, s_cs, margin_cells * j.g.cell);
}
LISREG_CROW_BUILD_ATTR void k_crow_build_pair(CrowJobs J, int use_r3)
{
    __shared__ int s_off[LISREG_CROW_WPB][64], s_js[LISREG_CROW_WPB][64];
    const int k = (int)blockIdx.x < J.j[0].nb ? 0 : 1;
    const CrowJob& j = J.j[k];
    crow_build_body(j.g, j.n_cells, j.need, j.omask, j.scan, j.cap, use_r3, (int)blockIdx.x - (k ? J.j[0].nb : 0), s_off, s_js);
}

// Coherence probe for sort_sources = auto: how many consecutive source points are further apart than `thr`?
// Scan order and voxel-grid order give a few per cent; an arbitrary order gives most of them.
__global__ __launch_bounds__(kBlockQ) void k_count_jumps(const BlockDesc* __restrict__ blocks, const Segment* __restrict__ segs,
                                                         float thr2, int* __restrict__ jumps)
{
    const BlockDesc bd = blocks[blockIdx.x];
    const Segment sg = segs[bd.seg];
    const int e = bd.start + threadIdx.x;
    bool jump = false;
    if ((int)threadIdx.x < bd.count && e + 1 < sg.n) {
        const float4 a = sg.src[e], b = sg.src[e + 1];
        const float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
        jump = dx * dx + dy * dy + dz * dz > thr2;
    }
    const unsigned long long m = __ballot(jump);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(jumps, __popcll(m));
}

// ---- §8 f-1: pcl::VoxelGrid (default settings) -----------------------------------------------------------------------
// Same deterministic bucket sort, keyed by PCL's voxel index idx = ijk0 + ijk1*div0 + ijk2*div0*div1
// (filters/impl/voxel_grid.hpp; call sites /root/reference/src/node/odomEstimationNode.cpp:196-201, 272-277 and
// src/include/subMap.h:1207-1249).  The index space can be as large as 2^31, so a bucket is a RANGE of `span`
// consecutive indices and the rank pass orders each bucket by (idx, input order): the sorted sequence is exactly
// "ascending idx, ties by input index" — PCL's output order with the summation order fixed.
__global__ __launch_bounds__(256) void k_voxel_keys(const float4* __restrict__ pts, int n, VoxelDesc d,
                                                    uint32_t* __restrict__ elem_bucket, uint32_t* __restrict__ elem_sub,
                                                    int* __restrict__ hist)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 p = pts[i];
    // ijk = static_cast<int>(floor(p * inverse_leaf) - static_cast<float>(min_b))
    const int i0 = (int)(floorf(p.x * d.inv_leaf) - (float)d.min_b0);
    const int i1 = (int)(floorf(p.y * d.inv_leaf) - (float)d.min_b1);
    const int i2 = (int)(floorf(p.z * d.inv_leaf) - (float)d.min_b2);
    const uint32_t idx = (uint32_t)(i0 + i1 * d.mul1 + i2 * d.mul2);
    const uint32_t b = idx / d.span;
    elem_bucket[i] = b;
    elem_sub[i] = idx - b * d.span;
    atomicAdd(&hist[b], 1);
}

// K clouds in one sort (lisreg_voxel_downsample_multi): the clouds are concatenated, every cloud keeps its own grid geometry, and its
// buckets follow the previous cloud's — the sorted sequence is cloud by cloud, each "ascending idx, ties by input index" as above.
__global__ __launch_bounds__(256) void k_voxel_keys_multi(const float4* __restrict__ pts, int n, VoxelMulti m,
                                                          uint32_t* __restrict__ elem_bucket, uint32_t* __restrict__ elem_sub,
                                                          int* __restrict__ hist)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int s = 0;
#pragma unroll
    for (int k = 1; k < kVoxelMultiMax; ++k) if (k < m.k && i >= m.off[k]) s = k;
    const VoxelDesc d = m.d[s];
    const float4 p = pts[i];
    const int i0 = (int)(floorf(p.x * d.inv_leaf) - (float)d.min_b0);
    const int i1 = (int)(floorf(p.y * d.inv_leaf) - (float)d.min_b1);
    const int i2 = (int)(floorf(p.z * d.inv_leaf) - (float)d.min_b2);
    const uint32_t idx = (uint32_t)(i0 + i1 * d.mul1 + i2 * d.mul2);
    const uint32_t lb = idx / d.span;
    const uint32_t b = (uint32_t)m.bucket_base[s] + lb;
    elem_bucket[i] = b;
    elem_sub[i] = idx - lb * d.span;
    atomicAdd(&hist[b], 1);
}

__global__ __launch_bounds__(256) void k_voxel_rank(int n, uint32_t span, const uint32_t* __restrict__ tmp_bucket,
                                                    const uint32_t* __restrict__ tmp_sub, const int* __restrict__ tmp_idx,
                                                    const int* __restrict__ bucket_start, int* __restrict__ order,
                                                    uint32_t* __restrict__ sidx)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    int e;
    const int dst = span == 1u ? rank_in_bucket<true>(p, tmp_bucket, tmp_sub, tmp_idx, bucket_start, &e)
                               : rank_in_bucket<false>(p, tmp_bucket, tmp_sub, tmp_idx, bucket_start, &e);
    order[dst] = e;
    sidx[dst] = tmp_bucket[p] * span + tmp_sub[p];
}

__global__ __launch_bounds__(256) void k_voxel_rank_multi(int n, VoxelMulti m, const uint32_t* __restrict__ tmp_bucket,
                                                          const uint32_t* __restrict__ tmp_sub, const int* __restrict__ tmp_idx,
                                                          const int* __restrict__ bucket_start, int* __restrict__ order,
                                                          uint32_t* __restrict__ sidx)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const uint32_t b = tmp_bucket[p];
    int s = 0;
#pragma unroll
    for (int k = 1; k < kVoxelMultiMax; ++k) if (k < m.k && b >= (uint32_t)m.bucket_base[k]) s = k;
    int e;
    const int dst = m.d[s].span == 1u ? rank_in_bucket<true>(p, tmp_bucket, tmp_sub, tmp_idx, bucket_start, &e)
                                      : rank_in_bucket<false>(p, tmp_bucket, tmp_sub, tmp_idx, bucket_start, &e);
    order[dst] = e;
    // joint voxel index: ascends through the sorted sequence and changes exactly where the voxel (or the cloud) changes
    sidx[dst] = m.idx_base[s] + (b - (uint32_t)m.bucket_base[s]) * m.d[s].span + tmp_sub[p];
}

__global__ __launch_bounds__(256) void k_voxel_heads(int n, const uint32_t* __restrict__ sidx, int* __restrict__ head)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    head[p] = (p == 0 || sidx[p] != sidx[p - 1]) ? 1 : 0;
}

__global__ __launch_bounds__(256) void k_voxel_starts(int n, const int* __restrict__ head, const int* __restrict__ slot,
                                                      int* __restrict__ vstart)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    if (head[p]) vstart[slot[p]] = p;
    if (p == n - 1) vstart[slot[p] + head[p]] = n;
}

constexpr int kVoteBig = 24;            // voxels with more points are summed and voted by one wavefront each (k_voxel_big)
// CentroidPoint (common/impl/accumulators.hpp): xyz and intensity = sequential float sums / n, label = most frequent
// (smallest on ties).  One thread per output voxel with a handful of points; the loads of four points are in flight together, the
// sums stay strictly in input order (that order is part of the result).
// w_mode 0: .w is an intensity (averaged), labels (if any) in `labels`; w_mode 1: .w is a lisreg_dpoint payload whose
// low 16 bits are the label (majority-voted into the output payload).
__global__ __launch_bounds__(256) void k_voxel_centroids(int n_vox, const float4* __restrict__ pts,
                                                         const uint32_t* __restrict__ labels, int w_mode,
                                                         const int* __restrict__ order, const int* __restrict__ vstart,
                                                         float4* __restrict__ out_pts, uint32_t* __restrict__ out_labels)
{
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= n_vox) return;
    const int a = vstart[v], b = vstart[v + 1];
    if (b - a > kVoteBig) return;                              // k_voxel_big
    const bool vote = w_mode == 1 || labels;
    float sx = 0.f, sy = 0.f, sz = 0.f, sw = 0.f;
    // AccumulatorLabel through up to 8 distinct (label, count) slots held in registers (compile-time indices); a voxel with more
    // distinct labels than that falls back to the quadratic count below
    uint32_t sl[8]; int sc[8]; int ns = 0; bool overflow = false;
#pragma unroll
    for (int s = 0; s < 8; ++s) { sl[s] = 0xffffffffu; sc[s] = 0; }
#define LISREG_VOX_ADD(q_, e_) do { \
        sx += (q_).x; sy += (q_).y; sz += (q_).z; \
        if (w_mode == 0) sw += (q_).w; \
        if (vote && !overflow) { \
            const uint32_t lp_ = (w_mode == 1 ? __float_as_uint((q_).w) : labels[e_]) & 0xffffu; \
            bool hit_ = false; \
            _Pragma("unroll") for (int s = 0; s < 8; ++s) if (sl[s] == lp_) { ++sc[s]; hit_ = true; } \
            if (!hit_) { \
                if (ns < 8) { _Pragma("unroll") for (int s = 0; s < 8; ++s) if (s == ns) { sl[s] = lp_; sc[s] = 1; } ++ns; } \
                else overflow = true; \
            } \
        } } while (0)
    int p = a;
    for (; p + 4 <= b; p += 4) {
        const int e0 = order[p], e1 = order[p + 1], e2 = order[p + 2], e3 = order[p + 3];
        const float4 q0 = pts[e0], q1 = pts[e1], q2 = pts[e2], q3 = pts[e3];
        LISREG_VOX_ADD(q0, e0); LISREG_VOX_ADD(q1, e1); LISREG_VOX_ADD(q2, e2); LISREG_VOX_ADD(q3, e3);
    }
    for (; p < b; ++p) { const int e = order[p]; const float4 q = pts[e]; LISREG_VOX_ADD(q, e); }
#undef LISREG_VOX_ADD
    const float cnt = (float)(b - a);
    uint32_t best = 0u;
    if (vote) {
        int bestc = 0;
        if (!overflow) {
#pragma unroll
            for (int s = 0; s < 8; ++s)
                if (sc[s] > bestc || (sc[s] == bestc && sc[s] > 0 && sl[s] < best)) { bestc = sc[s]; best = sl[s]; }
        } else {
            for (int p2 = a; p2 < b; ++p2) {
                const int e = order[p2];
                const uint32_t lp = (w_mode == 1 ? __float_as_uint(pts[e].w) : labels[e]) & 0xffffu;
                int c = 0;
                for (int m = a; m < b; ++m) {
                    const int em = order[m];
                    c += (((w_mode == 1 ? __float_as_uint(pts[em].w) : labels[em]) & 0xffffu) == lp) ? 1 : 0;
                }
                if (c > bestc || (c == bestc && lp < best)) { bestc = c; best = lp; }
            }
        }
    }
    out_pts[v] = make_float4(sx / cnt, sy / cnt, sz / cnt, w_mode == 1 ? __uint_as_float(best) : sw / cnt);
    if (out_labels) out_labels[v] = best;
}

// The voxels k_voxel_centroids left out (more than kVoteBig points: the cells next to the sensor hold hundreds to thousands, and one
// lane walking through them — a dependent gather per point — was most of a grid's time).  One wavefront per voxel.  Centroid: the lanes
// fetch 64 points at a time into LDS and lane 0 adds them up in input order, so the float sums are the serial ones bit for bit.  Label:
// one pass per DISTINCT label in ascending order; every pass counts the current label and finds the next larger one (lanes stride over
// the points, wave reductions), so the most frequent label — the smallest among equals, as the serial vote keeps it — falls out
// without any table.
__global__ __launch_bounds__(64) void k_voxel_big(int n_vox, const float4* __restrict__ pts, const uint32_t* __restrict__ labels, int w_mode,
                                                  const int* __restrict__ order, const int* __restrict__ vstart,
                                                  float4* __restrict__ out_pts, uint32_t* __restrict__ out_labels)
{
    __shared__ float4 s_q[64];
    const int v = blockIdx.x;
    if (v >= n_vox) return;
    const int a = vstart[v], b = vstart[v + 1];
    if (b - a <= kVoteBig) return;
    const int lane = threadIdx.x;
    const bool vote = w_mode == 1 || labels;
    float sx = 0.f, sy = 0.f, sz = 0.f, sw = 0.f;
    unsigned cur = 0xffffffffu;
    for (int p0 = a; p0 < b; p0 += 64) {
        const int p = p0 + lane;
        if (p < b) {
            const int e = order[p];
            const float4 q = pts[e];
            s_q[lane] = q;
            if (vote) cur = min(cur, (w_mode == 1 ? __float_as_uint(q.w) : labels[e]) & 0xffffu);
        }
        __builtin_amdgcn_wave_barrier();                       // one wavefront: its LDS operations execute in order
        if (lane == 0) {
            const int m = min(64, b - p0);
            for (int i = 0; i < m; ++i) {
                const float4 q = s_q[i];
                sx += q.x; sy += q.y; sz += q.z;
                if (w_mode == 0) sw += q.w;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    unsigned best = 0u;
    if (vote) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) cur = min(cur, (unsigned)__shfl_xor((int)cur, d));
        int bestc = 0;
        while (cur != 0xffffffffu) {
            int c = 0; unsigned nxt = 0xffffffffu;
            for (int p = a + lane; p < b; p += 64) {
                const int e = order[p];
                const unsigned lp = (w_mode == 1 ? __float_as_uint(pts[e].w) : labels[e]) & 0xffffu;
                c += lp == cur ? 1 : 0;
                if (lp > cur) nxt = min(nxt, lp);
            }
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) { c += __shfl_xor(c, d); nxt = min(nxt, (unsigned)__shfl_xor((int)nxt, d)); }
            if (c > bestc) { bestc = c; best = cur; }              // ascending labels: a later equal count does not replace
            cur = nxt;
        }
    }
    if (lane == 0) {
        const float cnt = (float)(b - a);
        out_pts[v] = make_float4(sx / cnt, sy / cnt, sz / cnt, w_mode == 1 ? __uint_as_float(best) : sw / cnt);
        if (out_labels) out_labels[v] = best;
    }
}

// PCL point structs as they arrive over PCIe (stride-byte records: x, y, z at 0/4/8, uint16 label at 20 for PointXYZIL,
// src/include/common.h:9,25-35) -> 16-byte device records.  The 32-byte structs of the reference take two 16-byte loads.
__global__ __launch_bounds__(256) void k_pack_cloud(const unsigned char* __restrict__ raw, size_t n, int stride, int has_label,
                                                    float4* __restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned char* r = raw + i * (size_t)stride;
    float4 o;
    if ((stride & 15) == 0 && stride >= 32) {
        const uint4 a = *reinterpret_cast<const uint4*>(r), b = *reinterpret_cast<const uint4*>(r + 16);
        o = make_float4(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(has_label ? (b.y & 0xffffu) : 0u));
    } else {
        unsigned int w[3];
        for (int k = 0; k < 3; ++k)
            w[k] = (unsigned)r[4 * k] | ((unsigned)r[4 * k + 1] << 8) | ((unsigned)r[4 * k + 2] << 16) | ((unsigned)r[4 * k + 3] << 24);
        const unsigned lab = has_label ? ((unsigned)r[20] | ((unsigned)r[21] << 8)) : 0u;
        o = make_float4(__uint_as_float(w[0]), __uint_as_float(w[1]), __uint_as_float(w[2]), __uint_as_float(lab));
    }
    out[i] = o;
}

// transformPointCloud (src/core/common.cpp:112-173): p' = R p + t, the fourth channel is copied
__device__ __forceinline__ void transform_cloud_point(const float4* __restrict__ in, int n, const float* __restrict__ M12,
                                                      float4* __restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 p = in[i];
    // ((m0 x + m1 y) + m2 z) + m3 with every product and sum rounded on its own, like the reference's x86 build (no FMA
    // contraction): the transformed clouds feed voxel grids and box crops whose parity bar is exact.  HBM-bound either way.
    auto row = [&](int r) {
#pragma clang fp contract(off)
        const float a = M12[4 * r] * p.x, b = M12[4 * r + 1] * p.y, cc = M12[4 * r + 2] * p.z;
        const float s1 = a + b;
        const float s2 = s1 + cc;
        return s2 + M12[4 * r + 3];
    };
    out[i] = make_float4(row(0), row(1), row(2), p.w);
}
__global__ __launch_bounds__(256) void k_transform_cloud(const float4* __restrict__ in, int n, const float* __restrict__ M12,
                                                         float4* __restrict__ out) { transform_cloud_point(in, n, M12, out); }
// the same with the matrix as a kernel argument: nothing to upload, nothing to wait for
__global__ __launch_bounds__(256) void k_transform_cloud_m(const float4* __restrict__ in, int n, Mat12 M, float4* __restrict__ out)
{ transform_cloud_point(in, n, M.m, out); }

}  // namespace

// ---- launchers ---------------------------------------------------------------------------------------------
void launch_bbox(const float4* pts, int n, float* bbox6, float* scratch /* >= 6*256 floats */, hipStream_t st)
{
    int nb = (n + 255) / 256;
    if (nb > 256) nb = 256;
    if (nb < 1) nb = 1;
    k_bbox_partial<<<nb, 256, 0, st>>>(pts, n, scratch);
    k_bbox_final<<<1, 64, 0, st>>>(scratch, nb, bbox6);
}

void launch_build_target(const float4* pts, int n, GridIndex g, float4* sorted_out, int* cell_start_out,
                         int n_cells, SortBuffers sb, hipStream_t st)
{
    (void)hipMemsetAsync(sb.hist, 0, sizeof(int) * (size_t)n_cells, st);
    if (n > 0) k_target_keys<<<(n + 255) / 256, 256, 0, st>>>(pts, n, g, sb.elem_bucket, sb.elem_sub, sb.hist);
    exclusive_scan(sb.hist, cell_start_out, sb.scan_tmp, n_cells, st);
    if (n > 0) {
        k_target_scatter_pts<<<(n + 255) / 256, 256, 0, st>>>(pts, n, g, sb.elem_sub, cell_start_out, sb.tmp_pts);
        k_target_rank_pts<<<(n + 255) / 256, 256, 0, st>>>(n, g, sb.tmp_pts, cell_start_out, sorted_out);
    }
}

void launch_build_targets_batched(const BlockDesc* blocks, int n_blocks, const TargetSeg* tsegs, int n_tsegs,
                                  int n_elems, int n_buckets, SortBuffers sb, hipStream_t st)
{
    if (n_tsegs <= 0) return;
    (void)hipMemsetAsync(sb.hist, 0, sizeof(int) * (size_t)n_buckets, st);
    if (n_blocks > 0) k_tseg_keys<<<n_blocks, kBlockQ, 0, st>>>(blocks, tsegs, sb.elem_bucket, sb.elem_sub, sb.hist);
    exclusive_scan(sb.hist, sb.bucket_start, sb.scan_tmp, n_buckets, st);
    if (n_elems > 0) {
        k_tseg_scatter_pts<<<n_blocks, kBlockQ, 0, st>>>(blocks, tsegs, sb.elem_sub, sb.bucket_start, sb.tmp_pts);
        k_tseg_rank_pts<<<(n_elems + 255) / 256, 256, 0, st>>>(tsegs, n_tsegs, n_elems, sb.tmp_pts, sb.bucket_start);
    }
    k_tseg_cell_starts<<<dim3(64, n_tsegs), 256, 0, st>>>(tsegs, n_tsegs, sb.bucket_start);
}

int launch_build_targets_strips(const BlockDesc* chunks, int n_chunks, const TargetSeg* tsegs, int n_tsegs, int n_strips, int max_units,
                                int max_strip_cells, int cap_small, StripBuffers sb, hipStream_t st, int* zero_ints_known)
{
    if (n_tsegs <= 0) return 0;
    const size_t hist_bytes = (size_t)(max_strip_cells + 1) * 4;
    const size_t lds_small = hist_bytes + (size_t)cap_small * 6;
    if (hist_bytes + 6 * 1024 > kStripLdsLarge || lds_small > kStripLdsLarge) return 2;
    const int cap_large = (int)std::min<size_t>((kStripLdsLarge - hist_bytes) / 6, 65535);
    const size_t lds_large = hist_bytes + (size_t)cap_large * 6;
    static bool attr_set_dev[64] = { false };       // the attribute is per device; contexts of several devices may share the process
    int dev = 0;
    (void)hipGetDevice(&dev);
    bool& attr_set = attr_set_dev[dev & 63];
    if (!attr_set) {
        hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_strip_build<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kStripLdsLarge);
        hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_strip_build<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kStripLdsLarge);
        if (e1 != hipSuccess || e2 != hipSuccess) return 1;
        attr_set = true;
    }
    // cnt and fill are one allocation; a build hands them back clean (k_strip_build<false>) when they are small enough for one workgroup to wipe
    const int need_zero = 2 * (n_strips + 1);
    const bool hand_back = zero_ints_known && need_zero <= 65536;
    if (!(zero_ints_known && *zero_ints_known >= need_zero)) (void)hipMemsetAsync(sb.cnt, 0, sizeof(int) * (size_t)need_zero, st);
    if (zero_ints_known) *zero_ints_known = hand_back ? need_zero : 0;
    if (n_chunks > 0) k_strip_partition<false><<<n_chunks, kPartThreads, 0, st>>>(chunks, tsegs, sb.cnt, nullptr, nullptr, nullptr);
    if (n_chunks > 0 && n_strips < kMaxStrips)
        k_strip_partition<true, true><<<n_chunks, kPartThreads, 0, st>>>(chunks, tsegs, sb.cnt, nullptr, sb.fill, sb.tmp_pts, sb.start, n_strips);
    else {
        exclusive_scan(sb.cnt, sb.start, sb.scan_tmp, n_strips, st);
        if (n_chunks > 0) k_strip_partition<true><<<n_chunks, kPartThreads, 0, st>>>(chunks, tsegs, nullptr, sb.start, sb.fill, sb.tmp_pts);
    }
    const dim3 grid((unsigned)std::max(max_units, 1), (unsigned)n_tsegs);
    // the two variants work on disjoint strips: the few big strips (one workgroup per CU each) run on a side stream underneath
    // the many small ones instead of after them
    // (worth the two event hops only for big batches: measured +20 us on a single 200 k target, -200 us on 128 of them)
    const bool fork = n_chunks >= 512 && sb.side && sb.ev_fork && sb.ev_join && hipEventRecord(sb.ev_fork, st) == hipSuccess &&
                      hipStreamWaitEvent(sb.side, sb.ev_fork, 0) == hipSuccess;
    k_strip_build<true><<<grid, 1024, lds_large, fork ? sb.side : st>>>(tsegs, sb.start, sb.tmp_pts, cap_small, cap_large, sb.slot_idx, sb.slot_pos);
    if (fork) (void)hipEventRecord(sb.ev_join, sb.side);
    k_strip_build<false><<<grid, 256, lds_small, st>>>(tsegs, sb.start, sb.tmp_pts, cap_small, cap_large, sb.slot_idx, sb.slot_pos,
                                                      hand_back ? sb.cnt : nullptr, hand_back ? need_zero : 0);
    if (fork) (void)hipStreamWaitEvent(st, sb.ev_join, 0);
    return 0;
}

void launch_build_graph(const BlockDesc* blocks, int n_blocks, const TargetSeg* tsegs, const GridIndex* grids,
                        hipStream_t st)
{
    if (n_blocks <= 0) return;
    k_graph_build_batched<<<n_blocks * (kBlockQ / (LISREG_GRAPH_WPB * kGraphPPW)), 64 * LISREG_GRAPH_WPB, 0, st>>>(blocks, tsegs, grids);
}

void launch_build_graph_one(GridIndex g, hipStream_t st)
{
    if (g.n <= 0 || !g.nbr) return;
    k_graph_build_one<<<(g.n + LISREG_GRAPH_WPB * kGraphPPW - 1) / (LISREG_GRAPH_WPB * kGraphPPW), 64 * LISREG_GRAPH_WPB, 0, st>>>(g);
}

// Query marks: one thread per source point of the batch.  Consecutive points of a sweep fall into the same cell a dozen at a time: a lane
// whose left neighbour names the same bit leaves it to that lane, and a lane that finds its bit set already (a plain, possibly stale,
// load: a stale miss only repeats an atomic) sends nothing — after the first few thousand wavefronts almost no atomic is sent.
__global__ __launch_bounds__(kBlockQ) void k_query_marks(const BlockDesc* __restrict__ blocks, const Segment* __restrict__ segs,
                                                         const GridIndex* __restrict__ grids, const ItemState* __restrict__ items)
{
    const BlockDesc bd = blocks[blockIdx.x];
    const Segment sg = segs[bd.seg];
    const GridIndex g = grids[sg.target];
    if (!g.qmark || g.n <= 0) return;
    const float* M = items[bd.item].M;
    const int tid = threadIdx.x, lane = tid & 63;
    const bool valid = tid < bd.count;
    float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) q0 = sg.src[bd.start + tid];
    const float qx = M[0] * q0.x + M[1] * q0.y + M[2] * q0.z + M[3];
    const float qy = M[4] * q0.x + M[5] * q0.y + M[6] * q0.z + M[7];
    const float qz = M[8] * q0.x + M[9] * q0.y + M[10] * q0.z + M[11];
    // the cell the scan would take the row of: the query's own, clamped into the grid (fminf / fmaxf first: a NaN or huge coordinate must not
    // reach the float -> int conversion)
    const int hx = (int)fminf(fmaxf(floorf((qx - g.ox) * g.inv_cell), 0.f), (float)(g.nx - 1));
    const int hy = (int)fminf(fmaxf(floorf((qy - g.oy) * g.inv_cell), 0.f), (float)(g.ny - 1));
    const int hz = (int)fminf(fmaxf(floorf((qz - g.oz) * g.inv_cell), 0.f), (float)(g.nz - 1));
    const int word = (hx * g.ny + hy) * g.qmark_w + (hz >> 5);
    const unsigned bit = 1u << (hz & 31);
    const int key = valid ? (word << 5) | (hz & 31) : -1 - lane;
    const int pk = __builtin_amdgcn_update_dpp(-2 - lane, key, 0x111, 0xF, 0xF, false);       // left neighbour in the 16-lane row
    if (valid && pk != key && (g.qmark[word] & bit) == 0u) atomicOr(&g.qmark[word], bit);
}

// reach = the marks of the (2 D + 1)^3 block around every cell: one thread per (column, word); z through shifts with the carries of the
// column's neighbouring words, x / y through the (2 D + 1)^2 columns around.  D cells = one metre (at least two cells): the distance the
// first Gauss-Newton steps may move a query from where its initial pose put it without leaving the cells that have rows.
__global__ __launch_bounds__(256) void k_reach_dilate(GridIndex g, unsigned* __restrict__ reach, int D)
{
    const int W = g.qmark_w;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= g.nx * g.ny * W) return;
    const int col = i / W, w = i - col * W, ix = col / g.ny, iy = col - ix * g.ny;
    unsigned out = 0u;
#pragma unroll 1
    for (int dx = -D; dx <= D; ++dx) {
        const int x = ix + dx;
        if (x < 0 || x >= g.nx) continue;
#pragma unroll 1
        for (int dy = -D; dy <= D; ++dy) {
            const int y = iy + dy;
            if (y < 0 || y >= g.ny) continue;
            const unsigned* c = g.qmark + (size_t)(x * g.ny + y) * W;
            const unsigned m = c[w], lo = w > 0 ? c[w - 1] : 0u, hi = w + 1 < W ? c[w + 1] : 0u;
            out |= m;
#pragma unroll 1
            for (int sft = 1; sft <= D; ++sft) out |= (m << sft) | (m >> sft) | (lo >> (32 - sft)) | (hi << (32 - sft));
        }
    }
    reach[i] = out;
}

void launch_query_marks(const BlockDesc* blocks, int n_blocks, const Segment* segs, const GridIndex* grids, const ItemState* items, hipStream_t st)
{
    if (n_blocks > 0) k_query_marks<<<n_blocks, kBlockQ, 0, st>>>(blocks, segs, grids, items);
}

void launch_reach_dilate(GridIndex g, unsigned* reach, hipStream_t st)
{
    if (!g.qmark || !reach || g.n <= 0) return;
    const int n = g.nx * g.ny * g.qmark_w;
    const int D = std::min(std::max((int)std::ceil(1.0f / g.cell - 1e-3f), 2), 16);
    k_reach_dilate<<<(n + 255) / 256, 256, 0, st>>>(g, reach, D);
}

// dynamic LDS of the tiled classification: the staged cell_start rows of the tile + its two-column rim, and the tile's octant masks
static size_t crow_classify_lds(int nz)
{
    return sizeof(int) * ((size_t)(kCtX + 2 * kCtRim) * (kCtY + 2 * kCtRim) * (size_t)(nz + 1) + (size_t)kCtX * kCtY * (size_t)nz);
}

float crow_oct_margin_cells()
{
    static const float margin = std::min(0.249f, std::max(0.f, getenv("LISREG_CROW_MARGIN") ? (float)atof(getenv("LISREG_CROW_MARGIN")) : 0.249f));
    return margin;
}

void launch_crow_classify(GridIndex g, int n_cells, CrowBuffers cb, hipStream_t st, int* omask_zero_cells)
{
    if (g.n <= 0 || n_cells <= 0) return;
    const bool zero_already = omask_zero_cells && *omask_zero_cells >= n_cells;
    if (omask_zero_cells) *omask_zero_cells = 0;              // the marks go in now; launch_crow_build takes them out again
    // octant margin in cells: under a quarter cell (half an octant's edge), so that a point is near at most two octants per axis
    const float margin = crow_oct_margin_cells();
    const size_t lds = crow_classify_lds(g.nz);
    if (lds <= 64 * 1024) {                                   // the tiled classification makes the marks itself, in LDS
        const int tiles_x = (g.nx + kCtX - 1) / kCtX, tiles_y = (g.ny + kCtY - 1) / kCtY;
        k_crow_classify<<<tiles_x * tiles_y, 256, lds, st>>>(g, tiles_y, cb.need, cb.omask, g.qmark ? cb.reach : nullptr, margin * g.cell);
    } else {
        if (!zero_already) (void)hipMemsetAsync(cb.omask, 0, sizeof(int) * (size_t)n_cells, st);
        k_crow_mark<<<(g.n + 255) / 256, 256, 0, st>>>(g, margin * g.cell, cb.omask);
        k_crow_classify_plain<<<(n_cells + 255) / 256, 256, 0, st>>>(g, n_cells, cb.need, cb.omask, g.qmark ? cb.reach : nullptr);
    }
    exclusive_scan(cb.need, cb.scan, cb.scan_tmp, n_cells, st);
}

void launch_crow_build(GridIndex g, int n_cells, CrowBuffers cb, hipStream_t st, int* omask_zero_cells)
{
    if (g.n <= 0 || n_cells <= 0 || cb.cap_rows <= 0 || !g.crow) return;
    if (omask_zero_cells) *omask_zero_cells = n_cells;        // (k_crow_build zeroes the mask of every cell it is dealt, with or without rows)
    static const int use_r3 = getenv("LISREG_CROW_R3") ? atoi(getenv("LISREG_CROW_R3")) : 1;
    k_crow_build<<<(n_cells + LISREG_CROW_WPB * kCrowCPW - 1) / (LISREG_CROW_WPB * kCrowCPW), 64 * LISREG_CROW_WPB, 0, st>>>(g, n_cells, cb.need, cb.omask, cb.scan, cb.cap_rows, use_r3);
}

// classification + rows of the corner (job 0) and surf (job 1) target of one slot, five launches on ONE stream (see CrowJob)
void launch_crow_rows_pair(const GridIndex g[2], const int n_cells[2], const CrowBuffers cb[2], hipStream_t st, int* omask_zero_cells[2])
{
    const float margin = crow_oct_margin_cells();
    static const int use_r3 = getenv("LISREG_CROW_R3") ? atoi(getenv("LISREG_CROW_R3")) : 1;
    CrowJobs J = {};
    bool on[2];
    for (int k = 0; k < 2; ++k) {
        on[k] = g[k].n > 0 && n_cells[k] > 0 && cb[k].cap_rows > 0 && g[k].crow != nullptr;
        CrowJob& j = J.j[k];
        j.g = g[k]; j.n_cells = n_cells[k]; j.need = cb[k].need; j.omask = cb[k].omask; j.scan = cb[k].scan; j.cap = cb[k].cap_rows;
        j.reach = g[k].qmark ? cb[k].reach : nullptr;
        if (!on[k]) continue;
        j.plain = crow_classify_lds(g[k].nz) > 64 * 1024 ? 1 : 0;          // (the tiled classification makes its marks itself, in LDS)
        const bool zero_already = omask_zero_cells[k] && *omask_zero_cells[k] >= n_cells[k];
        if (j.plain && !zero_already) (void)hipMemsetAsync(cb[k].omask, 0, sizeof(int) * (size_t)n_cells[k], st);
        if (omask_zero_cells[k]) *omask_zero_cells[k] = n_cells[k];      // (k_crow_build_pair hands every mask back as zero)
    }
    if (!on[0] && !on[1]) return;
    auto grid = [&](int a, int b) { J.j[0].nb = on[0] ? a : 0; J.j[1].nb = on[1] ? b : 0; return (unsigned)(J.j[0].nb + J.j[1].nb); };
    // marks in a launch of their own only for targets whose tile does not fit the LDS (z-ranges beyond ~100 cells)
    {
        const unsigned nb = grid(J.j[0].plain ? (g[0].n + 255) / 256 : 0, J.j[1].plain ? (g[1].n + 255) / 256 : 0);
        if (nb) k_crow_mark_pair<<<nb, 256, 0, st>>>(J, margin);
    }
    // classification: LDS-tiled where the tile fits, per cell otherwise (decided per target)
    {
        size_t lds = 0; int nbk[2] = { 0, 0 };
        for (int k = 0; k < 2; ++k) {
            if (!on[k]) continue;
            if (!J.j[k].plain) { J.j[k].tiles_y = (g[k].ny + kCtY - 1) / kCtY; nbk[k] = ((g[k].nx + kCtX - 1) / kCtX) * J.j[k].tiles_y; lds = std::max(lds, crow_classify_lds(g[k].nz)); }
            else nbk[k] = (n_cells[k] + 255) / 256;
        }
        const unsigned nb = grid(nbk[0], nbk[1]);
        k_crow_classify_pair<<<nb, 256, lds, st>>>(J, margin);
    }
    // row counts -> first rows: both scans in two launches (tiled form whatever the size)
    {
        ScanPair sp = {};
        for (int k = 0; k < 2; ++k) {
            sp.in[k] = cb[k].need; sp.out[k] = cb[k].scan; sp.tmp[k] = cb[k].scan_tmp; sp.n[k] = on[k] ? n_cells[k] : 0;
            sp.nb[k] = on[k] ? (n_cells[k] + kScanTile - 1) / kScanTile : 0;
        }
        if (sp.nb[0] <= kScanFusedMaxTiles && sp.nb[1] <= kScanFusedMaxTiles) {
            k_scan_local_pair<<<sp.nb[0] + sp.nb[1], kScanBlock, 0, st>>>(sp);
            k_scan_add_sum_pair<<<sp.nb[0] + sp.nb[1], kScanBlock, 0, st>>>(sp);
        } else
            for (int k = 0; k < 2; ++k) if (on[k]) exclusive_scan(cb[k].need, cb[k].scan, cb[k].scan_tmp, n_cells[k], st);
    }
    // rows
    {
        const int per = LISREG_CROW_WPB * kCrowCPW;
        const unsigned nb = grid((n_cells[0] + per - 1) / per, (n_cells[1] + per - 1) / per);
        k_crow_build_pair<<<nb, 64 * LISREG_CROW_WPB, 0, st>>>(J, use_r3);
    }
}

void launch_sort_sources(const BlockDesc* blocks, int n_blocks, const Segment* segs, int n_segs,
                         const ItemState* items, int n_elems, int n_buckets, SortBuffers sb, float4* sorted_all,
                         int* order_all, hipStream_t st)
{
    if (n_buckets <= 0) return;              // sorting disabled: the kernels read the caller's records in place
    (void)hipMemsetAsync(sb.hist, 0, sizeof(int) * (size_t)n_buckets, st);
    if (n_blocks > 0)
        k_source_keys<<<n_blocks, kBlockQ, 0, st>>>(blocks, segs, items, sb.elem_bucket, sb.elem_sub, sb.hist);
    exclusive_scan(sb.hist, sb.bucket_start, sb.scan_tmp, n_buckets, st);
    if (n_elems > 0) {
        k_scatter<<<(n_elems + 255) / 256, 256, 0, st>>>(sb.elem_bucket, sb.elem_sub, n_elems, sb.bucket_start,
                                                         sb.hist, sb.tmp_bucket, sb.tmp_sub, sb.tmp_idx);
        k_rank_source<<<(n_elems + 255) / 256, 256, 0, st>>>(segs, n_segs, n_elems, sb.tmp_bucket, sb.tmp_sub,
                                                             sb.tmp_idx, sb.bucket_start, sorted_all, order_all);
    }
}

// sort the cloud by voxel index; head flags + their exclusive scan (slot[n] = number of voxels) are left on the device
void launch_voxel_sort(const float4* pts, int n, VoxelDesc d, int n_buckets, SortBuffers sb, int* order, uint32_t* sidx,
                       int* head, int* slot, hipStream_t st)
{
    (void)hipMemsetAsync(sb.hist, 0, sizeof(int) * (size_t)n_buckets, st);
    k_voxel_keys<<<(n + 255) / 256, 256, 0, st>>>(pts, n, d, sb.elem_bucket, sb.elem_sub, sb.hist);
    exclusive_scan(sb.hist, sb.bucket_start, sb.scan_tmp, n_buckets, st);
    k_scatter<<<(n + 255) / 256, 256, 0, st>>>(sb.elem_bucket, sb.elem_sub, n, sb.bucket_start, sb.hist, sb.tmp_bucket,
                                               sb.tmp_sub, sb.tmp_idx);
    k_voxel_rank<<<(n + 255) / 256, 256, 0, st>>>(n, d.span, sb.tmp_bucket, sb.tmp_sub, sb.tmp_idx, sb.bucket_start, order, sidx);
    k_voxel_heads<<<(n + 255) / 256, 256, 0, st>>>(n, sidx, head);
    exclusive_scan(head, slot, sb.scan_tmp, n, st);
}

void launch_voxel_sort_multi(const float4* pts, int n, const VoxelMulti& m, int n_buckets, SortBuffers sb, int* order, uint32_t* sidx,
                             int* head, int* slot, hipStream_t st)
{
    (void)hipMemsetAsync(sb.hist, 0, sizeof(int) * (size_t)n_buckets, st);
    k_voxel_keys_multi<<<(n + 255) / 256, 256, 0, st>>>(pts, n, m, sb.elem_bucket, sb.elem_sub, sb.hist);
    exclusive_scan(sb.hist, sb.bucket_start, sb.scan_tmp, n_buckets, st);
    k_scatter<<<(n + 255) / 256, 256, 0, st>>>(sb.elem_bucket, sb.elem_sub, n, sb.bucket_start, sb.hist, sb.tmp_bucket,
                                               sb.tmp_sub, sb.tmp_idx);
    k_voxel_rank_multi<<<(n + 255) / 256, 256, 0, st>>>(n, m, sb.tmp_bucket, sb.tmp_sub, sb.tmp_idx, sb.bucket_start, order, sidx);
    k_voxel_heads<<<(n + 255) / 256, 256, 0, st>>>(n, sidx, head);
    exclusive_scan(head, slot, sb.scan_tmp, n, st);
}

void launch_voxel_centroids(int n, int n_vox, const float4* pts, const uint32_t* labels, int w_mode, const int* order,
                            const int* head, const int* slot, int* vstart, float4* out_pts, uint32_t* out_labels,
                            hipStream_t st)
{
    k_voxel_starts<<<(n + 255) / 256, 256, 0, st>>>(n, head, slot, vstart);
    k_voxel_centroids<<<(n_vox + 255) / 256, 256, 0, st>>>(n_vox, pts, labels, w_mode, order, vstart, out_pts, out_labels);
    if (n_vox > 0) k_voxel_big<<<n_vox, 64, 0, st>>>(n_vox, pts, labels, w_mode, order, vstart, out_pts, out_labels);
}

// K bounding boxes of the clouds of a VoxelMulti (cloud s = records [off[s], off[s + 1]) of `cat`) in two launches: 64 partial rows per
// cloud, then one wavefront per cloud; bbox_out[6 * s ...], scratch >= 6 * 64 * K floats.  An empty cloud gets the empty box.
__global__ __launch_bounds__(256) void k_bbox_partial_multi(const float4* __restrict__ cat, VoxelMulti m, float* __restrict__ part)
{
    const int s = blockIdx.y;
    k_bbox_block(cat + m.off[s], m.off[s + 1] - m.off[s], part + (size_t)s * 6 * gridDim.x);
}
__global__ void k_bbox_final_multi(const float* __restrict__ part, int nb, float* __restrict__ bbox_out)
{
    k_bbox_fold(part + (size_t)blockIdx.x * 6 * nb, nb, bbox_out + 6 * blockIdx.x);
}
void launch_bbox_multi(const float4* cat, const VoxelMulti& m, float* bbox_out, float* scratch, hipStream_t st)
{
    if (m.k <= 0) return;
    k_bbox_partial_multi<<<dim3(64, (unsigned)m.k), 256, 0, st>>>(cat, m, scratch);
    k_bbox_final_multi<<<m.k, 64, 0, st>>>(scratch, 64, bbox_out);
}

// the same for K clouds that live in buffers of their own
__global__ __launch_bounds__(256) void k_bbox_partial_jobs(BboxJobs j, float* __restrict__ part)
{
    k_bbox_block(j.pts[blockIdx.y], j.n[blockIdx.y], part + (size_t)blockIdx.y * 6 * gridDim.x);
}
void launch_bbox_jobs(const BboxJobs& j, float* bbox_out, float* scratch, hipStream_t st)
{
    if (j.k <= 0) return;
    k_bbox_partial_jobs<<<dim3(64, (unsigned)j.k), 256, 0, st>>>(j, scratch);
    k_bbox_final_multi<<<j.k, 64, 0, st>>>(scratch, 64, bbox_out);
}

// the same for ANY number of clouds, described by a device table (lisreg_map_index_set_batch)
__global__ __launch_bounds__(256) void k_bbox_partial_refs(const CloudRef* __restrict__ refs, float* __restrict__ part)
{
    const CloudRef r = refs[blockIdx.y];
    k_bbox_block(r.pts, r.n, part + (size_t)blockIdx.y * 6 * gridDim.x);
}
void launch_bbox_refs(const CloudRef* refs_dev, int k, float* bbox_out, float* scratch, hipStream_t st)
{
    if (k <= 0) return;
    k_bbox_partial_refs<<<dim3(64, (unsigned)k), 256, 0, st>>>(refs_dev, scratch);
    k_bbox_final_multi<<<k, 64, 0, st>>>(scratch, 64, bbox_out);
}

// K clouds copied end to end into one array (cloud s to [off[s], off[s + 1])) by one launch
__global__ __launch_bounds__(256) void k_concat_jobs(BboxJobs j, VoxelMulti m, float4* __restrict__ cat)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= m.off[m.k]) return;
    int s = 0;
#pragma unroll
    for (int q = 1; q < kVoxelMultiMax; ++q) if (q < m.k && i >= m.off[q]) s = q;
    cat[i] = j.pts[s][i - m.off[s]];
}
void launch_concat_jobs(const BboxJobs& j, const VoxelMulti& m, float4* cat, hipStream_t st)
{
    if (m.off[m.k] > 0) k_concat_jobs<<<(m.off[m.k] + 255) / 256, 256, 0, st>>>(j, m, cat);
}

// slot[off[s]] for s = 0..k (the voxel count in front of every cloud of a joint sort) into k + 1 consecutive ints: one read-back
__global__ void k_multi_bounds(const int* __restrict__ slot, VoxelMulti m, int* __restrict__ out)
{
    if ((int)threadIdx.x <= m.k) out[threadIdx.x] = slot[m.off[threadIdx.x]];
}
void launch_multi_bounds(const int* slot, const VoxelMulti& m, int* out, hipStream_t st) { k_multi_bounds<<<1, 64, 0, st>>>(slot, m, out); }

// the joint output cloud handed out to its K destinations: voxel v of cloud s (vo[s] <= v < vo[s + 1]) -> out[s][v - vo[s]]
__global__ __launch_bounds__(256) void k_hand_out(const float4* __restrict__ src, VoxelHandOut h)
{
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= h.vo[h.k]) return;
    int s = 0;
#pragma unroll
    for (int j = 1; j < kVoxelMultiMax; ++j) if (j < h.k && v >= h.vo[j]) s = j;
    h.out[s][v - h.vo[s]] = src[v];
}
void launch_hand_out(const float4* src, const VoxelHandOut& h, hipStream_t st)
{
    if (h.vo[h.k] > 0) k_hand_out<<<(h.vo[h.k] + 255) / 256, 256, 0, st>>>(src, h);
}

void launch_count_jumps(const BlockDesc* blocks, int n_blocks, const Segment* segs, float thr, int* jumps, hipStream_t st)
{
    (void)hipMemsetAsync(jumps, 0, sizeof(int), st);
    if (n_blocks > 0) k_count_jumps<<<n_blocks, kBlockQ, 0, st>>>(blocks, segs, thr * thr, jumps);
}

void launch_exclusive_scan(const int* in, int* out, int* tmp, int n, hipStream_t st) { exclusive_scan(in, out, tmp, n, st); }

void launch_pack_cloud(const void* raw_dev, size_t n, int stride, int has_label, float4* out, hipStream_t st)
{
    if (n > 0) k_pack_cloud<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(static_cast<const unsigned char*>(raw_dev), n, stride, has_label, out);
}

void launch_transform_cloud(const float4* in, int n, const float* M12_dev, float4* out, hipStream_t st)
{
    if (n > 0) k_transform_cloud<<<(n + 255) / 256, 256, 0, st>>>(in, n, M12_dev, out);
}
void launch_transform_cloud_m(const float4* in, int n, const float M12_host[12], float4* out, hipStream_t st)
{
    Mat12 M;
    for (int i = 0; i < 12; ++i) M.m[i] = M12_host[i];
    if (n > 0) k_transform_cloud_m<<<(n + 255) / 256, 256, 0, st>>>(in, n, M, out);
}

}  // namespace lisreg
