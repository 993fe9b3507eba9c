// lisreg_features.hip — SURVEY.md §8 f-2: the producer of cloud_info.
//
// Replaces LaserProcessing::projectPointCloud / cloudExtraction / calculateSmoothness / markOccludedPoints /
// extractFeatures (/root/reference/src/core/laserProcessing.cpp:467-510, 515-539, 544-563, 568-605, 610-713) for one
// LiDAR sweep, without the IMU de-skew (deskewPoint is the identity when no IMU data is available, :404-406).
//
// gfx950 mapping
//   * projection: one thread per input point, coalesced 16-B reads; "the first point to land in a pixel wins"
//     (:499 `if (rangeMat != FLT_MAX) continue`) becomes atomicMin on the INPUT INDEX per pixel — deterministic;
//   * extraction: the row-major `count++` loop is a flag + exclusive scan over the H x W image;
//   * smoothness (10-tap range stencil) and occlusion marks are flat streaming kernels (the marks only ever store 1,
//     so they are order-independent);
//   * selection: the greedy pick with +-5 suppression is sequential inside a ring (suppression spills into the next
//     sector) but rings are independent: one wave per ring; the ring's picked / curvature / column arrays live in LDS,
//     every sector is bitonic-sorted by (curvature, index) in LDS (std::sort is unstable: ties fixed by index); the
//     greedy passes run as ballot + ffs loops over 64 sorted candidates at a time, so only actual picks cost serial
//     time; picks go to per-ring lists that a last kernel concatenates in ring order.
// Everything is integer/byte work plus one float stencil: HBM-streaming passes, nothing GEMM-shaped.
// Defined behaviour where the reference has none (oracle/lisreg_oracle.h): per-frame arrays start at zero, the +-5
// neighbour accesses are bounds-checked against the extracted cloud.
#include "lisreg_internal.hpp"

#include <algorithm>

namespace lisreg {

namespace {

constexpr int kMaxRingPts = 4096 + 16;
constexpr int kListCap = 128;         // per ring: <= 6*20 corners, <= 6*4 sharp corners, <= 6*10 sharp surfaces
constexpr int kEmpty = 0x7f7f7f7f;   // pixel owner after a byte-wise 0x7f fill: larger than any input index

__global__ __launch_bounds__(256) void k_feat_project(const float4* __restrict__ pts, const uint32_t* __restrict__ rings,
                                                      int n, lisreg_feature_params P, int* __restrict__ owner)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 p = pts[i];
    const float range = sqrtf(p.x * p.x + p.y * p.y + p.z * p.z);                    // pointDistance (common.h:110-113)
    if (range < P.min_range || range > P.max_range) return;
    const int row = (int)((rings ? rings[i] : __float_as_uint(p.w)) & 0xffffu);
    if (row >= P.n_scan) return;
    if (row % P.downsample_rate != 0) return;
    // :489-494 — float atan2, then double arithmetic, round(), int conversion
    // float atan2 defined as the correctly rounded value (double atan2 rounded to float): libm float atan2 differs
    // between implementations in the last ulp, enough to move a point across a column boundary (see the oracle)
    const float at = (float)atan2((double)p.x, (double)p.y);
    const float horizonAngle = (float)((double)(at * 180) / 3.14159265358979323846);
    const float ang_res_x = (float)(360.0 / (double)(float)P.horizon_scan);
    int col = (int)(-round(((double)horizonAngle - 90.0) / (double)ang_res_x) + (double)(P.horizon_scan / 2));
    if (col >= P.horizon_scan) col -= P.horizon_scan;
    if (col < 0 || col >= P.horizon_scan) return;
    atomicMin(&owner[row * P.horizon_scan + col], i);                                // first point in input order wins
}

__global__ __launch_bounds__(256) void k_feat_valid(const int* __restrict__ owner, int hw, int* __restrict__ flag)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < hw) flag[i] = owner[i] != kEmpty ? 1 : 0;
}

// cloudExtraction (:515-539) + array initialisation
__global__ __launch_bounds__(256) void k_feat_extract(const float4* __restrict__ pts, const int* __restrict__ owner,
                                                      const int* __restrict__ pos, int H, int W, FeatureBuffers fb)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int hw = H * W;
    if (i < hw + 16) { fb.curv[i] = 0.f; fb.picked[i] = 0; fb.label[i] = 0; }
    if (i == 0) fb.counts[0] = pos[hw];
    if (i >= hw) return;
    const int o = owner[i];
    if (o == kEmpty) return;
    const int e = pos[i];
    const float4 p = pts[o];
    fb.col[e] = i % W;
    fb.range[e] = sqrtf(p.x * p.x + p.y * p.y + p.z * p.z);
    fb.src[e] = o;
}

// Batched extraction stacks S sweeps into one range image of S x H rows.  The reference's flat loops over the extracted cloud
// (smoothness, occlusion) stop 5 / 6 entries short of the cloud's two ends: in the stack those ends are the sweep's own
// [lo, hi) — the extracted positions of its first pixel and of the next sweep's first pixel.
__device__ __forceinline__ void sweep_range(const int* __restrict__ pos, int hw_sweep, int n_sweeps, int i, int& lo, int& hi)
{
    int a = 0, b = n_sweeps - 1;                                     // last sweep whose first extracted position is <= i
    while (a < b) { const int mid = (a + b + 1) >> 1; if (pos[(size_t)mid * hw_sweep] <= i) a = mid; else b = mid - 1; }
    lo = pos[(size_t)a * hw_sweep]; hi = pos[(size_t)(a + 1) * hw_sweep];
}

// calculateSmoothness (:544-563)
__global__ __launch_bounds__(256) void k_feat_smooth(const int* __restrict__ counts, const float* __restrict__ r,
                                                     float* __restrict__ curv, const int* __restrict__ pos, int hw_sweep, int n_sweeps)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= counts[0]) return;
    int lo, size;
    sweep_range(pos, hw_sweep, n_sweeps, i, lo, size);
    if (i < lo + 5 || i >= size - 5) return;
    const float d = r[i - 5] + r[i - 4] + r[i - 3] + r[i - 2] + r[i - 1] - r[i] * 10 + r[i + 1] + r[i + 2] + r[i + 3] +
                    r[i + 4] + r[i + 5];
    curv[i] = d * d;
}

// markOccludedPoints (:568-605): only ever stores 1 -> order-independent
__global__ __launch_bounds__(256) void k_feat_occlude(const int* __restrict__ counts, const float* __restrict__ r,
                                                      const int* __restrict__ col, int* __restrict__ picked,
                                                      const int* __restrict__ pos, int hw_sweep, int n_sweeps)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= counts[0]) return;
    int lo, size;
    sweep_range(pos, hw_sweep, n_sweeps, i, lo, size);
    if (i < lo + 5 || i >= size - 6) return;
    const float depth1 = r[i], depth2 = r[i + 1];
    const int columnDiff = abs(col[i + 1] - col[i]);
    if (columnDiff < 10) {
        if ((double)(depth1 - depth2) > 0.3) { for (int l = -5; l <= 0; ++l) picked[i + l] = 1; }
        else if ((double)(depth2 - depth1) > 0.3) { for (int l = 1; l <= 6; ++l) picked[i + l] = 1; }
    }
    const float diff1 = fabsf(r[i - 1] - r[i]), diff2 = fabsf(r[i + 1] - r[i]);
    if ((double)diff1 > 0.02 * (double)r[i] && (double)diff2 > 0.02 * (double)r[i]) picked[i] = 1;
}

// extractFeatures (:610-713): one wave per ring
__global__ __launch_bounds__(256) void k_feat_select(const int* __restrict__ pos, int H, int W, lisreg_feature_params P,
                                                    FeatureBuffers fb, int rows_per_sweep, int xp_stop)
{
    __shared__ int   s_picked[kMaxRingPts];
    __shared__ float s_curv[kMaxRingPts];
    __shared__ int   s_col[kMaxRingPts];
    __shared__ unsigned long long s_key[kMaxRingPts];   // (curvature bits << 32 | position): the sort key of every point of the window
    __shared__ int   s_sorted_all[kMaxRingPts];         // per sector, at the sector's own positions: its points ascending by (curvature, index)
    __shared__ unsigned char s_run[kMaxRingPts];     // per position: how far a pick there suppresses (forward | backward << 4), see below

    const int ring = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;   // 4 waves sort, wave 0 picks
    // the extracted cloud this ring belongs to: [base, size) — the whole cloud for one sweep, the sweep's own slice in a batch
    const int sweep = ring / rows_per_sweep;
    const int base = pos[(size_t)sweep * rows_per_sweep * W], size = pos[(size_t)(sweep + 1) * rows_per_sweep * W];
    const int r0 = pos[ring * W], r1 = pos[(ring + 1) * W];           // this ring's extracted range [r0, r1)
    const int startRing = r0 - 1 + 5, endRing = r1 - 1 - 5;           // startRingIndex / endRingIndex (:521, :537)
    int* lists = fb.ring_lists + (size_t)ring * 3 * kListCap;
    int n_corner = 0, n_csharp = 0, n_ssharp = 0;
    // window [lo, hi) of the extracted arrays mirrored in LDS: the ring +-6 (suppression reaches 5 beyond a pick)
    const int lo = max(r0 - 6, base), hi = min(r1 + 6, size);
    for (int k = lo + tid; k < hi; k += 256) { s_picked[k - lo] = fb.picked[k]; s_curv[k - lo] = fb.curv[k]; s_col[k - lo] = fb.col[k]; }
    __syncthreads();
    // How far a pick at position c suppresses (:647-659, :681-693): forward over c + 1 .. c + 5 and backward over c - 1 .. c - 5, each direction
    // stopping at the cloud's end or at the first pair of neighbours more than 10 columns apart.  Both directions test the same pair
    // predicate pair(a) = "a - 1 and a are both in the cloud and at most 10 columns apart": forward = the run of pair(c + 1), pair(c + 2), ...,
    // backward = the run of pair(c), pair(c - 1), ....  The runs depend on the columns only, so they are formed here once, by all four
    // wavefronts, instead of by ten lanes and an LDS round trip inside every pick of the sequential pass below.
    for (int a = lo + tid; a < hi; a += 256) {
        auto pair_ok = [&](int x) { return x - 1 >= lo && x < hi && x < size && x - 1 >= base && abs(s_col[x - lo] - s_col[x - 1 - lo]) <= 10; };
        int nf = 0, nb = 0;
        while (nf < 5 && pair_ok(a + nf + 1)) ++nf;
        while (nb < 5 && pair_ok(a - nb)) ++nb;
        s_run[a - lo] = (unsigned char)(nf | (nb << 4));
    }
    if (xp_stop == 1) return;                                          // timing experiments only (LISREG_XP_FEAT_STOP): wrong results

    // std::sort of every sector's curvatures (:620), ascending by (curvature, index) — by RANK COUNTING (round 5): a curvature is a square,
    // so its float bits order like the value, and (bits << 32 | position) is one 64-bit key per point; a point's place in its sector's
    // sorted order is the number of keys of the sector below its own.  1 800 points x 300 compares, one barrier, against the 45
    // barrier-separated stages of the bitonic network this replaces; the same order, since the keys
    // are distinct.  (A NaN curvature — there is none: ranges are finite — would sort behind every number.)
    for (int k = lo + tid; k < hi; k += 256)
        s_key[k - lo] = ((unsigned long long)__float_as_uint(s_curv[k - lo]) << 32) | (unsigned)k;
    __syncthreads();
    {
        int sps[6], ms[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            sps[j] = (startRing * (6 - j) + endRing * j) / 6;
            ms[j] = max((startRing * (5 - j) + endRing * (j + 1)) / 6 - 1 - sps[j], 0);      // std::sort range [sp, ep)
        }
        // 42 threads per sector, each with up to eight of its sector's points in registers per pass: one LDS read per compared key serves eight
        // counts (the threads of a sector read the same address: a broadcast)
        const int j = tid / 42, r = tid - j * 42;
        if (j < 6) {
            int sp = sps[0], m = ms[0];
#pragma unroll
            for (int jj = 1; jj < 6; ++jj) if (j == jj) { sp = sps[jj]; m = ms[jj]; }
            const unsigned long long* keys = s_key + (sp - lo);
            for (int t0 = r; t0 < m; t0 += 8 * 42) {
                unsigned long long mine[8];
                int below[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) { const int t = t0 + 42 * q; mine[q] = t < m ? keys[t] : 0ull; below[q] = 0; }
                for (int u = 0; u < m; ++u) {
                    const unsigned long long ku = keys[u];
#pragma unroll
                    for (int q = 0; q < 8; ++q) below[q] += ku < mine[q] ? 1 : 0;
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) { const int t = t0 + 42 * q; if (t < m) s_sorted_all[sp - lo + below[q]] = sp + t; }
            }
        }
    }
    __syncthreads();
    if (wave != 0 || xp_stop == 2) return;                             // the picking below is one wavefront's work; no barrier follows

    for (int j = 0; j < 6; ++j) {
        const int sp = (startRing * (6 - j) + endRing * j) / 6;
        const int ep = (startRing * (5 - j) + endRing * (j + 1)) / 6 - 1;
        if (sp >= ep) continue;                                        // wave-uniform
        const int* s_sorted = s_sorted_all + (sp - lo);                 // this sector's indices, ascending by (curvature, index)
        // The greedy passes are sequential in the SORTED order, but a candidate only costs time when it is picked:
        // every lane holds one sorted candidate, ballot + ffs finds the next one that qualifies and is still unpicked,
        // that lane marks itself and its +-5 neighbours in LDS, and candidates suppressed meanwhile are skipped for
        // free (cloudNeighborPicked only ever goes 0 -> 1, so "not eligible when reached" == "never eligible").
        if (wave == 0) {
// One batch = 64 candidates in sorted order, one per lane.  A lane reads its candidate's picked flag and suppression runs ONCE per batch; a
// pick then costs a ballot, three lane reads and a range test — every lane whose candidate lies inside the pick's run marks itself in a
// register, the run's positions are marked in LDS for the batches and sectors still to come (nothing in the loop waits for LDS).
#define LISREG_PICK_MARK(c_, run_) do { \
            const int nf_ = (run_) & 15, nb_ = (run_) >> 4; \
            if (ind >= (c_) - nb_ && ind <= (c_) + nf_) mypicked = true; \
            if (lane <= nf_ + nb_) s_picked[(c_) - nb_ + lane - lo] = 1; } while (0)
        {   // edge features: largest curvature first (:626-661), at most 20 per sector, the first 4 are "sharp"
            int largest = 0;
            bool stop = false;
            for (int kb = ep; kb >= sp && !stop; kb -= 64) {
                const int k = kb - lane;
                const bool inr = k >= sp;
                const int ind = inr ? ((k == ep) ? ep : s_sorted[k - sp]) : lo;   // element ep lies outside the sorted range
                const bool stat = inr && s_curv[ind - lo] > P.edge_threshold;
                bool mypicked = !stat || s_picked[ind - lo] != 0;
                const int myrun = s_run[ind - lo];
                unsigned long long todo = __ballot(stat);
                while (todo) {
                    const unsigned long long m = __ballot(!mypicked) & todo;
                    if (!m) break;
                    const int f = __ffsll((long long)m) - 1;
                    todo &= ~((2ull << f) - 1ull);                             // lanes up to f have had their turn
                    largest++;
                    if (largest > 20) { stop = true; break; }                  // `else break` at :641
                    if (lane == f) {
                        fb.label[ind] = 1;
                        lists[0 * kListCap + n_corner] = ind;
                        if (largest <= 4) lists[1 * kListCap + n_csharp] = ind;
                    }
                    const int c = __builtin_amdgcn_readlane(ind, f), run = __builtin_amdgcn_readlane(myrun, f);
                    LISREG_PICK_MARK(c, run);
                    n_corner++;
                    if (largest <= 4) n_csharp++;
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        {   // planar features: smallest curvature first (:663-695), the first 10 per sector are "sharp"
            int largest = 0;
            for (int kb = sp; kb <= ep; kb += 64) {
                const int k = kb + lane;
                const bool inr = k <= ep;
                const int ind = inr ? ((k == ep) ? ep : s_sorted[k - sp]) : lo;
                const bool stat = inr && s_curv[ind - lo] < P.surf_threshold;
                bool mypicked = !stat || s_picked[ind - lo] != 0;
                const int myrun = s_run[ind - lo];
                unsigned long long todo = __ballot(stat);
                while (todo) {
                    const unsigned long long m = __ballot(!mypicked) & todo;
                    if (!m) break;
                    const int f = __ffsll((long long)m) - 1;
                    todo &= ~((2ull << f) - 1ull);
                    largest++;
                    if (lane == f) {
                        fb.label[ind] = -1;
                        if (largest <= 10) lists[2 * kListCap + n_ssharp] = ind;
                    }
                    const int c = __builtin_amdgcn_readlane(ind, f), run = __builtin_amdgcn_readlane(myrun, f);
                    LISREG_PICK_MARK(c, run);
                    if (largest <= 10) n_ssharp++;
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
#undef LISREG_PICK_MARK
        }   // wave 0
    }
    if (tid == 0) {
        int* c = fb.ring_counts + ring * 4;
        c[0] = n_corner; c[1] = n_csharp; c[2] = n_ssharp;
    }
}

// surfaceCloud (:697-704): every k inside a processed sector with cloudLabel <= 0, in ascending k
__global__ __launch_bounds__(256) void k_feat_surface_flags(const int* __restrict__ pos, int H, int W, FeatureBuffers fb)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    const int hw = H * W;
    if (k >= hw + 16) return;
    int f = 0;
    if (k < fb.counts[0]) {
        int lo = 0, hi = H - 1;                                        // ring of k: last ring with pos[ring*W] <= k
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (pos[mid * W] <= k) lo = mid; else hi = mid - 1; }
        const int r0 = pos[lo * W], r1 = pos[(lo + 1) * W];
        const int startRing = r0 - 1 + 5, endRing = r1 - 1 - 5;
        for (int j = 0; j < 6; ++j) {
            const int sp = (startRing * (6 - j) + endRing * j) / 6;
            const int ep = (startRing * (5 - j) + endRing * (j + 1)) / 6 - 1;
            if (sp < ep && k >= sp && k <= ep) f = 1;
        }
        if (fb.label[k] > 0) f = 0;
    }
    fb.flag[k] = f;
}

__global__ __launch_bounds__(256) void k_feat_surface_write(int hw, const int* __restrict__ spos, FeatureBuffers fb)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k == 0) fb.counts[2] = spos[hw + 16];
    if (k >= hw + 16 || !fb.flag[k]) return;
    fb.lists[(size_t)1 * (hw + 16) + spos[k]] = fb.src[k];
}

// concatenate the per-ring pick lists in ring order: exclusive scan of the per-ring counts (one wave; ring_counts[r][3]
// is reused for nothing else, so the offsets go to a small LDS-free strided loop), then one workgroup per ring copies.
__global__ __launch_bounds__(256) void k_feat_offsets(int H, FeatureBuffers fb, int* __restrict__ ring_off /* [H + 1][3] */)
{
    // exclusive scan of the three per-ring counts over H rings (H up to tens of thousands in a batch): 256 threads, each owning a
    // contiguous chunk of rings; chunk sums are scanned through LDS
    __shared__ int s_sum[3][256];
    const int t = threadIdx.x;
    const int per = (H + 255) / 256, r_begin = t * per, r_end = min(H, r_begin + per);
    int acc[3] = { 0, 0, 0 };
    for (int ring = r_begin; ring < r_end; ++ring)
        for (int w = 0; w < 3; ++w) acc[w] += fb.ring_counts[ring * 4 + w];
    for (int w = 0; w < 3; ++w) s_sum[w][t] = acc[w];
    __syncthreads();
    if (t < 3) { int run = 0; for (int k = 0; k < 256; ++k) { const int v = s_sum[t][k]; s_sum[t][k] = run; run += v; }
                 const int slot[3] = { 1, 3, 4 };                      // counts[]: corner, corner_sharp, surface_sharp
                 fb.counts[slot[t]] = run; ring_off[H * 3 + t] = run; }
    __syncthreads();
    int run[3] = { s_sum[0][t], s_sum[1][t], s_sum[2][t] };
    for (int ring = r_begin; ring < r_end; ++ring)
        for (int w = 0; w < 3; ++w) { ring_off[ring * 3 + w] = run[w]; run[w] += fb.ring_counts[ring * 4 + w]; }
}

__global__ __launch_bounds__(128) void k_feat_concat(int hw, FeatureBuffers fb, const int* __restrict__ ring_off)
{
    const int ring = blockIdx.x;
    const int list_of[3] = { 0, 2, 3 };                                // lists[]: corner, corner_sharp, surface_sharp
    for (int w = 0; w < 3; ++w) {
        const int cnt = fb.ring_counts[ring * 4 + w], off = ring_off[ring * 3 + w];
        for (int t = threadIdx.x; t < cnt; t += 128)
            fb.lists[(size_t)list_of[w] * (hw + 16) + off + t] = fb.src[fb.ring_lists[((size_t)ring * 3 + w) * kListCap + t]];
    }
}

// ---- batched extraction: S sweeps as one stack of S x H rows ------------------------------------------------------------
struct SweepOffsets { int off[258]; };          // first point of each sweep in the concatenated cloud; off[S] = total

// row of every point in the stack (its ring + sweep * H), or 0xffff for points the single-sweep kernel would drop by ring
__global__ __launch_bounds__(256) void k_feat_batch_rows(const float4* __restrict__ cat, int n, SweepOffsets so, int n_sweeps, int H,
                                                         int rate, uint32_t* __restrict__ rows)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int a = 0, b = n_sweeps - 1;
    while (a < b) { const int mid = (a + b + 1) >> 1; if (so.off[mid] <= i) a = mid; else b = mid - 1; }
    const int ring = (int)(__float_as_uint(cat[i].w) & 0xffffu);
    rows[i] = (ring < H && ring % rate == 0) ? (uint32_t)(a * H + ring) : 0xffffu;
}

// per sweep: where its slice of each output list begins — B[s] = { extracted, corner, corner_sharp, surface_sharp, surface }
__global__ void k_feat_batch_bounds(int n_sweeps, int H, int hw_sweep, const int* __restrict__ pos, const int* __restrict__ spos,
                                    const int* __restrict__ ring_off, int* __restrict__ B)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s > n_sweeps) return;
    const int e = pos[(size_t)s * hw_sweep];
    B[s * 5 + 0] = e;
    for (int w = 0; w < 3; ++w) B[s * 5 + 1 + w] = ring_off[(size_t)s * H * 3 + w];
    B[s * 5 + 4] = spos[e];
}

// one launch hands every sweep its slice of one list: grid.y = sweep
struct GatherJob { float4* dst; int begin, count; };
__global__ __launch_bounds__(256) void k_feat_batch_gather(const float4* __restrict__ cat, const int* __restrict__ idx,
                                                           const GatherJob* __restrict__ jobs)
{
    const GatherJob j = jobs[blockIdx.y];
    for (int t = blockIdx.x * 256 + threadIdx.x; t < j.count; t += gridDim.x * 256) j.dst[t] = cat[idx[j.begin + t]];
}

// categoryMapping (/root/reference/src/node/semanticFusionNode.cpp:173-189): flag of "this point belongs to class k"
struct LabelMap { uint32_t m[32]; };
// categoryMapping as ONE stable five-way partition: the class flags of all points laid out class-major ([5][n]), one scan over the 5 n
// flags — the position of a flag IS the point's place in "class 0 in input order, then class 1, ..." — and one kernel that turns it into the
// five index lists and their sizes
__global__ __launch_bounds__(256) void k_sem_flags(const float4* __restrict__ pts, const uint32_t* __restrict__ labels, int n,
                                                   LabelMap map, int* __restrict__ flag)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t u = map.m[(labels ? labels[i] : __float_as_uint(pts[i].w)) & 31u];
    const int c = u == 10u ? 0 : (u == 40u ? 1 : (u == 50u ? 2 : (u == 81u ? 3 : 4)));
#pragma unroll
    for (int k = 0; k < 5; ++k) flag[(size_t)k * n + i] = c == k ? 1 : 0;
}

__global__ __launch_bounds__(256) void k_sem_write(int n, const int* __restrict__ flag, const int* __restrict__ pos,
                                                   int* __restrict__ idx_out, int* __restrict__ count_out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < 5) count_out[i] = pos[(size_t)(i + 1) * n] - pos[(size_t)i * n];
    if (i >= n) return;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const size_t f = (size_t)k * n + i;
        if (flag[f]) idx_out[(size_t)k * n + (pos[f] - pos[(size_t)k * n])] = i;
    }
}

// the five class clouds written by one launch (grid.y = class): out[k][j] = pts[idx[k * n + j]]
__global__ __launch_bounds__(256) void k_sem_gather(const float4* __restrict__ pts, const int* __restrict__ idx, int n, SemanticGather g)
{
    const int k = blockIdx.y;
    for (int j = blockIdx.x * 256 + threadIdx.x; j < g.count[k]; j += gridDim.x * 256) g.out[k][j] = pts[idx[(size_t)k * n + j]];
}

__global__ __launch_bounds__(256) void k_gather_points(const float4* __restrict__ pts, const int* __restrict__ idx, int n,
                                                       float4* __restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = pts[idx[i]];
}

// ---- IMU de-skew (deskewPoint / findRotation, laserProcessing.cpp:368-399, 427-462) ----------------------------------------
__global__ __launch_bounds__(256) void k_feat_first(const int* __restrict__ owner, int hw, int* __restrict__ first)
{
    __shared__ int s_min[4];
    int m = kEmpty;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < hw; i += gridDim.x * 256) m = min(m, owner[i]);
    for (int o = 32; o > 0; o >>= 1) m = min(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) s_min[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomicMin(first, min(min(s_min[0], s_min[1]), min(s_min[2], s_min[3])));
}

__device__ __forceinline__ void find_rotation(const DeskewTables& T, double pointTime, float rot[3])
{
    int front = 0;
    while (front < T.imu_pointer_cur) { if (pointTime < T.time[front]) break; ++front; }
    if (pointTime > T.time[front] || front == 0) { rot[0] = (float)T.rx[front]; rot[1] = (float)T.ry[front]; rot[2] = (float)T.rz[front]; }
    else {
        const int back = front - 1;
        const double rf = (pointTime - T.time[back]) / (T.time[front] - T.time[back]);
        const double rb = (T.time[front] - pointTime) / (T.time[front] - T.time[back]);
        rot[0] = (float)(T.rx[front] * rf + T.rx[back] * rb);
        rot[1] = (float)(T.ry[front] * rf + T.ry[back] * rb);
        rot[2] = (float)(T.rz[front] * rf + T.rz[back] * rb);
    }
}

// pcl::getTransformation(0, 0, 0, roll, pitch, yaw).linear(); cos / sin = the correctly rounded float (double evaluation)
__device__ __forceinline__ void rot_from_rpy(float roll, float pitch, float yaw, float R[9])
{
    const float A = (float)cos((double)yaw), B = (float)sin((double)yaw), C = (float)cos((double)pitch), D = (float)sin((double)pitch),
                E = (float)cos((double)roll), F = (float)sin((double)roll), DE = D * E, DF = D * F;
    R[0] = A * C; R[1] = A * DF - B * E; R[2] = B * F + A * DE;
    R[3] = B * C; R[4] = A * E + B * DF; R[5] = B * DE - A * F;
    R[6] = -D;    R[7] = C * F;          R[8] = C * E;
}

// transStartInverse from the first pixel-owning point (Eigen's cofactor inverse of the 3x3 linear part)
__global__ void k_feat_start_inverse(const float* __restrict__ times, const int* __restrict__ first, DeskewTables T, float* __restrict__ Rsi)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float m[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 }, inv[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
    const int f = *first;
    if (f != kEmpty) {
        float rot[3];
        find_rotation(T, T.time_scan_cur + (double)times[f], rot);
        rot_from_rpy(rot[0], rot[1], rot[2], m);
#define COF(i, j) (m[3 * (((i) + 1) % 3) + ((j) + 1) % 3] * m[3 * (((i) + 2) % 3) + ((j) + 2) % 3] - m[3 * (((i) + 1) % 3) + ((j) + 2) % 3] * m[3 * (((i) + 2) % 3) + ((j) + 1) % 3])
        const float c00 = COF(0, 0), c10 = COF(1, 0), c20 = COF(2, 0);
        const float det = (c00 * m[0] + c10 * m[3]) + c20 * m[6];
        const float invdet = 1.0f / det;
        inv[0] = c00 * invdet; inv[1] = c10 * invdet; inv[2] = c20 * invdet;
        inv[3] = COF(0, 1) * invdet; inv[4] = COF(1, 1) * invdet; inv[5] = COF(2, 1) * invdet;
        inv[6] = COF(0, 2) * invdet; inv[7] = COF(1, 2) * invdet; inv[8] = COF(2, 2) * invdet;
#undef COF
    }
    for (int k = 0; k < 9; ++k) Rsi[k] = inv[k];
}

// one thread per range-image pixel: rotate its owner into the frame of the first point (coordinates only; payload kept)
__global__ __launch_bounds__(256) void k_feat_deskew(const int* __restrict__ owner, int hw, const float* __restrict__ times,
                                                     DeskewTables T, const float* __restrict__ Rsi, float4* __restrict__ pts)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= hw) return;
    const int o = owner[i];
    if (o == kEmpty) return;
    float rot[3], Rf[9], Rb[9];
    find_rotation(T, T.time_scan_cur + (double)times[o], rot);
    rot_from_rpy(rot[0], rot[1], rot[2], Rf);
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) Rb[3 * a + b] = (Rsi[3 * a] * Rf[b] + Rsi[3 * a + 1] * Rf[3 + b]) + Rsi[3 * a + 2] * Rf[6 + b];
    const float4 p = pts[o];
    float4 q = p;
    q.x = ((Rb[0] * p.x + Rb[1] * p.y) + Rb[2] * p.z) + 0.0f;
    q.y = ((Rb[3] * p.x + Rb[4] * p.y) + Rb[5] * p.z) + 0.0f;
    q.z = ((Rb[6] * p.x + Rb[7] * p.y) + Rb[8] * p.z) + 0.0f;
    pts[o] = q;
}

}  // namespace

void launch_deskew(const int* owner, int hw, const float* times_dev, DeskewTables T, int* first_dev, float* rsi_dev, float4* pts_copy,
                   hipStream_t st)
{
    (void)hipMemsetAsync(first_dev, 0x7f, sizeof(int), st);
    k_feat_first<<<std::min((hw + 255) / 256, 256), 256, 0, st>>>(owner, hw, first_dev);
    k_feat_start_inverse<<<1, 64, 0, st>>>(times_dev, first_dev, T, rsi_dev);
    k_feat_deskew<<<(hw + 255) / 256, 256, 0, st>>>(owner, hw, times_dev, T, rsi_dev, pts_copy);
}

void launch_extract_features(const float4* pts, const uint32_t* rings, int n, lisreg_feature_params P, FeatureBuffers fb,
                             hipStream_t st, int n_sweeps)
{
    // n_sweeps > 1: P.n_scan is the height of the whole stack (n_sweeps x rings per sweep), `rings` holds stack rows
    const int H = P.n_scan, W = P.horizon_scan, hw = H * W;
    const int rows_per_sweep = H / n_sweeps, hw_sweep = rows_per_sweep * W;
    (void)hipMemsetAsync(fb.owner, 0x7f, sizeof(int) * (size_t)hw, st);             // every pixel = kEmpty
    if (n > 0) k_feat_project<<<(n + 255) / 256, 256, 0, st>>>(pts, rings, n, P, fb.owner);
    k_feat_valid<<<(hw + 255) / 256, 256, 0, st>>>(fb.owner, hw, fb.flag);
    launch_exclusive_scan(fb.flag, fb.pos, fb.scan_tmp, hw, st);
    k_feat_extract<<<(hw + 16 + 255) / 256, 256, 0, st>>>(pts, fb.owner, fb.pos, H, W, fb);
    k_feat_smooth<<<(hw + 255) / 256, 256, 0, st>>>(fb.counts, fb.range, fb.curv, fb.pos, hw_sweep, n_sweeps);
    k_feat_occlude<<<(hw + 255) / 256, 256, 0, st>>>(fb.counts, fb.range, fb.col, fb.picked, fb.pos, hw_sweep, n_sweeps);
#ifdef LISREG_XP_HOOKS      /* timing experiment (wrong features), compiled in only on request (csrc/Makefile: XP=1) */
    static const int xp_stop = getenv("LISREG_XP_FEAT_STOP") ? atoi(getenv("LISREG_XP_FEAT_STOP")) : 0;
    if (xp_stop) fprintf(stderr, "[lisreg] LISREG_XP_FEAT_STOP is set: feature selection stops early (timing experiment)\n");
#else
    const int xp_stop = 0;
#endif
    k_feat_select<<<H, 256, 0, st>>>(fb.pos, H, W, P, fb, rows_per_sweep, xp_stop);   // grid = sweeps x rings
    k_feat_surface_flags<<<(hw + 16 + 255) / 256, 256, 0, st>>>(fb.pos, H, W, fb);
    // the second scan goes to the upper half of `pos`; the lower half (ring boundaries) stays valid
    launch_exclusive_scan(fb.flag, fb.pos + (hw + 17), fb.scan_tmp, hw + 16, st);
    k_feat_surface_write<<<(hw + 16 + 255) / 256, 256, 0, st>>>(hw, fb.pos + (hw + 17), fb);
    int* ring_off = fb.flag;                                            // flag[] is free again after the surface scan
    k_feat_offsets<<<1, 256, 0, st>>>(H, fb, ring_off);
    k_feat_concat<<<H, 128, 0, st>>>(hw, fb, ring_off);
}

void launch_feature_batch_rows(const float4* cat, int n, const int* offsets /* host [n_sweeps + 1] */, int n_sweeps, int H, int rate,
                               uint32_t* rows, hipStream_t st)
{
    SweepOffsets so;
    for (int s = 0; s <= n_sweeps; ++s) so.off[s] = offsets[s];
    if (n > 0) k_feat_batch_rows<<<(n + 255) / 256, 256, 0, st>>>(cat, n, so, n_sweeps, H, rate, rows);
}

void launch_feature_batch_bounds(int n_sweeps, int H, int hw_sweep, FeatureBuffers fb, int hw_total, int* B, hipStream_t st)
{
    // ring offsets were left in fb.flag by launch_extract_features; the surface scan in the upper half of fb.pos
    k_feat_batch_bounds<<<(n_sweeps + 1 + 63) / 64, 64, 0, st>>>(n_sweeps, H, hw_sweep, fb.pos, fb.pos + (hw_total + 17), fb.flag, B);
}

void launch_feature_batch_gather(const float4* cat, const int* idx, const void* jobs_dev, int n_sweeps, int max_count, hipStream_t st)
{
    if (max_count <= 0) return;
    const int gx = std::min(64, (max_count + 255) / 256);
    k_feat_batch_gather<<<dim3(gx, n_sweeps), 256, 0, st>>>(cat, idx, static_cast<const GatherJob*>(jobs_dev));
}

// stable five-way partition: idx_out[k*n ..] = input indices of class k in input order, counts[k] = its size; flag and pos hold 5 n + 1 ints
void launch_semantic_split(const float4* pts, const uint32_t* labels, int n, const uint32_t map[32], int* flag, int* pos,
                           int* scan_tmp, int* idx_out, int* counts, hipStream_t st)
{
    LabelMap lm;
    for (int i = 0; i < 32; ++i) lm.m[i] = map[i];
    k_sem_flags<<<(n + 255) / 256, 256, 0, st>>>(pts, labels, n, lm, flag);
    launch_exclusive_scan(flag, pos, scan_tmp, 5 * n, st);
    k_sem_write<<<(n + 255) / 256, 256, 0, st>>>(n, flag, pos, idx_out, counts);
}

void launch_semantic_gather(const float4* pts, const int* idx, int n, const SemanticGather& g, hipStream_t st)
{
    int mx = 0;
    for (int k = 0; k < 5; ++k) mx = std::max(mx, g.count[k]);
    if (mx > 0) k_sem_gather<<<dim3((unsigned)std::min(256, (mx + 255) / 256), 5), 256, 0, st>>>(pts, idx, n, g);
}

void launch_gather_points(const float4* pts, const int* idx, int n, float4* out, hipStream_t st)
{
    if (n > 0) k_gather_points<<<(n + 255) / 256, 256, 0, st>>>(pts, idx, n, out);
}

}  // namespace lisreg
