// lisreg_assoc.hip — the hot kernel: fused correspondence search + residual model + normal-equation partials.
//
// One launch = one Gauss-Newton iteration's cornerOptimization + surfOptimization + combineOptimizationCoeffs +
// the Jacobian / AtA / AtB build of LMOptimization for EVERY registration of the batch:
//   /root/reference/src/node/odomEstimationNode.cpp:633-747 (point-to-line), :749-827 (point-to-plane),
//   :829-850 (compaction — fused away: accepted rows go straight into the reduction), :862-920 (J rows, AtA, AtB)
// and the label-weighted copies src/node/subMapOptmizationNode.cpp:1557-1917, 4556-4916.
//
// gfx950 mapping (search front-ends share one residual / reduction tail; `search_mode` selects)
//   * a workgroup = 256 consecutive queries of one (item, stage), kept in caller order (scan / voxel order is
//     spatially coherent, which is what the cell walk's L1 hit rate needs);
//   * pcl::KdTreeFLANN::nearestKSearch(k=5) + `sqDist[4] < tau` is replaced by an exact fixed-radius 5-NN over the
//     uniform grid of lisreg_index.hip, with a sorted top-5 in registers initialised at tau:
//       k_assoc_walk<.., kGraph = false>  every lane walks only the cells that can hold a point closer than its current
//                      5th-best distance, seeded with last iteration's neighbours (any five points bound the radius);
//                      candidates are 16-B records read through L1/L2, four loads in flight per lane;
//       k_assoc_walk<.., kGraph = true>   certified scan of the k-NN graph row of last iteration's nearest neighbour, the cell walk
//                      only without a certificate (big shared-target batches);
//       k_assoc_staged (first version, kept for cross-checks) workgroup bounding box + sqrt(tau), covered cell runs
//                      staged through LDS, all lanes scan via LDS broadcast;
//     all return the same neighbour SETS (tests require identical correspondence counts);
//   * the 5 neighbours are gathered once (5 x 16 B), the 3x3 eigen / 5x3 QR fit, weights and the Jacobian row stay in
//     registers, and the 28 normal-equation scalars are reduced by a wave-level halving butterfly -> LDS -> one partial row
//     per workgroup (fixed order, no float atomics: reproducible).
// No dense contraction exists here (K = n_corr, M = N = 6), so MFMA is not used.  Measured: the
// submap is L2 / Infinity-Cache resident, HBM traffic is below the algorithmic bytes, and the kernel is bound by VALU
// issue at 8 waves/SIMD (<= 64 VGPRs).
#include "lisreg_internal.hpp"

// Two translation units are built from this source (csrc/Makefile):
//   lisreg_assoc.o        the production arithmetic: FMA contraction where the source says a*b+c, 1-ulp hardware reciprocal / square
//                         root in the line and plane fits, `* 0.2f` for `/ 5`, fp32 sums inside a wavefront;
//   lisreg_assoc_exact.o  (-DLISREG_EXACT=1 -ffp-contract=off) the reference's arithmetic operation for operation: no contraction,
//                         IEEE division and square root wherever the reference divides or calls sqrt, cv::eigen's pivoted Jacobi,
//                         fp64 sums throughout.  Selected at run time with lisreg_set_option("exact_arithmetic", 1); it is the parity
//                         anchor (tests require its integer outputs — accept flags, n_corr, iteration counts — to EQUAL the oracle's)
//                         and runs the same search front-ends, so the production build's deviations can be attributed against it.
#ifndef LISREG_EXACT
#define LISREG_EXACT 0
#endif
#if LISREG_EXACT
#define LISREG_ASSOC_NS exact_arith
#define LISREG_LAUNCH_ASSOC launch_assoc_exact
#define LISREG_LAUNCH_TEST_FIT launch_test_fit_exact
#else
#define LISREG_ASSOC_NS fast_arith
#define LISREG_LAUNCH_ASSOC launch_assoc
#define LISREG_LAUNCH_TEST_FIT launch_test_fit
#endif

namespace lisreg {

namespace {
namespace LISREG_ASSOC_NS {

constexpr bool kExactArith = LISREG_EXACT != 0;
#ifndef LISREG_KEEP_SRC
#define LISREG_KEEP_SRC 1        // one-lane-per-query kernels keep the source record in registers across the search (see k_assoc_walk)
#endif
// (Round 5 built three variants of this file that measured no gain and were taken out again — the 28 wave sums on the matrix pipe
// (v_mfma_f32_16x16x4_f32: gfx950's f32 MFMA runs at the f32 vector rate), the heads of the cell rows staged in LDS, the plane fit's pivot
// swaps as selects: profiles/r05_kernel_experiments.md sections 1, 3, 11; the code is profiles/r05_xp_mfma_stage_select.patch.)
#ifndef LISREG_MED3_INSERT
#define LISREG_MED3_INSERT 1
#endif
// LDS of the reduction: one private region per wavefront (kRedWaveFloats floats) inside one array of the kernel.
//   the region starts with the wave's 28 sums (doubles)
constexpr int kRedWaveFloats = 1024;
// the reference's `/` and sqrt(): IEEE in the exact build, one v_rcp_f32 / v_sqrt_f32 (1 ulp) in the production build
__device__ __forceinline__ float fdiv(float a, float b) { return kExactArith ? a / b : a * __builtin_amdgcn_rcpf(b); }
__device__ __forceinline__ float fsqrt(float x) { return kExactArith ? sqrtf(x) : __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float fdiv5(float a) { return kExactArith ? a / 5.f : a * 0.2f; }

__device__ __forceinline__ float hypot_f(float a, float b)
{
    a = fabsf(a); b = fabsf(b);
    if (kExactArith) {                         // cv::eigen's hypot (JacobiImpl_), division and sqrt as written
        if (a > b) { b /= a; return a * sqrtf(1.f + b * b); }
        if (b > 0.f) { a /= b; return b * sqrtf(1.f + a * a); }
        return 0.f;
    }
    const float mx = fmaxf(a, b), mn = fminf(a, b);
    const float r = mn * __builtin_amdgcn_rcpf(mx);          // 1-ulp reciprocal and square root (v_rcp_f32, v_sqrt_f32): the Jacobi
    return mx > 0.f ? mx * __builtin_amdgcn_sqrtf(1.f + r * r) : 0.f;      // sweeps absorb a last-bit difference per rotation
}

// one Jacobi rotation annihilating A[k][l] of a symmetric 3x3 held in registers (cv::eigen's rotation formulas)
#define LISREG_ROT(v0, v1) do { const float a0_ = (v0), b0_ = (v1); (v0) = a0_ * c - b0_ * s; (v1) = a0_ * s + b0_ * c; } while (0)
// rotation parameters for the pivot p = A[k][l], y = (w_l - w_k) / 2: leaves c, s, t (the eigenvalue shift)
#define LISREG_JACOBI_CST(p, y) \
        float t = fabsf(y) + hypot_f((p), (y)); \
        float s = hypot_f((p), t); \
        const float c = fdiv(t, s); \
        s = fdiv((p), s); t = fdiv((p), t) * (p); \
        if ((y) < 0.f) { s = -s; t = -t; }

// Largest eigenvector and the two largest eigenvalues of the symmetric 3x3 {a11..a33}.
//   production build: cyclic Jacobi, at most six sweeps (same eigen-system to a few ulp, no pivot search);
//   exact build: cv::eigen as OpenCV 3.2 runs it on CV_32F (JacobiImpl_): the pivot is the largest off-diagonal element found through
//     the per-row / per-column maximum tables indR / indC, which are refreshed only for the two rotated indices — the stale
//     entries are part of the algorithm and are carried here (for n = 3: indR[0] in {1, 2} and indC[2] in {0, 1}; indR[1] = 2 and
//     indC[1] = 0 always) — at most n * n * 30 rotations, stop at |pivot| <= FLT_EPSILON.
__device__ __forceinline__ void eigen_sym3(float a11, float a12, float a13, float a22, float a23, float a33,
                                           float& l0, float& l1, float v0[3])
{
    float w0 = a11, w1 = a22, w2 = a33;
    float v00 = 1, v01 = 0, v02 = 0, v10 = 0, v11 = 1, v12 = 0, v20 = 0, v21 = 0, v22 = 1;   // rows = vectors
    if (kExactArith) {
        int indR0 = fabsf(a12) < fabsf(a13) ? 2 : 1;
        int indC2 = fabsf(a13) < fabsf(a23) ? 1 : 0;
#pragma unroll 1
        for (int it = 0; it < 270; ++it) {
            // pivot search in cv::eigen's order: rows 0, 1 through indR, then columns 1, 2 through indC; strict '<' keeps the first maximum
            int k = 0, l = indR0;
            float mv = fabsf(indR0 == 1 ? a12 : a13);
            { const float v = fabsf(a23); if (mv < v) { mv = v; k = 1; l = 2; } }
            { const float v = fabsf(a12); if (mv < v) { mv = v; k = 0; l = 1; } }
            { const float v = fabsf(indC2 == 0 ? a13 : a23); if (mv < v) { mv = v; k = indC2; l = 2; } }
            if (k == 0 && l == 1) {
                const float p = a12;
                if (fabsf(p) <= 1.1920929e-7f) break;
                const float y = (w1 - w0) * 0.5f;
                LISREG_JACOBI_CST(p, y)
                a12 = 0.f; w0 -= t; w1 += t;
                LISREG_ROT(a13, a23);
                LISREG_ROT(v00, v10); LISREG_ROT(v01, v11); LISREG_ROT(v02, v12);
                indR0 = fabsf(a12) < fabsf(a13) ? 2 : 1;          // idx = k = 0: indR[0]; idx = l = 1: indR[1] = 2, indC[1] = 0
            } else if (k == 0) {
                const float p = a13;
                if (fabsf(p) <= 1.1920929e-7f) break;
                const float y = (w2 - w0) * 0.5f;
                LISREG_JACOBI_CST(p, y)
                a13 = 0.f; w0 -= t; w2 += t;
                LISREG_ROT(a12, a23);
                LISREG_ROT(v00, v20); LISREG_ROT(v01, v21); LISREG_ROT(v02, v22);
                indR0 = fabsf(a12) < fabsf(a13) ? 2 : 1;          // idx = 0: indR[0]; idx = 2: indC[2]
                indC2 = fabsf(a13) < fabsf(a23) ? 1 : 0;
            } else {
                const float p = a23;
                if (fabsf(p) <= 1.1920929e-7f) break;
                const float y = (w2 - w1) * 0.5f;
                LISREG_JACOBI_CST(p, y)
                a23 = 0.f; w1 -= t; w2 += t;
                LISREG_ROT(a12, a13);
                LISREG_ROT(v10, v20); LISREG_ROT(v11, v21); LISREG_ROT(v12, v22);
                indC2 = fabsf(a13) < fabsf(a23) ? 1 : 0;          // idx = 1: indR[1] = 2, indC[1] = 0; idx = 2: indC[2]; indR[0] goes stale
            }
        }
        // descending selection sort with row swaps (:280-287 of the restatement): only D[0], D[1] and V row 0 are read (:692-702)
        float e0 = w0, e1 = w1, e2 = w2, x = v00, y = v01, z = v02;
        {   // k = 0: m = first index of the maximum under strict '<'
            int m = 0; float wm = e0;
            if (wm < e1) { m = 1; wm = e1; }
            if (wm < e2) { m = 2; wm = e2; }
            if (m == 1) { const float tw = e0; e0 = e1; e1 = tw; x = v10; y = v11; z = v12; }
            else if (m == 2) { const float tw = e0; e0 = e2; e2 = tw; x = v20; y = v21; z = v22; }
        }
        l0 = e0; l1 = (e1 < e2) ? e2 : e1;
        v0[0] = x; v0[1] = y; v0[2] = z;
        return;
    }
#pragma unroll 1
    for (int sweep = 0; sweep < 6; ++sweep) {
        if (fabsf(a12) + fabsf(a13) + fabsf(a23) <= 1e-30f) break;
        {   // (k,l) = (0,1)
            const float p = a12;
            if (fabsf(p) > 0.f) {
                const float y = (w1 - w0) * 0.5f;
                LISREG_JACOBI_CST(p, y)
                a12 = 0.f; w0 -= t; w1 += t;
                LISREG_ROT(a13, a23);
                LISREG_ROT(v00, v10); LISREG_ROT(v01, v11); LISREG_ROT(v02, v12);
            }
        }
        {   // (0,2)
            const float p = a13;
            if (fabsf(p) > 0.f) {
                const float y = (w2 - w0) * 0.5f;
                LISREG_JACOBI_CST(p, y)
                a13 = 0.f; w0 -= t; w2 += t;
                LISREG_ROT(a12, a23);
                LISREG_ROT(v00, v20); LISREG_ROT(v01, v21); LISREG_ROT(v02, v22);
            }
        }
        {   // (1,2)
            const float p = a23;
            if (fabsf(p) > 0.f) {
                const float y = (w2 - w1) * 0.5f;
                LISREG_JACOBI_CST(p, y)
                a23 = 0.f; w1 -= t; w2 += t;
                // rows 1,2 rotate: elements A[0][1], A[0][2]
                LISREG_ROT(a12, a13);
                LISREG_ROT(v10, v20); LISREG_ROT(v11, v21); LISREG_ROT(v12, v22);
            }
        }
    }
    // descending order; only the top vector is needed (odomEstimationNode.cpp:692-702 read D[0], D[1], V row 0)
    float e0 = w0, e1 = w1, e2 = w2;
    float x = v00, y = v01, z = v02;
    if (e1 > e0) { float t = e0; e0 = e1; e1 = t; x = v10; y = v11; z = v12; }
    if (e2 > e0) { float t = e0; e0 = e2; e2 = t; x = v20; y = v21; z = v22; }
    l0 = e0; l1 = fmaxf(e1, e2);
    v0[0] = x; v0[1] = y; v0[2] = z;
}

// cornerOptimization body for one point (odomEstimationNode.cpp:658-742), split in two:
//   corner_model : depends only on the five neighbours (centroid, principal direction, lambda0 > 3*lambda1 test);
//   corner_eval  : depends on the transformed query point (line residual, robust weight, accept test).
// m0 = (cx, cy, cz, valid ? 1 : NaN), m1 = (vx, vy, vz, 0).
__device__ __forceinline__ void corner_model(const float4 nb[5], const DevParams& P, float4& m0, float4& m1)
{
    float cx = 0, cy = 0, cz = 0;
#pragma unroll
    for (int j = 0; j < 5; ++j) { cx += nb[j].x; cy += nb[j].y; cz += nb[j].z; }
    cx = fdiv5(cx); cy = fdiv5(cy); cz = fdiv5(cz);          // /5 (:664; production: reciprocal multiply, <= 1 ulp from the division)
    float a11 = 0, a12 = 0, a13 = 0, a22 = 0, a23 = 0, a33 = 0;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const float ax = nb[j].x - cx, ay = nb[j].y - cy, az = nb[j].z - cz;
        a11 += ax * ax; a12 += ax * ay; a13 += ax * az; a22 += ay * ay; a23 += ay * az; a33 += az * az;
    }
    a11 = fdiv5(a11); a12 = fdiv5(a12); a13 = fdiv5(a13); a22 = fdiv5(a22); a23 = fdiv5(a23); a33 = fdiv5(a33);
    float l0, l1, v[3];
    eigen_sym3(a11, a12, a13, a22, a23, a33, l0, l1, v);
    const bool ok = l0 > P.line_ratio * l1;                               // :692
    m0 = make_float4(cx, cy, cz, ok ? 1.f : __int_as_float(0x7fc00000));
    m1 = make_float4(v[0], v[1], v[2], 0.f);
}

__device__ __forceinline__ bool corner_eval(const float4 m0, const float4 m1, float x0, float y0, float z0, float w,
                                            const DevParams& P, float cf[4])
{
    const float cx = m0.x, cy = m0.y, cz = m0.z;
    // `cx + 0.1 * v` is double arithmetic in the reference (:697-702)
    const float x1 = (float)((double)cx + 0.1 * (double)m1.x);
    const float y1 = (float)((double)cy + 0.1 * (double)m1.y);
    const float z1 = (float)((double)cz + 0.1 * (double)m1.z);
    const float x2 = (float)((double)cx - 0.1 * (double)m1.x);
    const float y2 = (float)((double)cy - 0.1 * (double)m1.y);
    const float z2 = (float)((double)cz - 0.1 * (double)m1.z);
    const float m11 = (x0 - x1) * (y0 - y2) - (x0 - x2) * (y0 - y1);
    const float m22 = (x0 - x1) * (z0 - z2) - (x0 - x2) * (z0 - z1);
    const float m33 = (y0 - y1) * (z0 - z2) - (y0 - y2) * (z0 - z1);
    const float a012 = fsqrt(m11 * m11 + m22 * m22 + m33 * m33);
    const float l12 = fsqrt((x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2) + (z1 - z2) * (z1 - z2));
    float la, lb, lc;
    if (kExactArith) {                                                    // :713-723 `/ a012 / l12`
        la = ((y1 - y2) * m11 + (z1 - z2) * m22) / a012 / l12;
        lb = -((x1 - x2) * m11 - (z1 - z2) * m33) / a012 / l12;
        lc = -((x1 - x2) * m22 + (y1 - y2) * m33) / a012 / l12;
    } else {
        const float inv_al = __builtin_amdgcn_rcpf(a012 * l12);           // one 1-ulp reciprocal for the three `/ a012 / l12`
        la = ((y1 - y2) * m11 + (z1 - z2) * m22) * inv_al;
        lb = -((x1 - x2) * m11 - (z1 - z2) * m33) * inv_al;
        lc = -((x1 - x2) * m22 + (y1 - y2) * m33) * inv_al;
    }
    const float ld2 = a012 / l12;
    const float s = (float)(1.0 - 0.9 * (double)fabsf(ld2));
    if (kExactArith) { cf[0] = w * s * la; cf[1] = w * s * lb; cf[2] = w * s * lc; cf[3] = w * s * ld2; }      // :729-732 (w * s) * la
    else { const float ws = w * s; cf[0] = ws * la; cf[1] = ws * lb; cf[2] = ws * lc; cf[3] = ws * ld2; }
    return (m0.w == 1.f) && (s > P.accept_s);                             // :734 (uses s, not w*s)
}

__device__ __forceinline__ bool corner_coeff(const float4 nb[5], float x0, float y0, float z0, float w,
                                             const DevParams& P, float cf[4])
{
    float4 m0, m1;
    corner_model(nb, P, m0, m1);
    return corner_eval(m0, m1, x0, y0, z0, w, P, cf);
}

// Column-pivoted Householder QR least squares  [p_j] n = -1  (Eigen colPivHouseholderQr().solve, :783), all in
// registers: column swaps are branch-free selects so nothing is dynamically indexed.
__device__ __forceinline__ void lstsq5x3(const float4 nb[5], float X[3])
{
    float a[5][3], c[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) { a[i][0] = nb[i].x; a[i][1] = nb[i].y; a[i][2] = nb[i].z; c[i] = -1.f; }
    float n0 = 0, n1 = 0, n2 = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i) { n0 += a[i][0] * a[i][0]; n1 += a[i][1] * a[i][1]; n2 += a[i][2] * a[i][2]; }
    const float maxn = fsqrt(fmaxf(n0, fmaxf(n1, n2)));          // production: v_sqrt_f32 (1 ulp; the arguments are far from the denormal range)
    const float thr = fdiv5((maxn * 1.1920929e-7f) * (maxn * 1.1920929e-7f));     // / rows (a rank threshold)
    int p0 = 0, p1 = 1, p2 = 2;        // perm: column k of the working matrix is original column p_k
    int rank = 3;
    float y[3] = { 0.f, 0.f, 0.f };

#define LISREG_SWAPCOL(ca, cb, pa, pb) do { \
        _Pragma("unroll") for (int i_ = 0; i_ < 5; ++i_) { const float t_ = a[i_][ca]; a[i_][ca] = a[i_][cb]; a[i_][cb] = t_; } \
        const int tp_ = pa; pa = pb; pb = tp_; } while (0)

#define LISREG_HOUSEHOLDER(k) do { \
        const float c0_ = a[k][k]; float tail_ = 0.f; \
        _Pragma("unroll") for (int i_ = k + 1; i_ < 5; ++i_) tail_ += a[i_][k] * a[i_][k]; \
        float beta_, tau_, v_[5]; \
        if (tail_ <= 1.17549435e-38f) { tau_ = 0.f; beta_ = c0_; _Pragma("unroll") for (int i_ = 0; i_ < 5; ++i_) v_[i_] = 0.f; } \
        else { beta_ = fsqrt(c0_ * c0_ + tail_); if (c0_ >= 0.f) beta_ = -beta_; \
               _Pragma("unroll") for (int i_ = 0; i_ < 5; ++i_) v_[i_] = (i_ > k) ? fdiv(a[i_][k], c0_ - beta_) : 0.f; \
               tau_ = fdiv(beta_ - c0_, beta_); } \
        a[k][k] = beta_; \
        _Pragma("unroll") for (int j_ = k + 1; j_ < 3; ++j_) { \
            float dot_ = a[k][j_]; \
            _Pragma("unroll") for (int i_ = k + 1; i_ < 5; ++i_) dot_ += v_[i_] * a[i_][j_]; \
            dot_ *= tau_; a[k][j_] -= dot_; \
            _Pragma("unroll") for (int i_ = k + 1; i_ < 5; ++i_) a[i_][j_] -= dot_ * v_[i_]; } \
        { float dot_ = c[k]; \
          _Pragma("unroll") for (int i_ = k + 1; i_ < 5; ++i_) dot_ += v_[i_] * c[i_]; \
          dot_ *= tau_; c[k] -= dot_; \
          _Pragma("unroll") for (int i_ = k + 1; i_ < 5; ++i_) c[i_] -= dot_ * v_[i_]; } \
    } while (0)

    // k = 0
    {
        if (n1 > n0 && n1 >= n2) LISREG_SWAPCOL(0, 1, p0, p1);
        else if (n2 > n0 && n2 > n1) LISREG_SWAPCOL(0, 2, p0, p2);
        const float big = fmaxf(n0, fmaxf(n1, n2));
        if (big < thr * 5.f) rank = 0;
        else LISREG_HOUSEHOLDER(0);
    }
    if (rank == 3) {   // k = 1
        float m1 = 0, m2 = 0;
#pragma unroll
        for (int i = 1; i < 5; ++i) { m1 += a[i][1] * a[i][1]; m2 += a[i][2] * a[i][2]; }
        if (m2 > m1) LISREG_SWAPCOL(1, 2, p1, p2);
        if (fmaxf(m1, m2) < thr * 4.f) rank = 1;
        else LISREG_HOUSEHOLDER(1);
    }
    if (rank == 3) {   // k = 2
        float m2 = 0;
#pragma unroll
        for (int i = 2; i < 5; ++i) m2 += a[i][2] * a[i][2];
        if (m2 < thr * 3.f) rank = 2;
        else LISREG_HOUSEHOLDER(2);
    }
    if (rank == 3) {
        y[2] = fdiv(c[2], a[2][2]);
        y[1] = fdiv(c[1] - a[1][2] * y[2], a[1][1]);
        y[0] = fdiv(c[0] - a[0][1] * y[1] - a[0][2] * y[2], a[0][0]);
    } else if (rank == 2) {
        y[1] = c[1] / a[1][1];
        y[0] = (c[0] - a[0][1] * y[1]) / a[0][0];
    } else if (rank == 1) {
        y[0] = c[0] / a[0][0];
    }
    X[0] = X[1] = X[2] = 0.f;
    // x[perm[i]] = y[i]
    X[0] = (p0 == 0) ? y[0] : ((p1 == 0) ? y[1] : y[2]);
    X[1] = (p0 == 1) ? y[0] : ((p1 == 1) ? y[1] : y[2]);
    X[2] = (p0 == 2) ? y[0] : ((p1 == 2) ? y[1] : y[2]);
#undef LISREG_SWAPCOL
#undef LISREG_HOUSEHOLDER
}

// surfOptimization body for one point (odomEstimationNode.cpp:776-821), split like the edge case:
//   surf_model : plane (pa,pb,pc,pd) of the five neighbours + the |n.p+d| <= 0.2 validity test; pd = NaN if invalid;
//   surf_eval  : point-to-plane residual, range-scaled robust weight, accept test for the transformed query.
// The least-squares solution of [p_j] n = -1 in closed form (production arithmetic).  With c the centroid of the five points, d_j = p_j - c
// and S = sum d_j d_j^T (the 3 x 3 scatter matrix; sum d_j = 0), the normal equations (S + 5 c c^T) n = -5 c give, by Sherman-Morrison,
// n = -5 S^-1 c / (1 + 5 c^T S^-1 c): with w = adj(S) c and D = det(S),  n / |n| = -w / |w|  and  1 / |n| = (D / 5 + c . w) / |w|  —
// exactly the (pa, pb, pc, pd) the reference forms from Eigen's column-pivoted QR solution (:783-791), from ~110 vector instructions
// instead of ~300, and better conditioned in float: the QR works on coordinates of magnitude |c| (tens of metres) whose five rows differ
// by decimetres (relative error ~ cond(A) eps ~ 1e-4 for a far patch), the scatter matrix on the differences alone.  Where the five points
// are close to ONE LINE (second eigenvalue of S under 1e-2 of the first: adj(S) loses its digits, and the reference's answer is whatever
// the pivoted QR makes of a rank-deficient system) the QR runs as before: `false` is returned.
#ifndef LISREG_PLANE_CLOSED
#define LISREG_PLANE_CLOSED 1
#endif
#ifndef LISREG_PLANE_LINE_RATIO
#define LISREG_PLANE_LINE_RATIO 1e-2f
#endif
__device__ __forceinline__ bool plane5_closed(const float4 nb[5], float& pa, float& pb, float& pc, float& pd)
{
    const float cx = ((nb[0].x + nb[1].x) + (nb[2].x + nb[3].x) + nb[4].x) * 0.2f;
    const float cy = ((nb[0].y + nb[1].y) + (nb[2].y + nb[3].y) + nb[4].y) * 0.2f;
    const float cz = ((nb[0].z + nb[1].z) + (nb[2].z + nb[3].z) + nb[4].z) * 0.2f;
    float sxx = 0.f, sxy = 0.f, sxz = 0.f, syy = 0.f, syz = 0.f, szz = 0.f;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const float dx = nb[j].x - cx, dy = nb[j].y - cy, dz = nb[j].z - cz;
        sxx += dx * dx; sxy += dx * dy; sxz += dx * dz; syy += dy * dy; syz += dy * dz; szz += dz * dz;
    }
    const float axx = syy * szz - syz * syz, axy = sxz * syz - sxy * szz, axz = sxy * syz - sxz * syy;
    const float ayy = sxx * szz - sxz * sxz, ayz = sxy * sxz - sxx * syz, azz = sxx * syy - sxy * sxy;
    const float tr = sxx + syy + szz, tra = axx + ayy + azz;               // l1 + l2 + l3,  l1 l2 + l1 l3 + l2 l3
    const float wx = axx * cx + axy * cy + axz * cz, wy = axy * cx + ayy * cy + ayz * cz, wz = axz * cx + ayz * cy + azz * cz;
    const float det = sxx * axx + sxy * axy + sxz * axz;
    const float ww = wx * wx + wy * wy + wz * wz;
    const float iw = __builtin_amdgcn_rsqf(ww);
    pa = -wx * iw; pb = -wy * iw; pc = -wz * iw;
    pd = (det * 0.2f + (cx * wx + cy * wy + cz * wz)) * iw;
    return tra > LISREG_PLANE_LINE_RATIO * (tr * tr) && ww > 0.f && ww < 3.0e38f;
}

__device__ __forceinline__ float4 surf_model(const float4 nb[5], const DevParams& P)
{
    float pa, pb, pc, pd;
    bool closed = false;
    if (!kExactArith && LISREG_PLANE_CLOSED) closed = plane5_closed(nb, pa, pb, pc, pd);
#if LISREG_PLANE_CLOSED == 2        /* timing experiment: never the QR */
    closed = true;
#endif
    if (!closed) {
        float X[3];
        lstsq5x3(nb, X);
        pa = X[0]; pb = X[1]; pc = X[2]; pd = 1.f;
        const float ps = fsqrt(pa * pa + pb * pb + pc * pc);
        if (kExactArith) { pa /= ps; pb /= ps; pc /= ps; pd /= ps; }          // :790-791
        else { const float ips = __builtin_amdgcn_rcpf(ps); pa *= ips; pb *= ips; pc *= ips; pd = ips; }   // 1-ulp reciprocal (<= 2 ulp from the divisions)
    }
    bool valid = true;
#pragma unroll
    for (int j = 0; j < 5; ++j)
        valid = valid && !(fabsf(pa * nb[j].x + pb * nb[j].y + pc * nb[j].z + pd) > P.plane_tol);
    return make_float4(pa, pb, pc, valid ? pd : __int_as_float(0x7fc00000));
}

__device__ __forceinline__ bool surf_eval(const float4 m, float x0, float y0, float z0, float w, const DevParams& P,
                                          float cf[4])
{
    const float pa = m.x, pb = m.y, pc = m.z, pd = m.w;
    const float pd2 = pa * x0 + pb * y0 + pc * z0 + pd;
    const float rng = fsqrt(fsqrt(x0 * x0 + y0 * y0 + z0 * z0));
    // s = 1 - 0.9 * fabs(pd2) / sqrt(sqrt(...)) is a DOUBLE expression in the reference (:806, 0.9 is a double literal).
    float s;
    if (kExactArith) {
        s = (float)(1.0 - 0.9 * (double)fabsf(pd2) / (double)rng);
    } else {
        // the quotient through a reciprocal refined twice from the 1-ulp float seed (relative error ~2^-52, against a ~20-instruction IEEE
        // double division with its quarter-rate v_rcp_f64): the float s differs from the divided one about once in 10^8 evaluations, by one ulp
        const double rd = (double)rng;
        double ir = (double)__builtin_amdgcn_rcpf(rng);
        ir = ir * (2.0 - rd * ir);
        ir = ir * (2.0 - rd * ir);
        s = (float)(1.0 - (0.9 * (double)fabsf(pd2)) * ir);
    }
    if (kExactArith) { cf[0] = w * s * pa; cf[1] = w * s * pb; cf[2] = w * s * pc; cf[3] = w * s * pd2; }      // :808-811
    else { const float ws = w * s; cf[0] = ws * pa; cf[1] = ws * pb; cf[2] = ws * pc; cf[3] = ws * pd2; }
    return (pd == pd) && (s > P.accept_s);
}

__device__ __forceinline__ bool surf_coeff(const float4 nb[5], float x0, float y0, float z0, float w,
                                           const DevParams& P, float cf[4])
{
    return surf_eval(surf_model(nb, P), x0, y0, z0, w, P, cf);
}

// LMOptimization row (odomEstimationNode.cpp:862-915); (ox,oy,oz) is the UNtransformed source point
__device__ __forceinline__ void jacobian_row(const float* K /* ItemState::jk */, float ox, float oy, float oz, const float cf[4],
                                             float row[6], float& b)
{
    // the pose-only factors K come from write_pose_cache (lisreg_solve.hip), formed with the reference's association
    const float px = oy, py = oz, pz = ox;
    const float cx = cf[1], cy = cf[2], cz = cf[0];
    const float arx = (K[0] * px + K[1] * py - K[2] * pz) * cx +
                      (K[3] * px - K[4] * py - K[5] * pz) * cy +
                      (K[6] * px + K[7] * py - K[8] * pz) * cz;
    const float ary = (K[9] * px + K[10] * py + K[11] * pz) * cx +
                      (K[12] * px + K[13] * py - K[14] * pz) * cz;
    const float arz = (K[15] * px + K[12] * py) * cx +
                      (K[16] * px - K[17] * py) * cy +
                      (K[10] * px + K[18] * py) * cz;
    row[0] = arz; row[1] = arx; row[2] = ary; row[3] = cz; row[4] = cx; row[5] = cy;
    b = -cf[3];
}

typedef unsigned int v2u __attribute__((ext_vector_type(2)));
#if LISREG_EXACT
__device__ __forceinline__ double shfl_xor_d(double v, int mask)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, mask); hi = __shfl_xor(hi, mask);
    return __hiloint2double(hi, lo);
}

// gfx950 lane swaps (V_PERMLANE32_SWAP / V_PERMLANE16_SWAP): exchange the upper 32 lanes (odd 16-lane rows) of `a` with the
// lower 32 lanes (even rows) of `b`.  After the swap a + b is the halving-butterfly step with no select and no LDS traffic:
// lanes of the lower half hold a(l) + a(partner), lanes of the upper half hold b(partner) + b(l).
__device__ __forceinline__ double swap_add32(double a, double b)
{
    const v2u lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const v2u hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return __hiloint2double((int)hi.x, (int)lo.x) + __hiloint2double((int)hi.y, (int)lo.y);
}
__device__ __forceinline__ double swap_add16(double a, double b)
{
    const v2u lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const v2u hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return __hiloint2double((int)hi.x, (int)lo.x) + __hiloint2double((int)hi.y, (int)lo.y);
}
#endif

// The index arrays are reached through pointers stored in device structs, which the compiler can only treat as
// generic (flat_load).  They are always global memory: say so, and get global_load with a scalar base.
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const v4f* gptr_f4;
typedef float v3f __attribute__((ext_vector_type(3)));
typedef __attribute__((address_space(1))) const v3f* gptr_f3;
typedef __attribute__((address_space(1))) const int* gptr_i32;
typedef __attribute__((address_space(1))) int*       gptr_i32w;
typedef int v4i __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const v4i* gptr_i4;
typedef float v2f __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) const v2f* gptr_f2;
typedef __attribute__((address_space(1))) v4i*       gptr_i4w;

// gather of record `i` (16-byte records) from a wave-uniform base: a 32-bit unsigned byte offset lets the compiler use the
// scalar-base + vector-offset form of global_load (no 64-bit address arithmetic per lane).  i * 16 < 2^32: targets up to 268 M points.
typedef __attribute__((address_space(1))) const char* gptr_c;
#define LISREG_LD3(base, i) (*(gptr_f3)((gptr_c)(base) + ((unsigned)(i) << 4)))
#define LISREG_LD4(base, i) (*(gptr_f4)((gptr_c)(base) + ((unsigned)(i) << 4)))

__device__ __forceinline__ int grid_coord(float v, float origin, float inv_cell)
{
    return (int)floorf((v - origin) * inv_cell);
}

// w = 2.0 - LabelSorce[label] (subMapOptmizationNode.cpp:1671, :1795).  LabelSorce is a std::map read with operator[]: a label
// that is not in config/label.yaml:214-234 yields 0, i.e. w = 2.0 — wtab carries that for 20..31, labels >= 32 get it here.
__device__ __forceinline__ float label_weight(const DevParams& P, float payload)
{
    const unsigned lab = __float_as_uint(payload) & 0xffffu;
    return lab < 32u ? P.wtab[lab] : 2.0f;
}

// Jacobian row + fixed-order fp64 reduction of the 28 normal-equation terms: wave halving butterfly -> LDS ->
// one partial row per workgroup.  `ok` = this lane contributes a correspondence with coefficients cf.
__device__ __forceinline__ void row_and_reduce(bool ok, const float cf[4], const float4 q4, const float* jk,
                                               const DevParams& P, float* s_red, double* __restrict__ out);

// Residual model shared by the search front-ends that re-fit every iteration.
// (i0..i4) index g.pts, ascending by distance; i4 < 0 means "fewer than five neighbours within sqrt(tau)".
__device__ __forceinline__ bool residual_coeffs(bool valid, int i0, int i1, int i2, int i3, int i4, const GridIndex& g,
                                                const float4 q4, float qx, float qy, float qz, const DevParams& P, int kind,
                                                float cf[4])
{
    cf[0] = cf[1] = cf[2] = cf[3] = 0.f;
    bool ok = false;
    const bool found = valid && (i4 >= 0);          // five neighbours with sqDist < tau  (:657 / :776)
    if (found) {
        float4 nb[5];
        const gptr_f4 gp = (gptr_f4)g.pts;
#ifdef LISREG_XP_NOGATHER      /* timing experiment (wrong results): five synthetic neighbours on a plane through the query, no memory access */
        (void)gp;
        v4f n0, n1, n2, n3, n4;
        n0.x = qx + 0.11f; n0.y = qy + 0.02f; n0.z = qz; n0.w = 0.f;  n1.x = qx - 0.07f; n1.y = qy + 0.13f; n1.z = qz; n1.w = 0.f;
        n2.x = qx - 0.12f; n2.y = qy - 0.09f; n2.z = qz; n2.w = 0.f;  n3.x = qx + 0.05f; n3.y = qy - 0.14f; n3.z = qz; n3.w = 0.f;
        n4.x = qx + 0.01f * (float)(i4 & 7); n4.y = qy + 0.21f; n4.z = qz; n4.w = 0.f;
#else
        v4f n0, n1, n2, n3, n4;
        n0 = LISREG_LD4(gp, i0); n1 = LISREG_LD4(gp, i1); n2 = LISREG_LD4(gp, i2); n3 = LISREG_LD4(gp, i3); n4 = LISREG_LD4(gp, i4);
#endif
        nb[0] = make_float4(n0.x, n0.y, n0.z, n0.w); nb[1] = make_float4(n1.x, n1.y, n1.z, n1.w);
        nb[2] = make_float4(n2.x, n2.y, n2.z, n2.w); nb[3] = make_float4(n3.x, n3.y, n3.z, n3.w);
        nb[4] = make_float4(n4.x, n4.y, n4.z, n4.w);
        float w = 1.f;
        if (P.use_label) w = label_weight(P, q4.w);
#ifdef LISREG_XP_NOFIT          /* timing experiment (wrong results): no line / plane fit, the gathered points stay live */
        cf[0] = nb[0].x + nb[1].y + w; cf[1] = nb[2].z + nb[3].x; cf[2] = nb[4].y + nb[0].z; cf[3] = (nb[1].x + nb[2].y + nb[3].z + nb[4].x) * 1e-3f;
        ok = cf[3] < 1.0e30f;
#else
        ok = (kind == 0) ? corner_coeff(nb, qx, qy, qz, w, P, cf) : surf_coeff(nb, qx, qy, qz, w, P, cf);
#endif
    }
    return ok;
}

__device__ __forceinline__ void residual_and_reduce(bool valid, int i0, int i1, int i2, int i3, int i4,
                                                    const GridIndex& g, const float4 q4, float qx, float qy, float qz,
                                                    const float* jk, const DevParams& P, int kind,
                                                    float* s_red, double* __restrict__ out, int* dbg_ok = nullptr)
{
    float cf[4];
    const bool ok = residual_coeffs(valid, i0, i1, i2, i3, i4, g, q4, qx, qy, qz, P, kind, cf);
    if (dbg_ok && valid) *dbg_ok = ok ? 1 : 0;          // "dump_neighbors": row 5 = this point contributed a correspondence
    row_and_reduce(ok, cf, q4, jk, P, s_red, out);
}

__device__ __forceinline__ void row_and_reduce(bool ok, const float cf[4], const float4 q4, const float* jk,
                                               const DevParams& P, float* s_red, double* __restrict__ out)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef LISREG_XP_NOREDUCE       /* timing experiment (wrong results): no Jacobian row, no sums */
    if (ok && tid < kNumAcc) out[tid] = (double)(cf[0] + cf[1] + cf[2] + cf[3] + q4.x);
    return;
#endif
    double (*s_acc)[kRedWaveFloats / 2] = reinterpret_cast<double (*)[kRedWaveFloats / 2]>(s_red);     // s_acc[wave][k]: the wave's region as doubles
    (void)s_acc;
    float row[6] = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f }, rb = 0.f, one = 0.f;
    if (ok) { jacobian_row(jk, q4.x, q4.y, q4.z, cf, row, rb); one = 1.f; }
#if LISREG_EXACT
    // the 28 normal-equation terms of this row, produced on demand (float x float is exact in double):
    //   k = 0..20 upper triangle of row^T row (row-major), 21..26 row * b, 27 the correspondence count
    auto term = [&](int k) -> double {
        constexpr int R[21] = { 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4, 4, 5 };
        constexpr int C[21] = { 0, 1, 2, 3, 4, 5, 1, 2, 3, 4, 5, 2, 3, 4, 5, 3, 4, 5, 4, 5, 5 };
        if (k < 21) return (double)row[R[k]] * (double)row[C[k]];
        if (k < 27) return (double)row[k - 21] * (double)rb;
        if (k == 27) return (double)one;
        return 0.0;
    };
    // ---- fixed-order fp64 reduction: halving butterfly across the wave -> LDS -> one partial row ---------------
    // Each step exchanges HALF of the values with the partner lane (xor 32, 16, 8, 4, 2), so 32 values need
    // 16+8+4+2+1(+1) 64-bit shuffles instead of 32 x 6; after five steps lane l holds the 32-lane sum of value
    // index bits(l)[5:1] and one last xor-1 step completes it.  Deterministic: the pairing is fixed.
    double v16[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v16[i] = swap_add32(term(i), term(i + 16));       // lanes 0-31: term i, lanes 32-63: term i + 16
    double v8[8], v4[4], v2[2], v1;
#pragma unroll
    for (int i = 0; i < 8; ++i) v8[i] = swap_add16(v16[i], v16[i + 8]);            // even rows: v16[i], odd rows: v16[i + 8]
    {
        const bool up = (lane & 8) != 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) v4[i] = (up ? v8[i + 4] : v8[i]) + shfl_xor_d(up ? v8[i] : v8[i + 4], 8);
    }
    {
        const bool up = (lane & 4) != 0;
#pragma unroll
        for (int i = 0; i < 2; ++i) v2[i] = (up ? v4[i + 2] : v4[i]) + shfl_xor_d(up ? v4[i] : v4[i + 2], 4);
    }
    {
        const bool up = (lane & 2) != 0;
        v1 = (up ? v2[1] : v2[0]) + shfl_xor_d(up ? v2[0] : v2[1], 2);
    }
    v1 += shfl_xor_d(v1, 1);
    {
        // value index held by this lane: bit5 -> +16, bit4 -> +8, bit3 -> +4, bit2 -> +2, bit1 -> +1
        const int idx = ((lane >> 5) & 1) * 16 + ((lane >> 4) & 1) * 8 + ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
        if ((lane & 1) == 0 && idx < kNumAcc) s_acc[wave][idx] = v1;
    }
#else
    // the 28 normal-equation terms of this row, produced on demand:
    //   k = 0..20 upper triangle of row^T row (row-major), 21..26 row * b, 27 the correspondence count
    // ---- fixed-order reduction of a wavefront's 64 rows: fp32 halving butterfly, then fp64 from the wave sums on ------------------
    // The kernel is bound by vector instructions and fp64 ones cost double: 28 fp64 products + a 32-step fp64 butterfly were 17 % of
    // a steady-state launch.  Here the products and the 64-term tree are fp32 (relative error of a wave sum <= ~2e-7 of the sum of
    // magnitudes — the reference's own matAtA is a float cv::gemm, and the step is solved in float); everything above a wave (four
    // waves of a workgroup, workgroups of a registration) stays fp64 in a fixed order.  The exact build (LISREG_EXACT) runs the all-fp64
    // form above.  Each step exchanges HALF of the values with a partner lane in the other half of the
    // group (xor 32 and 16 by V_PERMLANE32/16_SWAP; 15, 7, 3, 1 — row mirror, half-row mirror, quad mirror, neighbour — as DPP
    // operands of the add), so 32 values need 16 + 8 + 4 + 2 + 1 (+ 1) adds; lane l ends with the sum of value index bits(l)[5:1].
    auto term = [&](int k) -> float {
        constexpr int R[21] = { 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4, 4, 5 };
        constexpr int C[21] = { 0, 1, 2, 3, 4, 5, 1, 2, 3, 4, 5, 2, 3, 4, 5, 3, 4, 5, 4, 5, 5 };
        if (k < 21) return row[R[k]] * row[C[k]];
        if (k < 27) return row[k - 21] * rb;
        if (k == 27) return one;
        return 0.f;
    };
    auto swap32 = [](float a, float b) -> float {
        const v2u r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
        return __uint_as_float(r.x) + __uint_as_float(r.y);
    };
    auto swap16 = [](float a, float b) -> float {
        const v2u r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
        return __uint_as_float(r.x) + __uint_as_float(r.y);
    };
    float v16[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v16[i] = swap32(term(i), term(i + 16));           // lanes 0-31: term i, lanes 32-63: term i + 16
    float v8[8], v4[4], v2[2], v1;
#pragma unroll
    for (int i = 0; i < 8; ++i) v8[i] = swap16(v16[i], v16[i + 8]);                // even rows: v16[i], odd rows: v16[i + 8]
    {
        const bool up = (lane & 8) != 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            v4[i] = (up ? v8[i + 4] : v8[i]) + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(up ? v8[i] : v8[i + 4]), 0x140, 0xF, 0xF, true));   // row_mirror: lane ^ 15
    }
    {
        const bool up = (lane & 4) != 0;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            v2[i] = (up ? v4[i + 2] : v4[i]) + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(up ? v4[i] : v4[i + 2]), 0x141, 0xF, 0xF, true));   // row_half_mirror: lane ^ 7
    }
    {
        const bool up = (lane & 2) != 0;
        v1 = (up ? v2[1] : v2[0]) + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(up ? v2[0] : v2[1]), 0x1B, 0xF, 0xF, true));                  // quad_perm [3,2,1,0]: lane ^ 3
    }
    v1 += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v1), 0xB1, 0xF, 0xF, true));                                                              // quad_perm [1,0,3,2]: lane ^ 1
    {
        // value index held by this lane: bit5 -> +16, bit4 -> +8, bit3 -> +4, bit2 -> +2, bit1 -> +1
        const int idx = ((lane >> 5) & 1) * 16 + ((lane >> 4) & 1) * 8 + ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
        if ((lane & 1) == 0 && idx < kNumAcc) s_acc[wave][idx] = (double)v1;
    }
#endif
    __syncthreads();
    if (tid < kNumAcc) {
        double v = s_acc[0][tid];
#pragma unroll
        for (int w = 1; w < kBlockQ / 64; ++w) v += s_acc[w][tid];        // fixed order: ((w0 + w1) + w2) + w3
        out[tid] = v;
    }
}


__global__ __launch_bounds__(kBlockQ) void k_assoc_staged(const BlockDesc* __restrict__ blocks,
                                                   const Segment* __restrict__ segs,
                                                   const GridIndex* __restrict__ grids,
                                                   const ItemState* __restrict__ items, const DevParams P,
                                                   const float4* __restrict__ sorted_all,
                                                   double* __restrict__ partials)
{
    __shared__ float4 s_pts[kStageCap];
    __shared__ int    s_run_start[kBlockQ];
    __shared__ int    s_run_off[kBlockQ + 1];
    __shared__ int    s_wave[4];
    __shared__ float  s_bb[4][6];
    __shared__ __attribute__((aligned(16))) float s_red[4 * kRedWaveFloats];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const BlockDesc bd = blocks[blockIdx.x];
    const ItemState* it = &items[bd.item];
    if (it->done) return;                       // converged / guarded-out item: solve kernel skips it too
    const Segment sg = segs[bd.seg];
    const GridIndex g = grids[sg.target];
    double* out = partials + (size_t)blockIdx.x * kNumAcc;
    if (g.n < 5) {                              // nearestKSearch cannot return 5 neighbours: no correspondences
        if (tid < kNumAcc) out[tid] = 0.0;
        return;
    }

    const float* M = it->M;            // trans2Affine3f(T), cached by the solve kernel (uniform -> SGPRs)

    const bool valid = tid < bd.count;
    float4 q4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) q4 = sorted_all ? sorted_all[sg.flat_base + bd.start + tid] : sg.src[bd.start + tid];
    // pointAssociateToMap (:243-258)
    float qx = M[0] * q4.x + M[1] * q4.y + M[2] * q4.z + M[3];
    float qy = M[4] * q4.x + M[5] * q4.y + M[6] * q4.z + M[7];
    float qz = M[8] * q4.x + M[9] * q4.y + M[10] * q4.z + M[11];

    // ---- workgroup bounding box of the transformed queries --------------------------------------------------
    float lo[3] = { valid ? qx : 3.0e38f, valid ? qy : 3.0e38f, valid ? qz : 3.0e38f };
    float hi[3] = { valid ? qx : -3.0e38f, valid ? qy : -3.0e38f, valid ? qz : -3.0e38f };
#pragma unroll
    for (int d = 32; d > 0; d >>= 1)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], d));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], d));
        }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { s_bb[wave][k] = lo[k]; s_bb[wave][3 + k] = hi[k]; }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        lo[k] = fminf(fminf(s_bb[0][k], s_bb[1][k]), fminf(s_bb[2][k], s_bb[3][k]));
        hi[k] = fmaxf(fmaxf(s_bb[0][3 + k], s_bb[1][3 + k]), fmaxf(s_bb[2][3 + k], s_bb[3][3 + k]));
    }
    if (!valid) { qx = qy = qz = 3.0e18f; }     // never closer than tau to anything

    // ---- covered cell range: every target point with |p - q|_inf <= margin lies inside ----------------------
    const float margin = sqrtf(P.tau) * 1.0005f + 1e-3f;
    int ix0 = grid_coord(lo[0] - margin, g.ox, g.inv_cell), ix1 = grid_coord(hi[0] + margin, g.ox, g.inv_cell);
    int iy0 = grid_coord(lo[1] - margin, g.oy, g.inv_cell), iy1 = grid_coord(hi[1] + margin, g.oy, g.inv_cell);
    int iz0 = grid_coord(lo[2] - margin, g.oz, g.inv_cell), iz1 = grid_coord(hi[2] + margin, g.oz, g.inv_cell);
    ix0 = max(ix0, 0); iy0 = max(iy0, 0); iz0 = max(iz0, 0);
    ix1 = min(ix1, g.nx - 1); iy1 = min(iy1, g.ny - 1); iz1 = min(iz1, g.nz - 1);
    const int nrx = ix1 - ix0 + 1, nry = iy1 - iy0 + 1;
    const int nruns = (nrx > 0 && nry > 0 && iz1 >= iz0) ? nrx * nry : 0;

    // ---- exact fixed-radius 5-NN: sorted top-5 in registers, initialised at tau ------------------------------
    float b0 = P.tau, b1 = P.tau, b2 = P.tau, b3 = P.tau, b4 = P.tau;
    int   i0 = -1, i1 = -1, i2 = -1, i3 = -1, i4 = -1;

    for (int rb = 0; rb < nruns; rb += kBlockQ) {
        const int r = rb + tid;
        int rs = 0, rl = 0;
        if (r < nruns) {
            const int ix = ix0 + r / nry, iy = iy0 + r % nry;
            const int base = (ix * g.ny + iy) * g.nz;
            rs = g.cell_start[base + iz0];
            rl = g.cell_start[base + iz1 + 1] - rs;
        }
        // workgroup exclusive scan of run lengths
        int inc = rl;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(inc, d); if (lane >= d) inc += t; }
        if (lane == 63) s_wave[wave] = inc;
        __syncthreads();
        int wbase = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { const int s = s_wave[w]; if (w < wave) wbase += s; total += s; }
        s_run_start[tid] = rs;
        s_run_off[tid] = wbase + inc - rl;
        if (tid == 0) s_run_off[kBlockQ] = total;
        __syncthreads();

        for (int chunk = 0; chunk < total; chunk += kStageCap) {
            const int cnt = min(kStageCap, total - chunk);
            // stage: flat candidate id -> (run, offset) by binary search over the run offsets
            for (int j = tid; j < cnt; j += kBlockQ) {
                const int gidx = chunk + j;
                int l = 0, h = kBlockQ - 1;
                while (l < h) {
                    const int mid = (l + h + 1) >> 1;
                    if (s_run_off[mid] <= gidx) l = mid; else h = mid - 1;
                }
                const int src = s_run_start[l] + (gidx - s_run_off[l]);
                float4 v = g.pts[src];
                v.w = __int_as_float(src);                 // position in the sorted target array
                s_pts[j] = v;
            }
            __syncthreads();
            // scan: every lane tests every staged point (LDS broadcast read)
#pragma unroll 4
            for (int j = 0; j < cnt; ++j) {
                const float4 c = s_pts[j];
                const float dx = qx - c.x, dy = qy - c.y, dz = qz - c.z;
                const float d2 = dx * dx + dy * dy + dz * dz;     // flann::L2_Simple order
                if (d2 < b4) {
                    const int id = __float_as_int(c.w);
                    const bool c3 = d2 < b3, c2 = d2 < b2, c1 = d2 < b1, c0 = d2 < b0;
                    b4 = c3 ? b3 : d2;               i4 = c3 ? i3 : id;
                    b3 = c3 ? (c2 ? b2 : d2) : b3;   i3 = c3 ? (c2 ? i2 : id) : i3;
                    b2 = c2 ? (c1 ? b1 : d2) : b2;   i2 = c2 ? (c1 ? i1 : id) : i2;
                    b1 = c1 ? (c0 ? b0 : d2) : b1;   i1 = c1 ? (c0 ? i0 : id) : i1;
                    b0 = c0 ? d2 : b0;               i0 = c0 ? id : i0;
                }
            }
            __syncthreads();
        }
    }

    residual_and_reduce(valid, i0, i1, i2, i3, i4, g, q4, qx, qy, qz, it->jk, P, sg.kind, s_red, out);
}

// sorted top-5 insertion (ascending); `id` must not already be in the list
#if LISREG_MED3_INSERT
// the inner distances of the new list are medians of three (the list ascends and d2 < b4 here): three v_med3_f32 instead of six selects,
// same values (d2 equal to an entry: both forms keep the entry in front)
#define LISREG_INSERT(d2, id) do { \
        const bool c3_ = (d2) < b3, c2_ = (d2) < b2, c1_ = (d2) < b1, c0_ = (d2) < b0; \
        const float n4_ = c3_ ? b3 : (d2), n3_ = __builtin_amdgcn_fmed3f(b2, b3, (d2)); \
        const float n2_ = __builtin_amdgcn_fmed3f(b1, b2, (d2)), n1_ = __builtin_amdgcn_fmed3f(b0, b1, (d2)); \
        const float n0_ = c0_ ? (d2) : b0; \
        i4 = c3_ ? i3 : (id); \
        i3 = c3_ ? (c2_ ? i2 : (id)) : i3; \
        i2 = c2_ ? (c1_ ? i1 : (id)) : i2; \
        i1 = c1_ ? (c0_ ? i0 : (id)) : i1; \
        i0 = c0_ ? (id) : i0; \
        b4 = n4_; b3 = n3_; b2 = n2_; b1 = n1_; b0 = n0_; } while (0)
#else
#define LISREG_INSERT(d2, id) do { \
        const bool c3_ = (d2) < b3, c2_ = (d2) < b2, c1_ = (d2) < b1, c0_ = (d2) < b0; \
        b4 = c3_ ? b3 : (d2);               i4 = c3_ ? i3 : (id); \
        b3 = c3_ ? (c2_ ? b2 : (d2)) : b3;  i3 = c3_ ? (c2_ ? i2 : (id)) : i3; \
        b2 = c2_ ? (c1_ ? b1 : (d2)) : b2;  i2 = c2_ ? (c1_ ? i1 : (id)) : i2; \
        b1 = c1_ ? (c0_ ? b0 : (d2)) : b1;  i1 = c1_ ? (c0_ ? i0 : (id)) : i1; \
        b0 = c0_ ? (d2) : b0;               i0 = c0_ ? (id) : i0; } while (0)
#endif

// Per-lane grid walk: every lane visits only the cells that can hold a point closer than its current 5th-best
// distance (pruned per x-slab, per (x,y) column and per z-range), reading candidates straight from the
// cell-sorted target through L1/L2 (16-B records, contiguous per cell run, 4 loads in flight per lane).
// `lim2` caps the search radius of this pass (squared).
#define LISREG_TRY(d2_, j_) do { \
        if ((d2_) < b4) { \
            const bool dup_ = (j_) == i0 || (j_) == i1 || (j_) == i2 || (j_) == i3 || (j_) == i4; \
            if (!dup_) LISREG_INSERT((d2_), (j_)); } } while (0)

// Equal distances (kTies instantiations; option "canonical_ties", always on in the exact build).  "First met wins" makes the five kept —
// and their order, hence the float sums of the fit — depend on the order in which a front-end meets its candidates (cell walk, graph
// scan, eight lanes per query all differ) when two target points are equidistant from the query to the last bit (about one query in
// 10^5-10^6 on scan data; every query of a map that holds a point twice).  With kTies the search NOTES a tie that can matter (`tie`, a
// lane mask) and a noted lane re-selects its five among the points inside its final radius by (distance, ORIGINAL index) —
// LISREG_CANONICAL_FIVE below — so every front-end returns the same five in the same order, always.  Where it is noted: once per
// candidate GROUP, after its insertions: (1) a candidate of the group at exactly the new 5th-best distance that is not the new fifth
// itself — it competes with it whichever was met first; (2) two equal distances next to each other in the list.  That is complete: a tie
// can only enter the list through an insertion (caught by (2) at the end of that group), an entry can only be DROPPED tied with the fifth
// that stays if the two were neighbours in the list at the start of the group (caught by (2) at the end of the group before, or by the
// check after the seeds / the sorting network), and a candidate equal to an older entry is inserted BEHIND it (strict '<'), so it is the
// candidate that falls first.  Cost (measured): ~170 vector instructions per wavefront, +7 % on a steady-state launch.
#define LISREG_GROUP_TIES(e0_, k0_, e1_, k1_, e2_, k2_, e3_, k3_) do { if (kTies) { \
        /* bitwise, not short-circuit: a dozen compares into lane masks and scalar ors, no branches */ \
        /* (2) between REAL entries only (empty slots all carry tau), whether or not the list is full yet: a pair that ties while the list \
           still has room can lose its second member to closer candidates of a later group — the member that stays must be the canonical one */ \
        const bool t_ = ((int)(b0 == b1) & (int)(i1 >= 0)) | ((int)(b1 == b2) & (int)(i2 >= 0)) | ((int)(b2 == b3) & (int)(i3 >= 0)) | \
                        ((int)(b3 == b4) & (int)(i4 >= 0)) | \
                        ((int)(i4 >= 0) & (((int)((e0_) == b4) & (int)((k0_) != i4)) | ((int)((e1_) == b4) & (int)((k1_) != i4)) | \
                                           ((int)((e2_) == b4) & (int)((k2_) != i4)) | ((int)((e3_) == b4) & (int)((k3_) != i4)))); \
        tie = (int)tie | (int)t_; } } while (0)
#define LISREG_LIST_TIES() do { if (kTies) tie = (int)tie | ((int)(b0 == b1) & (int)(i1 >= 0)) | ((int)(b1 == b2) & (int)(i2 >= 0)) | \
                                                  ((int)(b2 == b3) & (int)(i3 >= 0)) | ((int)(b3 == b4) & (int)(i4 >= 0)); } while (0)
// the "nothing closer" gate of a group: with kTies a candidate AT the bound has to be looked at too
#define LISREG_GATE(m_) (kTies ? (m_) <= b4 : (m_) < b4)

// Flattened walk.  Nested (x, y, run) loops — the first version of this kernel — cost, per wave, the SUM over columns of the
// longest run any lane has in that column (16.7 candidate groups in the steady state of configs[1]), although
// the busiest lane needs 11 and the average lane 7.  Here every lane first collects the non-empty cell runs of its columns
// into a small per-lane list in LDS (kWalkCap entries), then streams through the list in ONE loop, so a wave pays the longest TOTAL, not the sum of
// per-column maxima.  Lists are flushed whenever a lane's list is full, which also re-prunes against the shrinking bound
// during the wide early-iteration walks.  Same columns, same candidate order, same inserts: the result is identical.
constexpr int kWalkCap = 8;
constexpr int kShareRun = 32;          // kQ lanes per query: candidate runs of this length or more are shared by the kQ lanes
// INNER: restrict this pass to the 3 x 3 columns around the query's own column and remember that box (sx0..sy1);
// SKIP: leave out the columns of the remembered box (a preceding INNER pass covered them with a z-range at least as wide).
// An INNER pass followed by a SKIP pass is the same walk centre-first: the bound is tight before the outer columns are
// looked at, which then mostly fail the pruning test without touching memory (matters for the wide walks of the first
// Gauss-Newton iterations, whose seeds are a pose step away from the truth).
// four consecutive candidates j_ .. j_ + 3 of a run that ends at l_ (indices clamped into the run: a repeated candidate is a duplicate)
#define LISREG_WALK_FOUR(j_, l_) do { \
        const int j1_ = min((j_) + 1, (l_)), j2_ = min((j_) + 2, (l_)), j3_ = min((j_) + 3, (l_)); \
        /* only x, y, z take part in the search: 12-byte loads (the original index in .w is read for the final five) */ \
        const v3f c0_ = LISREG_LD3(pts, (j_)), c1_ = LISREG_LD3(pts, j1_), c2_ = LISREG_LD3(pts, j2_), c3_ = LISREG_LD3(pts, j3_);     /* scalar base + 32-bit offset: with 64-bit lane addresses the counted loop below lost 6 % on configs[3] to the scheduler (a wait in front of the fourth load) */ \
        const float ax_ = qx - c0_.x, ay_ = qy - c0_.y, az_ = qz - c0_.z; \
        const float bx_ = qx - c1_.x, by_ = qy - c1_.y, bz_ = qz - c1_.z; \
        const float gx_ = qx - c2_.x, gy_ = qy - c2_.y, gz_ = qz - c2_.z; \
        const float hx_ = qx - c3_.x, hy_ = qy - c3_.y, hz_ = qz - c3_.z; \
        const float e0_ = ax_ * ax_ + ay_ * ay_ + az_ * az_, e1_ = bx_ * bx_ + by_ * by_ + bz_ * bz_; \
        const float e2_ = gx_ * gx_ + gy_ * gy_ + gz_ * gz_, e3_ = hx_ * hx_ + hy_ * hy_ + hz_ * hz_; \
        if (LISREG_GATE(fminf(fminf(e0_, e1_), fminf(e2_, e3_)))) { \
            LISREG_TRY(e0_, (j_)); LISREG_TRY(e1_, j1_); LISREG_TRY(e2_, j2_); LISREG_TRY(e3_, j3_); \
            LISREG_GROUP_TIES(e0_, (j_), e1_, j1_, e2_, j2_, e3_, j3_); \
        } } while (0)

#define LISREG_WALK_LIST(lim2_expr, INNER, SKIP) do { \
        const float lim_ = fminf(b4, (lim2_expr)); \
        const float rad_ = __builtin_amdgcn_sqrtf(lim_) * 1.0001f + kEps; \
        int cx0_ = max(grid_coord(qx - rad_, g.ox, g.inv_cell), 0), cx1_ = min(grid_coord(qx + rad_, g.ox, g.inv_cell), g.nx - 1); \
        int cy0_ = max(grid_coord(qy - rad_, g.oy, g.inv_cell), 0), cy1_ = min(grid_coord(qy + rad_, g.oy, g.inv_cell), g.ny - 1); \
        const int cz0_ = max(grid_coord(qz - rad_, g.oz, g.inv_cell), 0), cz1_ = min(grid_coord(qz + rad_, g.oz, g.inv_cell), g.nz - 1); \
        if (INNER) { \
            const int hx_ = grid_coord(qx, g.ox, g.inv_cell), hy_ = grid_coord(qy, g.oy, g.inv_cell); \
            cx0_ = max(cx0_, hx_ - 1); cx1_ = min(cx1_, hx_ + 1); cy0_ = max(cy0_, hy_ - 1); cy1_ = min(cy1_, hy_ + 1); \
            sx0_ = cx0_; sx1_ = cx1_; sy0_ = cy0_; sy1_ = cy1_; \
        } \
        /* kQ lanes per query: the columns of the box are dealt round-robin in box order (lane s takes columns s, s + kQ, ...), and every \
           lane steps through ITS columns only — the probes of the kQ lanes are in flight together, a kQ-th of the dependent rounds */ \
        const int nyb_ = cy1_ - cy0_ + 1; \
        int ix_ = cx0_, iy_ = cy0_; \
        /* an empty box first (a non-finite query has one: its cell coordinates saturate) — nothing below may step from there */ \
        if (cz0_ > cz1_ || cy0_ > cy1_ || cx0_ > cx1_) ix_ = cx1_ + 1; \
        else if (kQ > 1) { iy_ += sub_q; while (iy_ > cy1_) { iy_ -= nyb_; ++ix_; } } \
        for (;;) { \
            /* kQ lanes per query: the group goes on while any of its lanes has columns left (its lanes share the candidates below) */ \
            if (kQ == 1 || !kShare) { if (!(ix_ <= cx1_)) break; } \
            else if (((__builtin_amdgcn_ballot_w64(ix_ <= cx1_) >> ((tid & 63) & ~(kQ - 1))) & ((1ull << kQ) - 1ull)) == 0ull) break; \
            int cnt_ = 0, grp_ = 0; \
            bool lng_ = false; \
            /* phase 1: collect up to kWalkCap non-empty runs (grp_: their groups of four candidates) */ \
            while (cnt_ < kWalkCap && ix_ <= cx1_) { \
                const float xl_ = g.ox + (float)ix_ * g.cell, yl_ = g.oy + (float)iy_ * g.cell; \
                const float dx_ = fmaxf(fmaxf(xl_ - qx, qx - (xl_ + g.cell)) - kEps, 0.f); \
                const float dy_ = fmaxf(fmaxf(yl_ - qy, qy - (yl_ + g.cell)) - kEps, 0.f); \
                const bool covered_ = (SKIP) && ix_ >= sx0_ && ix_ <= sx1_ && iy_ >= sy0_ && iy_ <= sy1_; \
                if (!covered_ && dx_ * dx_ + dy_ * dy_ < fminf(b4, lim_)) { \
                    const int base_ = (ix_ * g.ny + iy_) * g.nz; \
                    const int js_ = cells[base_ + cz0_], je_ = cells[base_ + cz1_ + 1]; \
                    if (js_ < je_) { \
                        s_runs_at(cnt_, tid) = make_int2(js_, je_); ++cnt_; \
                        if (kQ > 1 && kShare && je_ - js_ >= kShareRun) lng_ = true; else grp_ += (je_ - js_ + 3) >> 2; \
                    } \
                } \
                if (kQ == 1) { if (++iy_ > cy1_) { iy_ = cy0_; ++ix_; } } \
                else { iy_ += kQ; while (iy_ > cy1_) { iy_ -= nyb_; ++ix_; } } \
            } \
            /* phase 2: one loop over all collected candidates */ \
            if (kQ == 1 || !kShare) { \
                /* the loop leaves at its head only (a counted loop over the groups): with the exit inside the run switch the compiler kept the \
                   five-best list in two register sets and copied it twice per pass (19-29 v_mov per group of four, round 5) */ \
                int r_ = 0, j_ = 0, e_ = 0; \
                _Pragma("unroll 1") for (int g2_ = 0; g2_ < grp_; ++g2_) { \
                    if (j_ >= e_) { const int2 t_ = s_runs_at(r_, tid); j_ = t_.x; e_ = t_.y; ++r_; } \
                    LISREG_WALK_FOUR(j_, e_ - 1); \
                    j_ += 4; \
                } \
            } else { \
                /* every lane its own SHORT runs; a long run — a column that holds a pole or a stretch of wall, hundreds of points — is \
                   dealt out to all kQ lanes in groups of four candidates instead of keeping its lane busy alone */ \
                const bool long_ = lng_; \
                { \
                    int r_ = 0, j_ = 0, e_ = 0; \
                    _Pragma("unroll 1") for (int g2_ = 0; g2_ < grp_; ++g2_) {      /* (grp_ counts the groups of the SHORT runs here) */ \
                        if (j_ >= e_) { \
                            int2 t_; \
                            do { t_ = s_runs_at(r_, tid); ++r_; } while (t_.y - t_.x >= kShareRun); \
                            j_ = t_.x; e_ = t_.y; \
                        } \
                        LISREG_WALK_FOUR(j_, e_ - 1); \
                        j_ += 4; \
                    } \
                } \
                if (((__builtin_amdgcn_ballot_w64(long_) >> ((tid & 63) & ~(kQ - 1))) & ((1ull << kQ) - 1ull)) != 0ull) { \
                    __builtin_amdgcn_wave_barrier(); \
                    _Pragma("unroll 1") for (int s_ = 0; s_ < kQ; ++s_) { \
                        const int peer_ = (tid & ~(kQ - 1)) + s_; \
                        const int pc_ = __shfl(long_ ? cnt_ : 0, ((tid & 63) & ~(kQ - 1)) + s_); \
                        _Pragma("unroll 1") for (int r_ = 0; r_ < pc_; ++r_) { \
                            const int2 t_ = s_runs_at(r_, peer_); \
                            if (t_.y - t_.x >= kShareRun) \
                                _Pragma("unroll 1") for (int j_ = t_.x + 4 * sub_q; j_ < t_.y; j_ += 4 * kQ) LISREG_WALK_FOUR(j_, t_.y - 1); \
                        } \
                    } \
                    __builtin_amdgcn_wave_barrier(); \
                } \
            } \
        } } while (0)

// kQ lanes per query (small batches, see k_assoc_walk): after a pass every lane holds the best five of ITS columns; a butterfly
// over the kQ lanes leaves all of them with the best five of the union, so the next pass starts from one common bound (and one
// common column box).  Entries travel as (distance, index) pairs; empty slots carry tau and never insert.
#define LISREG_GROUP_MERGE() do { \
        if (kQ > 1) { \
            _Pragma("unroll") for (int d_ = 1; d_ < kQ; d_ <<= 1) { \
                const float p0_ = __shfl_xor(b0, d_), p1_ = __shfl_xor(b1, d_), p2_ = __shfl_xor(b2, d_), p3_ = __shfl_xor(b3, d_), p4_ = __shfl_xor(b4, d_); \
                const int   j0_ = __shfl_xor(i0, d_), j1_ = __shfl_xor(i1, d_), j2_ = __shfl_xor(i2, d_), j3_ = __shfl_xor(i3, d_), j4_ = __shfl_xor(i4, d_); \
                LISREG_TRY(p0_, j0_); LISREG_TRY(p1_, j1_); LISREG_TRY(p2_, j2_); LISREG_TRY(p3_, j3_); LISREG_TRY(p4_, j4_); \
                LISREG_GROUP_TIES(p0_, j0_, p1_, j1_, p2_, j2_, p3_, j3_); if (kTies && i4 >= 0 && p4_ == b4 && j4_ != i4) tie = true; \
            } } } while (0)

// A second candidate for a query's anchor while the pose still moves by decimetres per iteration (GN iterations 1 .. P.cell_anchor_until):
// the nearest of up to four points out of the z-window [hz - 1, hz + 1] of the query's own (x, y) column of the grid — one contiguous
// run of the cell-sorted array, two table reads.  Any point will do for exactness (the scan certifies against whatever anchor it is
// given); a near one keeps the certificate radius c5 + d_a small.  -1: nothing there.
__device__ __forceinline__ int cell_anchor(const GridIndex& g, gptr_i32 cells, gptr_f4 pts, float qx, float qy, float qz, float& d2)
{
    d2 = 3.0e38f;
    const int hx = grid_coord(qx, g.ox, g.inv_cell), hy = grid_coord(qy, g.oy, g.inv_cell), hz = grid_coord(qz, g.oz, g.inv_cell);
    if (hx < 0 || hx >= g.nx || hy < 0 || hy >= g.ny || hz < -1 || hz > g.nz) return -1;      // (a non-finite query saturates: out)
    const int z0 = min(max(hz - 1, 0), g.nz - 1), z1 = min(max(hz + 1, 0), g.nz - 1);
    const int base = (hx * g.ny + hy) * g.nz;
    const int js = cells[base + z0], je = cells[base + z1 + 1];
    if (js >= je) return -1;
    const int n = je - js;
    const int p1 = js + (n >> 2), p2 = js + (n >> 1), p3 = je - 1;
    const v3f c0 = LISREG_LD3(pts, js), c1 = LISREG_LD3(pts, p1), c2 = LISREG_LD3(pts, p2), c3 = LISREG_LD3(pts, p3);
    const float ax = qx - c0.x, ay = qy - c0.y, az = qz - c0.z, bx = qx - c1.x, by = qy - c1.y, bz = qz - c1.z;
    const float ex = qx - c2.x, ey = qy - c2.y, ez = qz - c2.z, fx = qx - c3.x, fy = qy - c3.y, fz = qz - c3.z;
    const float d0 = ax * ax + ay * ay + az * az, d1 = bx * bx + by * by + bz * bz;
    const float e2 = ex * ex + ey * ey + ez * ez, d3 = fx * fx + fy * fy + fz * fz;
    int a = js; float d = d0;
    if (d1 < d) { d = d1; a = p1; }
    if (e2 < d) { d = e2; a = p2; }
    if (d3 < d) { d = d3; a = p3; }
    d2 = d;
    return a;
}

// Graph scan (search_mode 3, GN iterations >= 1).  The target carries a k-NN graph (lisreg_index.hip: kGraphK nearest other
// points per point, ascending, plus the coverage radius rho with "|x - a| < rho(a) => x is listed").  The query keeps ONE id
// from the last iteration — its nearest neighbour, the anchor a — and scans {a} + list(a) in list order.  With d_a = |q - a|
// and c5 = the current 5th-best distance (sqrt(tau) while fewer than five are known), any point not yet seen is at least
// l - d_a away, l being the list distance reached; so the scan stops at the first entry with l > c5 + d_a, and an exhausted
// list certifies through rho(a) > c5 + d_a.  Either way the five kept are exactly the five nearest inside sqrt(tau) (triangle
// inequality; kEps absorbs the rounding of the recomputed list distances).  No certificate -> hop to the nearest point found
// and rescan; still none after `graph_hops` lists -> the cell walk, seeded with what was found.  Same sets as the walk, at
// ~10 candidate tests per query (wave maximum ~21) instead of ~27 plus ~4 column probes, with no per-column address work.

// insertion without the duplicate test: inside ONE neighbour list (plus its anchor) every id occurs once
#define LISREG_TRY_ND(d2_, j_) do { if ((d2_) < b4) LISREG_INSERT((d2_), (j_)); } while (0)

// one group of four list entries E0..E3 = (x, y, z, id bits) straight from the anchor's row — no id -> point gather: the row carries the
// neighbours' coordinates (padded entries: the anchor's own coordinates, id -1).  Candidates are tested against the running five best;
// the scan stops once the list has moved past c5 + d_a.
#define LISREG_GRAPH_GROUP(E0, E1, E2, E3, TRYM) do { \
        const int k0_ = __float_as_int((E0).w), k1_ = __float_as_int((E1).w), k2_ = __float_as_int((E2).w), k3_ = __float_as_int((E3).w); \
        const float ax_ = qx - (E0).x, ay_ = qy - (E0).y, az_ = qz - (E0).z; \
        const float bx_ = qx - (E1).x, by_ = qy - (E1).y, bz_ = qz - (E1).z; \
        const float gx_ = qx - (E2).x, gy_ = qy - (E2).y, gz_ = qz - (E2).z; \
        const float hx_ = qx - (E3).x, hy_ = qy - (E3).y, hz_ = qz - (E3).z; \
        const float e0_ = ax_ * ax_ + ay_ * ay_ + az_ * az_; \
        const float e1_ = k1_ < 0 ? 3.0e38f : bx_ * bx_ + by_ * by_ + bz_ * bz_; \
        const float e2_ = k2_ < 0 ? 3.0e38f : gx_ * gx_ + gy_ * gy_ + gz_ * gz_; \
        const float e3_ = k3_ < 0 ? 3.0e38f : hx_ * hx_ + hy_ * hy_ + hz_ * hz_; \
        /* list distance of the group's LAST entry: the list ascends (the build computed these very values, same expression), so it \
           is the group's largest; a padded last entry aliases the anchor (0) and then the list ends here: rho decides */ \
        const float tx_ = ap_.x - (E3).x, ty_ = ap_.y - (E3).y, tz_ = ap_.z - (E3).z; \
        const float l3_ = tx_ * tx_ + ty_ * ty_ + tz_ * tz_; \
        if (LISREG_GATE(fminf(fminf(e0_, e1_), fminf(e2_, e3_)))) { \
            TRYM(e0_, k0_); TRYM(e1_, k1_); TRYM(e2_, k2_); TRYM(e3_, k3_); \
            LISREG_GROUP_TIES(e0_, k0_, e1_, k1_, e2_, k2_, e3_, k3_); \
            const float thr_ = __builtin_amdgcn_sqrtf(b4) * 1.0001f + da_; thr2_ = thr_ * thr_; \
        } \
        if (l3_ > thr2_) stop_ = true; } while (0)

// One list: the first group, the row's (rho^2, count) and the anchor point are fetched together; on the FIRST list of a query, anchor
// + first four candidates seed the five-best list through a 9-comparator network and the inserts need no duplicate test; later groups
// stream out of the same row (consecutive 64-byte pieces).  Leaves stop_ = "the five kept are certified".
#define LISREG_GRAPH_HOP(FIRST) \
            const gptr_f4 R_ = (gptr_f4)(nbr + (size_t)a_ * kGraphK); \
            const v4f r0_ = R_[0], r1_ = R_[1], r2_ = R_[2], r3_ = R_[3]; \
            const v2f am_ = meta[a_]; \
            const v4f ap_ = pts[a_]; \
            const float rho2_ = am_.x; \
            const int cnt_ = __float_as_int(am_.y); \
            const float ux_ = qx - ap_.x, uy_ = qy - ap_.y, uz_ = qz - ap_.z; \
            const float da2_ = ux_ * ux_ + uy_ * uy_ + uz_ * uz_; \
            const float da_ = __builtin_amdgcn_sqrtf(da2_) * 1.0001f + kEps; \
            stop_ = false; \
            float thr2_; \
            if ((FIRST) && cnt_ >= 4) { \
                /* anchor + the first four entries: all distinct, nothing in the list yet -> sort the five and take what is inside tau */ \
                float sd[5]; int sid[5] = { a_, __float_as_int(r0_.w), __float_as_int(r1_.w), __float_as_int(r2_.w), __float_as_int(r3_.w) }; \
                sd[0] = da2_; \
                { const float x_ = qx - r0_.x, y_ = qy - r0_.y, z_ = qz - r0_.z; sd[1] = x_ * x_ + y_ * y_ + z_ * z_; } \
                { const float x_ = qx - r1_.x, y_ = qy - r1_.y, z_ = qz - r1_.z; sd[2] = x_ * x_ + y_ * y_ + z_ * z_; } \
                { const float x_ = qx - r2_.x, y_ = qy - r2_.y, z_ = qz - r2_.z; sd[3] = x_ * x_ + y_ * y_ + z_ * z_; } \
                { const float x_ = qx - r3_.x, y_ = qy - r3_.y, z_ = qz - r3_.z; sd[4] = x_ * x_ + y_ * y_ + z_ * z_; } \
                const float lx_ = ap_.x - r3_.x, ly_ = ap_.y - r3_.y, lz_ = ap_.z - r3_.z; \
                const float l3_ = lx_ * lx_ + ly_ * ly_ + lz_ * lz_; \
                LISREG_CE5(0, 1); LISREG_CE5(3, 4); LISREG_CE5(2, 4); LISREG_CE5(2, 3); LISREG_CE5(0, 3); \
                LISREG_CE5(0, 2); LISREG_CE5(1, 4); LISREG_CE5(1, 3); LISREG_CE5(1, 2); \
                if (__builtin_amdgcn_ballot_w64(!(sd[4] < P.tau)) == 0) {           /* the usual case: all five inside tau in every lane */ \
                    b0 = sd[0]; b1 = sd[1]; b2 = sd[2]; b3 = sd[3]; b4 = sd[4]; i0 = sid[0]; i1 = sid[1]; i2 = sid[2]; i3 = sid[3]; i4 = sid[4]; \
                } else { \
                    b0 = fminf(sd[0], P.tau); b1 = fminf(sd[1], P.tau); b2 = fminf(sd[2], P.tau); b3 = fminf(sd[3], P.tau); b4 = fminf(sd[4], P.tau); \
                    i0 = sd[0] < P.tau ? sid[0] : -1; i1 = sd[1] < P.tau ? sid[1] : -1; i2 = sd[2] < P.tau ? sid[2] : -1; \
                    i3 = sd[3] < P.tau ? sid[3] : -1; i4 = sd[4] < P.tau ? sid[4] : -1; \
                } \
                LISREG_LIST_TIES(); \
                const float thr_ = __builtin_amdgcn_sqrtf(b4) * 1.0001f + da_; thr2_ = thr_ * thr_; \
                if (l3_ > thr2_) stop_ = true; \
            } else { \
                if (FIRST) LISREG_LIST_INIT(); \
                if (kTies && i4 >= 0 && da2_ == b4 && a_ != i4) tie = true; \
                LISREG_TRY(da2_, a_); \
                const float thr_ = __builtin_amdgcn_sqrtf(b4) * 1.0001f + da_; thr2_ = thr_ * thr_; \
                if (cnt_ > 0) LISREG_GRAPH_GROUP(r0_, r1_, r2_, r3_, LISREG_TRY); \
            } \
            if (FIRST) { \
                _Pragma("unroll 1") for (int g_ = 1; !stop_ && 4 * g_ < cnt_; ++g_) { \
                    const v4f e0g_ = R_[4 * g_], e1g_ = R_[4 * g_ + 1], e2g_ = R_[4 * g_ + 2], e3g_ = R_[4 * g_ + 3]; \
                    LISREG_GRAPH_GROUP(e0g_, e1g_, e2g_, e3g_, LISREG_TRY_ND); \
                } \
            } else { \
                _Pragma("unroll 1") for (int g_ = 1; !stop_ && 4 * g_ < cnt_; ++g_) { \
                    const v4f e0g_ = R_[4 * g_], e1g_ = R_[4 * g_ + 1], e2g_ = R_[4 * g_ + 2], e3g_ = R_[4 * g_ + 3]; \
                    LISREG_GRAPH_GROUP(e0g_, e1g_, e2g_, e3g_, LISREG_TRY); \
                } \
            } \
            if (!stop_) stop_ = rho2_ > thr2_;                 /* list exhausted: the coverage radius decides */

// The first list is straight-line code (in the steady state it is the only one); the hops after it — taken only without a certificate —
// are a loop of their own, so that the five-best list is not a loop-carried value of the common path (the compiler kept it in a
// second register set and copied 11 registers per pass, plus three rounds of initialisation).
#define LISREG_GRAPH_SCAN() do { \
        int a_ = anchor; \
        bool stop_; \
        { LISREG_GRAPH_HOP(true) } \
        if (stop_) certified = true; \
        else { \
            _Pragma("unroll 1") for (int hop_ = 1; hop_ < graph_hops; ++hop_) { \
                if (i0 == a_ || i0 < 0) break;                 /* nowhere better to hop to */ \
                a_ = i0; \
                { LISREG_GRAPH_HOP(false) } \
                if (stop_) { certified = true; break; } \
            } \
        } } while (0)

// Cell rows (search_mode 5, every GN iteration).  The row is anchored at the centre m of the grid cell — or of the octant of it — the query
// falls into (lisreg_index.hip, k_crow_build), so d_m = |q - m| is bounded by the cell size whatever the query's distance from the surface:
// the same certificate as the graph scan (stop at the first entry farther from m than c5 + d_m; an exhausted list certifies through
// rho(m) > c5 + d_m), no anchor carried between iterations, no anchor point to fetch, no hop.  No certificate -> the cell walk, seeded.
#define LISREG_CELL_SCAN() do { \
            const gptr_f4 R_ = (gptr_f4)(crow + (size_t)row_ * kGraphK); \
            const v4f r0_ = R_[0], r1_ = R_[1], r2_ = R_[2], r3_ = R_[3]; \
            const float rho2_ = am_.x; \
            const int cnt_ = __float_as_int(am_.y); \
            v4f ap_; ap_.x = mx_; ap_.y = my_; ap_.z = mz_; ap_.w = 0.f; \
            const float ux_ = qx - ap_.x, uy_ = qy - ap_.y, uz_ = qz - ap_.z; \
            const float da2_ = ux_ * ux_ + uy_ * uy_ + uz_ * uz_; \
            const float da_ = __builtin_amdgcn_sqrtf(da2_) * 1.0001f + kEps; \
            bool stop_ = false; \
            float thr2_; \
            if (cnt_ >= 4) { \
                /* the first four entries: all distinct, nothing in the list yet -> sort them and take what is inside tau */ \
                float sd[4]; int sid[4] = { __float_as_int(r0_.w), __float_as_int(r1_.w), __float_as_int(r2_.w), __float_as_int(r3_.w) }; \
                { const float x_ = qx - r0_.x, y_ = qy - r0_.y, z_ = qz - r0_.z; sd[0] = x_ * x_ + y_ * y_ + z_ * z_; } \
                { const float x_ = qx - r1_.x, y_ = qy - r1_.y, z_ = qz - r1_.z; sd[1] = x_ * x_ + y_ * y_ + z_ * z_; } \
                { const float x_ = qx - r2_.x, y_ = qy - r2_.y, z_ = qz - r2_.z; sd[2] = x_ * x_ + y_ * y_ + z_ * z_; } \
                { const float x_ = qx - r3_.x, y_ = qy - r3_.y, z_ = qz - r3_.z; sd[3] = x_ * x_ + y_ * y_ + z_ * z_; } \
                const float lx_ = ap_.x - r3_.x, ly_ = ap_.y - r3_.y, lz_ = ap_.z - r3_.z; \
                const float l3_ = lx_ * lx_ + ly_ * ly_ + lz_ * lz_; \
                LISREG_CE5(0, 1); LISREG_CE5(2, 3); LISREG_CE5(0, 2); LISREG_CE5(1, 3); LISREG_CE5(1, 2); \
                if (__builtin_amdgcn_ballot_w64(!(sd[3] < P.tau)) == 0) {           /* the usual case: all four inside tau in every lane */ \
                    b0 = sd[0]; b1 = sd[1]; b2 = sd[2]; b3 = sd[3]; i0 = sid[0]; i1 = sid[1]; i2 = sid[2]; i3 = sid[3]; \
                } else { \
                    b0 = fminf(sd[0], P.tau); b1 = fminf(sd[1], P.tau); b2 = fminf(sd[2], P.tau); b3 = fminf(sd[3], P.tau); \
                    i0 = sd[0] < P.tau ? sid[0] : -1; i1 = sd[1] < P.tau ? sid[1] : -1; i2 = sd[2] < P.tau ? sid[2] : -1; \
                    i3 = sd[3] < P.tau ? sid[3] : -1; \
                } \
                b4 = P.tau; i4 = -1; \
                LISREG_LIST_TIES(); \
                const float thr_ = __builtin_amdgcn_sqrtf(b4) * 1.0001f + da_; thr2_ = thr_ * thr_; \
                if (l3_ > thr2_) stop_ = true; \
            } else { \
                LISREG_LIST_INIT(); \
                const float thr_ = __builtin_amdgcn_sqrtf(b4) * 1.0001f + da_; thr2_ = thr_ * thr_; \
                if (cnt_ > 0) LISREG_GRAPH_GROUP(r0_, r1_, r2_, r3_, LISREG_TRY_ND); \
            } \
            _Pragma("unroll 1") for (int g_ = 1; !stop_ && 4 * g_ < cnt_ && LISREG_XP_SCAN_GROUPS(g_); ++g_) { \
                const v4f e0g_ = R_[4 * g_], e1g_ = R_[4 * g_ + 1], e2g_ = R_[4 * g_ + 2], e3g_ = R_[4 * g_ + 3]; \
                LISREG_GRAPH_GROUP(e0g_, e1g_, e2g_, e3g_, LISREG_TRY_ND); \
            } \
            if (!stop_) stop_ = rho2_ > thr2_;                 /* list exhausted: the coverage radius decides */ \
            certified = stop_ || LISREG_XP_ALWAYS_CERTIFIED; } while (0)

#ifdef LISREG_XP_NOSCAN         /* timing experiment (wrong results): two groups of the row, no stop test, never a walk */
#define LISREG_XP_SCAN_GROUPS(g_) ((g_) < 2)
#define LISREG_XP_ALWAYS_CERTIFIED true
#else
#define LISREG_XP_SCAN_GROUPS(g_) true
#define LISREG_XP_ALWAYS_CERTIFIED false
#endif
#define LISREG_CE5(a, b) do { const bool sw_ = sd[b] < sd[a]; const float ta_ = sd[a]; const int ia_ = sid[a]; \
                              sd[a] = sw_ ? sd[b] : ta_; sd[b] = sw_ ? ta_ : sd[b]; sid[a] = sw_ ? sid[b] : ia_; sid[b] = sw_ ? ia_ : sid[b]; } while (0)

// Canonical five of a lane that noted a tie: every point with d^2 <= the final 5th-best distance (there are five or a few more), ordered
// by (d^2, original index of the point in the caller's cloud), the first five kept.  Plain loops over the cells the ball touches; the
// squared distance is formed by the same expression as in the search, so it has the same bits.  With kQ lanes per query every lane of
// the group runs it (the note is shared first), and all arrive at the same list.
#define LISREG_CANON_INSERT(d2_, o_, j_) do { \
        const bool c3_ = (d2_) < b3 || ((d2_) == b3 && (o_) < o3), c2_ = (d2_) < b2 || ((d2_) == b2 && (o_) < o2); \
        const bool c1_ = (d2_) < b1 || ((d2_) == b1 && (o_) < o1), c0_ = (d2_) < b0 || ((d2_) == b0 && (o_) < o0); \
        b4 = c3_ ? b3 : (d2_);               i4 = c3_ ? i3 : (j_);               o4 = c3_ ? o3 : (o_); \
        b3 = c3_ ? (c2_ ? b2 : (d2_)) : b3;  i3 = c3_ ? (c2_ ? i2 : (j_)) : i3;  o3 = c3_ ? (c2_ ? o2 : (o_)) : o3; \
        b2 = c2_ ? (c1_ ? b1 : (d2_)) : b2;  i2 = c2_ ? (c1_ ? i1 : (j_)) : i2;  o2 = c2_ ? (c1_ ? o1 : (o_)) : o2; \
        b1 = c1_ ? (c0_ ? b0 : (d2_)) : b1;  i1 = c1_ ? (c0_ ? i0 : (j_)) : i1;  o1 = c1_ ? (c0_ ? o0 : (o_)) : o1; \
        b0 = c0_ ? (d2_) : b0;               i0 = c0_ ? (j_) : i0;               o0 = c0_ ? (o_) : o0; } while (0)
#define LISREG_CANONICAL_FIVE() do { if (kTies) { \
        /* no short-circuit here: a lane that skipped the exchange because its own note is set would be INACTIVE in it, and the lanes \
           reading from it would get nothing — the one lane that saw the tie is exactly the one the others have to hear from */ \
        if (kQ > 1) { _Pragma("unroll") for (int d_ = 1; d_ < kQ; d_ <<= 1) { const int o_ = __shfl_xor((int)tie, d_); tie = (int)tie | (int)(o_ != 0); } } \
        LISREG_LIST_TIES(); \
        if (tie && i4 >= 0) { \
            const float lim_ = b4; \
            const float rad_ = __builtin_amdgcn_sqrtf(lim_) * 1.0001f + kEps; \
            const int cx0_ = max(grid_coord(qx - rad_, g.ox, g.inv_cell), 0), cx1_ = min(grid_coord(qx + rad_, g.ox, g.inv_cell), g.nx - 1); \
            const int cy0_ = max(grid_coord(qy - rad_, g.oy, g.inv_cell), 0), cy1_ = min(grid_coord(qy + rad_, g.oy, g.inv_cell), g.ny - 1); \
            const int cz0_ = max(grid_coord(qz - rad_, g.oz, g.inv_cell), 0), cz1_ = min(grid_coord(qz + rad_, g.oz, g.inv_cell), g.nz - 1); \
            b0 = b1 = b2 = b3 = b4 = 3.0e38f; i0 = i1 = i2 = i3 = i4 = -1; \
            int o0 = 0x7fffffff, o1 = 0x7fffffff, o2 = 0x7fffffff, o3 = 0x7fffffff, o4 = 0x7fffffff; \
            if (cz0_ <= cz1_) \
            _Pragma("unroll 1") for (int ix_ = cx0_; ix_ <= cx1_; ++ix_) \
            _Pragma("unroll 1") for (int iy_ = cy0_; iy_ <= cy1_; ++iy_) { \
                const int base_ = (ix_ * g.ny + iy_) * g.nz; \
                const int js_ = cells[base_ + cz0_], je_ = cells[base_ + cz1_ + 1]; \
                _Pragma("unroll 1") for (int j_ = js_; j_ < je_; ++j_) { \
                    const v4f c_ = pts[j_]; \
                    const float ax_ = qx - c_.x, ay_ = qy - c_.y, az_ = qz - c_.z; \
                    const float e_ = ax_ * ax_ + ay_ * ay_ + az_ * az_; \
                    const int o_ = __float_as_int(c_.w); \
                    if (e_ <= lim_ && (e_ < b4 || (e_ == b4 && o_ < o4))) LISREG_CANON_INSERT(e_, o_, j_); \
                } \
            } \
            (void)o4; \
        } } } while (0)

// kQ = lanes per query.  1 for big batches (every lane its own query: throughput).  A single odometry-sized registration is a
// few hundred waves on a chip with 8192 wave slots, i.e. one wave per SIMD with nothing to hide its dependent cell -> candidate
// loads behind: there kQ = 8 lanes share one query (columns dealt round-robin, five-best lists merged by butterfly), which cuts
// the serial chain of the walk by the same factor.  The five neighbours, and everything computed from them, are identical.
// kShare (kQ > 1 only): long candidate runs are shared by the kQ lanes of a query (LISREG_WALK_LIST); that variant needs ~90 registers, so it
// runs four waves per SIMD — right for the few thousand wavefronts of a downsampled frame, wrong for a full 64 x 1800 sweep (14 k wavefronts),
// which keeps the 64-register variant without sharing.  The launcher picks by size; the five neighbours are the same either way.
template <bool kWide, int kGraph, int kQ, bool kTies, bool kShare = false>
__global__ __launch_bounds__(kBlockQ) __attribute__((amdgpu_waves_per_eu(kShare ? 4 : 8, kShare ? 4 : 8))) void k_assoc_walk(const BlockDesc* __restrict__ blocks,
                                                        const Segment* __restrict__ segs,
                                                        const GridIndex* __restrict__ grids,
                                                        const ItemState* __restrict__ items, const DevParams P,
                                                        const float4* __restrict__ sorted_all,
                                                        int* __restrict__ nn_, int n_elems, float first_pass_r2,
                                                        int graph_hops, unsigned long long* __restrict__ counters,
                                                        int* __restrict__ dbg_nn, float4* __restrict__ coef, int* __restrict__ coef_ok,
                                                        double* __restrict__ partials, const int* __restrict__ xcd_order)
{
    // per-lane lists of candidate runs for the flattened walk, one 4-KB region per wavefront (s_runs[wave][entry][lane]); the same memory
    // serves the reduction once a wavefront's search is over (row_and_reduce: every wavefront works in its own region until the barrier)
    __shared__ __attribute__((aligned(16))) int2 s_runs_w[kBlockQ / 64][kWalkCap][64];
    static_assert(sizeof(int2) * kWalkCap * 64 == sizeof(float) * kRedWaveFloats, "one region per wavefront");
    float* s_red = reinterpret_cast<float*>(&s_runs_w[0][0][0]);
#define s_runs_at(r_, t_) s_runs_w[(t_) >> 6][(r_)][(t_) & 63]
    constexpr float kEps = 1e-3f;

    const int tid = threadIdx.x;
    // XCD-aware dispatch order (launch_xcd_order below): workgroup p runs on XCD p % 8; the table (blocks ranked by sector) hands it a
    // block of "its" eighth of the ranking.  Partial rows stay addressed by the block id, so the sums do not depend on the order.
    int bid = blockIdx.x;
    if (xcd_order) {
        // XCD x owns dispatch positions x, x + 8, ... (m + 1 of them for x < t, else m) and the ranks [x * m + min(x, t), ...) of the sector order
        const int m = gridDim.x >> 3, t = gridDim.x & 7, x = blockIdx.x & 7;
        bid = xcd_order[x * m + min(x, t) + (int)(blockIdx.x >> 3)];
    }
    const BlockDesc bd = blocks[bid];
    const ItemState* it = &items[bd.item];
    if (it->done) return;
    const Segment sg = segs[bd.seg];
    const GridIndex g = grids[sg.target];
    double* out = partials + (size_t)bid * kNumAcc;
    if (g.n < 5) {
        if (kQ == 1 && tid < kNumAcc) out[tid] = 0.0;      // kQ > 1: k_rows_reduce owns the partial rows
        return;
    }
    const gptr_f4 pts = (gptr_f4)g.pts;
    const gptr_i32 cells = (gptr_i32)g.cell_start;
    const gptr_i32w nn = (gptr_i32w)nn_;
    const gptr_i4w nn4 = (gptr_i4w)nn_;
    const float* M = it->M;            // trans2Affine3f(T), cached by the solve kernel (uniform -> SGPRs)

    const int sub_q = tid & (kQ - 1); (void)sub_q;
    const bool valid = tid / kQ < bd.count;
    const int qflat = sg.flat_base + bd.start + tid / kQ;
    // sources: the tile-sorted copy if the batch was sorted, else the caller's own records (no flattening copy).  Both are
    // addressed with the batch-wide position qflat, so that one register serves the source and the seed arrays.
    const float4* qsrc = sorted_all ? sorted_all : (const float4*)((uintptr_t)sg.src - (uintptr_t)sg.flat_base * sizeof(float4));
    float qx, qy, qz;
#if LISREG_KEEP_SRC
    float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f);
#endif
    {
#if !LISREG_KEEP_SRC
        float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f);
#endif
        if (valid) q0 = qsrc[qflat];
        qx = M[0] * q0.x + M[1] * q0.y + M[2] * q0.z + M[3];
        qy = M[4] * q0.x + M[5] * q0.y + M[6] * q0.z + M[7];
        qz = M[8] * q0.x + M[9] * q0.y + M[10] * q0.z + M[11];
    }

    // the five-best list (ascending).  Not initialised here: the common path (graph scan, first list with >= 4 entries) overwrites all
    // ten registers, and an initialisation up front is executed by every wavefront (the compiler even emitted it twice)
    float b0, b1, b2, b3, b4;
    int   i0, i1, i2, i3, i4;
    bool  tie = false;                                     // kTies: an equal-distance pair that can matter was met
#define LISREG_LIST_INIT() do { b0 = b1 = b2 = b3 = b4 = P.tau; i0 = i1 = i2 = i3 = i4 = -1; } while (0)
    if (kGraph) {
        // search_mode 3: one anchor id per query instead of five seeds; graph scan first, cell walk only without a certificate
        bool need_walk = valid, scanned = false;
        if (kGraph == 2) {
            // search_mode 5: the row of the cell (or octant) the query is in — nothing carried between iterations
            int row_ = -1;
            float mx_ = 0.f, my_ = 0.f, mz_ = 0.f;
            if (valid && g.crow_tab) {
                // a query outside the grid (the target's bounding box: a wall seen from a pose that is still off puts half of its points
                // there) takes the row of the nearest boundary cell — any row is valid for any query, the certificate works with the
                // actual |q - m|
                // (a NaN query lands in some cell and never certifies: every comparison with its distances is false)
                const float tx = (qx - g.ox) * g.inv_cell, ty = (qy - g.oy) * g.inv_cell, tz = (qz - g.oz) * g.inv_cell;
                const int gx = (int)floorf(tx), gy = (int)floorf(ty), gz = (int)floorf(tz);
                const int hx = min(max(gx, 0), g.nx - 1), hy = min(max(gy, 0), g.ny - 1), hz = min(max(gz, 0), g.nz - 1);
                const int v = ((gptr_i32)g.crow_tab)[(hx * g.ny + hy) * g.nz + hz];
                if (P.reach_miss) {                        // rows filtered by the batch's query marks: how many queries find their cell without them
                    const unsigned long long mm = __ballot(v == -1);
                    if (mm != 0ull && (int)(__ffsll((long long)mm) - 1) == (tid & 63)) atomicAdd(P.reach_miss, __popcll(mm));
                }
                if (v >= 0) {
                    float fx = 0.5f, fy = 0.5f, fz = 0.5f;
                    row_ = v >> 8;
                    {
                        const bool ux = tx - (float)hx >= 0.5f, uy = ty - (float)hy >= 0.5f, uz = tz - (float)hz >= 0.5f;
                        const int oct = (int)ux + 2 * (int)uy + 4 * (int)uz, bit = 1 << oct;
                        if (v & bit) {                     // the octant has a row of its own
                            fx = ux ? 0.75f : 0.25f; fy = uy ? 0.75f : 0.25f; fz = uz ? 0.75f : 0.25f;
                            row_ += 1 + __popc(v & (bit - 1));
                        }
                    }
                    mx_ = crow_centre(g.ox, g.cell, hx, fx); my_ = crow_centre(g.oy, g.cell, hy, fy); mz_ = crow_centre(g.oz, g.cell, hz, fz);
                } else if (v == -2 && hx == gx && hy == gy && hz == gz && P.tau <= (2.f * g.cell - 2.f * kEps) * (2.f * g.cell - 2.f * kEps)) {
                    need_walk = false;                     // nothing within two cells of this one: no neighbour inside sqrt(tau)
                }
            }
            const gptr_f4 crow = (gptr_f4)g.crow;
            const gptr_f2 cmeta = (gptr_f2)g.crow_meta;
            const bool scan_ = row_ >= 0;
            v2f am_; am_.x = 0.f; am_.y = 0.f;
            if (scan_) am_ = cmeta[row_];                  // (rho^2, count)
            if (scan_) {
                bool certified = false;
                scanned = true;
                LISREG_CELL_SCAN();
                need_walk = !certified;
            }
        } else
        if (valid && it->iter > 0 && g.nbr) {
            int anchor = nn[qflat];
            if (it->iter <= P.cell_anchor_until) {
                float cd2;
                const int ca = cell_anchor(g, cells, pts, qx, qy, qz, cd2);
                if (ca >= 0) {
                    float od2 = 3.0e38f;
                    if (anchor >= 0) { const v3f ap = LISREG_LD3(pts, anchor); const float x = qx - ap.x, y = qy - ap.y, z = qz - ap.z; od2 = x * x + y * y + z * z; }
                    if (cd2 < od2) anchor = ca;
                }
            }
            if (anchor >= 0) {
                const gptr_f4 nbr = (gptr_f4)g.nbr;
                const gptr_f2 meta = (gptr_f2)g.nbr_meta;
                bool certified = false;
                scanned = true;
                LISREG_GRAPH_SCAN();
                need_walk = !certified;
            }
        }
        if (!scanned) LISREG_LIST_INIT();
        if (counters && (kGraph == 2 || it->iter > 0) && it->iter < 32) {       // diagnostics: lanes that fell back to the cell walk
            const int nw = __popcll(__ballot(need_walk)), nv = __popcll(__ballot(valid));
            if ((tid & 63) == 0) {
                atomicAdd(&counters[it->iter], ((unsigned long long)nw << 32) | (unsigned long long)nv);
                atomicAdd(&counters[64 + it->iter], ((unsigned long long)(nw > 0) << 32) | (unsigned long long)(nv > 0));   // waves
            }
        }
        if (need_walk) {
            int sx0_ = 1, sx1_ = 0, sy0_ = 1, sy1_ = 0;
            if (kWide) {
                LISREG_WALK_LIST(3.0e38f, true, false);
                LISREG_WALK_LIST(3.0e38f, false, true);
            } else {
                LISREG_WALK_LIST(3.0e38f, false, false);
            }
            (void)sx0_; (void)sx1_; (void)sy0_; (void)sy1_;
        }
        LISREG_CANONICAL_FIVE();
        if (kGraph == 1 && valid) nn[qflat] = i0;              // next iteration's anchor: the nearest neighbour (-1: none)
    } else if (!valid) {
        LISREG_LIST_INIT();
    } else {
        LISREG_LIST_INIT();
        bool seeded = false;
        if (it->iter > 0) {
            // seeds: last iteration's neighbours bound the new 5th-nearest distance (any 5 points do), so the walk
            // below only has to look inside that radius.  Exactness does not depend on the seeds being right.
            // seed layout of this kernel: ids 0..3 as one 16-byte record per query, the fifth id (-1 = no valid set) after them;
            // both loads are issued together (one latency, two instructions instead of five)
            const v4i s03 = nn4[qflat];
            const int s4 = nn[4 * (size_t)n_elems + qflat];
            if (s4 >= 0) {
                int sid[5] = { s03.x, s03.y, s03.z, s03.w, s4 };
                v4f sp[5];
#pragma unroll
                for (int k = 0; k < 5; ++k) sp[k] = pts[sid[k]];
                float sd[5];
#pragma unroll
                for (int k = 0; k < 5; ++k) {
                    const float ex = qx - sp[k].x, ey = qy - sp[k].y, ez = qz - sp[k].z;
                    sd[k] = ex * ex + ey * ey + ez * ez;
                }
                // the seeds are nearly sorted already; a 9-comparator network orders (distance, index) pairs
#define LISREG_CE(a, b) do { const bool sw_ = sd[b] < sd[a]; const float ta_ = sd[a]; const int ia_ = sid[a]; \
                             sd[a] = sw_ ? sd[b] : ta_; sd[b] = sw_ ? ta_ : sd[b]; sid[a] = sw_ ? sid[b] : ia_; sid[b] = sw_ ? ia_ : sid[b]; } while (0)
                LISREG_CE(0, 1); LISREG_CE(3, 4); LISREG_CE(2, 4); LISREG_CE(2, 3); LISREG_CE(0, 3);
                LISREG_CE(0, 2); LISREG_CE(1, 4); LISREG_CE(1, 3); LISREG_CE(1, 2);
#undef LISREG_CE
                if (sd[4] < P.tau) {
                    b0 = sd[0]; b1 = sd[1]; b2 = sd[2]; b3 = sd[3]; b4 = sd[4];
                    i0 = sid[0]; i1 = sid[1]; i2 = sid[2]; i3 = sid[3]; i4 = sid[4];
                } else {
#pragma unroll
                    for (int k = 0; k < 5; ++k) if (sd[k] < b4) LISREG_INSERT(sd[k], sid[k]);
                }
                seeded = true;
                LISREG_LIST_TIES();
            }
        }
        int sx0_ = 1, sx1_ = 0, sy0_ = 1, sy1_ = 0;            // columns covered by an INNER pass (empty)
        if (!seeded && kWide) {
            // no seeds (GN iteration 0): the 3 x 3 columns around the query over the full tau-deep z-range first, then only
            // the columns outside that box which the bound still reaches.  A radius-limited first pass (450 ... 800 mm, with the
            // skip only when it settles five neighbours) measured 1-2 % slower than this.
            LISREG_WALK_LIST(3.0e38f, true, false);
            LISREG_GROUP_MERGE();
            LISREG_WALK_LIST(3.0e38f, false, true);
        } else if (!seeded) {
            LISREG_WALK_LIST(first_pass_r2, false, false);      // tight first pass establishes a bound cheaply
            LISREG_GROUP_MERGE();
            if (!(b4 <= first_pass_r2)) LISREG_WALK_LIST(3.0e38f, false, false);
        } else if (kWide) {
            LISREG_WALK_LIST(3.0e38f, true, false);             // centre first ...
            LISREG_GROUP_MERGE();
            LISREG_WALK_LIST(3.0e38f, false, true);             // ... then whatever the tightened bound still reaches
        } else {
            LISREG_WALK_LIST(3.0e38f, false, false);
        }
        LISREG_GROUP_MERGE();
        (void)sx0_; (void)sx1_; (void)sy0_; (void)sy1_;
        LISREG_CANONICAL_FIVE();
        // remember the neighbours for the next iteration (slot 4 = -1 marks "no valid set").  Skipping this store
        // while the set is unchanged was measured twice: keeping the five ids live costs an occupancy step (8 -> 7,
        // 4 % slower); a one-register XOR signature keeps 8 waves but gains nothing — the kernel is not HBM-bound.
        if (kQ == 1 || sub_q == 0) {
            { v4i w; w.x = i0; w.y = i1; w.z = i2; w.w = i3; nn4[qflat] = w; }
            nn[4 * (size_t)n_elems + qflat] = i4;
        }
    }
    if (dbg_nn && valid && (kQ == 1 || sub_q == 0)) {     // "dump_neighbors" (tests): ORIGINAL indices of the five neighbours, -1 = none
        const int ids[5] = { i0, i1, i2, i3, i4 };
        int og[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) og[k] = ids[k] >= 0 ? __float_as_int(pts[ids[k]].w) : -1;
        // (this view only) a list that stayed INCOMPLETE — fewer than five points inside sqrt(tau): the query contributes nothing, and the
        // canonical re-selection of equal distances runs for complete lists only — still shows equal distances in original-index order, so
        // that the front-ends' dumps compare entry for entry (tests/frontend_sweep.py seed 60: two candidates 5e-7 apart in float64, equal in float)
        if (kTies && i4 < 0) {
            const float ds[4] = { b0, b1, b2, b3 };
#pragma unroll
            for (int pass = 0; pass < 3; ++pass)
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    if (og[k] >= 0 && og[k + 1] >= 0 && ds[k] == ds[k + 1] && og[k] > og[k + 1]) { const int t = og[k]; og[k] = og[k + 1]; og[k + 1] = t; }
        }
        bool same = true;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int o = og[k];
            same = same && dbg_nn[(size_t)k * n_elems + qflat] == o;
            dbg_nn[(size_t)k * n_elems + qflat] = o;
        }
        if (counters && it->iter > 0 && it->iter < 32) {       // diagnostics: queries whose ordered neighbour set did not change
            const unsigned long long m = __ballot(same);
            if (__ffsll((long long)__ballot(true)) - 1 == (tid & 63))
                atomicAdd(&counters[96 + it->iter], ((unsigned long long)__popcll(m) << 32) | (unsigned long long)(m == __ballot(true)));
        }
    }
    // the source record: one lane per query keeps it in registers across the search (round 5: the closed-form plane fit freed the registers —
    // 55-57 of 64 with it — and the second read, a dependent load at the end of a wavefront's chain, was 6 us of a 172 us launch);
    // eight lanes per query read it again (the sharing variant is at its register limit)
    float4 q4 = make_float4(0.f, 0.f, 0.f, 0.f);
#ifdef LISREG_XP_NOSRC2         /* timing experiment (wrong results): no second read of the source record */
    q4 = make_float4(qx, qy, qz, 0.f);
#else
#if LISREG_KEEP_SRC
    if (kQ == 1) q4 = q0;
    else
#endif
    if (valid) { const v4f t = __builtin_nontemporal_load((const v4f*)&qsrc[qflat]); q4 = make_float4(t.x, t.y, t.z, t.w); }
#endif
    if (kQ == 1) {
        residual_and_reduce(valid, i0, i1, i2, i3, i4, g, q4, qx, qy, qz, it->jk, P, sg.kind, s_red, out,
                                         dbg_nn ? dbg_nn + 5 * (size_t)n_elems + qflat : nullptr);
    } else {
        // kQ lanes per query: the coefficients go to memory and k_rows_reduce builds the partial rows with the SAME 256-query
        // workgroups and the SAME reduction tree as the kQ = 1 kernel, so the normal equations — hence every pose — are
        // bit-identical whichever variant a batch runs with (a frame gives the same result alone and inside a big batch)
        float cf[4];
        const bool ok = residual_coeffs(valid, i0, i1, i2, i3, i4, g, q4, qx, qy, qz, P, sg.kind, cf);
        if (valid && sub_q == 0) {
            coef[qflat] = make_float4(cf[0], cf[1], cf[2], cf[3]);
            coef_ok[qflat] = ok ? 1 : 0;
            if (dbg_nn) dbg_nn[5 * (size_t)n_elems + qflat] = ok ? 1 : 0;
        }
    }
}

// Second half of the kQ > 1 path: Jacobian rows and the partial normal equations from the stored coefficients, on the 256-query
// block descriptors — identical code and order to the tail of k_assoc_walk<.., 1>.
__global__ __launch_bounds__(kBlockQ) void k_rows_reduce(const BlockDesc* __restrict__ blocks, const Segment* __restrict__ segs,
                                                         const GridIndex* __restrict__ grids, const ItemState* __restrict__ items,
                                                         const DevParams P, const float4* __restrict__ sorted_all,
                                                         const float4* __restrict__ coef, const int* __restrict__ coef_ok,
                                                         double* __restrict__ partials)
{
    __shared__ __attribute__((aligned(16))) float s_red[4 * kRedWaveFloats];
    const int tid = threadIdx.x;
    const BlockDesc bd = blocks[blockIdx.x];
    const ItemState* it = &items[bd.item];
    if (it->done) return;
    const Segment sg = segs[bd.seg];
    double* out = partials + (size_t)blockIdx.x * kNumAcc;
    if (grids[sg.target].n < 5) {
        if (tid < kNumAcc) out[tid] = 0.0;
        return;
    }
    const bool valid = tid < bd.count;
    const int qflat = sg.flat_base + bd.start + tid;
    const float4* qsrc = sorted_all ? sorted_all : (const float4*)((uintptr_t)sg.src - (uintptr_t)sg.flat_base * sizeof(float4));
    float4 q4 = make_float4(0.f, 0.f, 0.f, 0.f), c4 = make_float4(0.f, 0.f, 0.f, 0.f);
    bool ok = false;
    if (valid) {
        const v4f t = __builtin_nontemporal_load((const v4f*)&qsrc[qflat]); q4 = make_float4(t.x, t.y, t.z, t.w);
        c4 = coef[qflat];
        ok = coef_ok[qflat] != 0;
    }
    const float cf[4] = { c4.x, c4.y, c4.z, c4.w };
    row_and_reduce(ok, cf, q4, it->jk, P, s_red, out);
}

__global__ __launch_bounds__(64) void k_test_fit_models(int kind, int n, const float* __restrict__ nb15, const float* __restrict__ q3,
                                                        DevParams P, float* __restrict__ out)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    float4 nb[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) nb[j] = make_float4(nb15[i * 15 + 3 * j], nb15[i * 15 + 3 * j + 1], nb15[i * 15 + 3 * j + 2], 0.f);
    const float qx = q3[3 * i], qy = q3[3 * i + 1], qz = q3[3 * i + 2];
    float cf[4] = { 0.f, 0.f, 0.f, 0.f };
    float* o = out + (size_t)i * 10;
    if (kind == 1) {
        float pa, pb, pc, pd;
        bool closed = false;
        if (!kExactArith && LISREG_PLANE_CLOSED) closed = plane5_closed(nb, pa, pb, pc, pd);
        const float4 m = surf_model(nb, P);
        const bool ok = surf_eval(m, qx, qy, qz, 1.f, P, cf);
        o[0] = m.x; o[1] = m.y; o[2] = m.z; o[3] = m.w; o[4] = closed ? 1.f : 0.f;
        o[5] = cf[0]; o[6] = cf[1]; o[7] = cf[2]; o[8] = cf[3]; o[9] = ok ? 1.f : 0.f;
    } else {
        float4 m0, m1;
        corner_model(nb, P, m0, m1);
        const bool ok = corner_eval(m0, m1, qx, qy, qz, 1.f, P, cf);
        o[0] = m0.x; o[1] = m0.y; o[2] = m0.z; o[3] = m1.x; o[4] = m1.y;
        o[5] = cf[0]; o[6] = cf[1]; o[7] = cf[2]; o[8] = cf[3]; o[9] = ok ? 1.f : 0.f;
    }
}

}  // namespace LISREG_ASSOC_NS

#if !LISREG_EXACT
// ---- XCD-aware dispatch order for shared-target batches (search_mode 3) ----------------------------------------------------------
// MI355X hands workgroup p of a launch to XCD p % 8, and every XCD has its own 4 MB L2.  In dispatch order = block order, each
// XCD sees queries from everywhere in the target, i.e. the whole graph (rows + points: 30-60 MB for a 200 k-point submap) streams
// through every L2.  Here the blocks are ranked by the azimuth of their middle query around the target's centre (map frame, initial
// pose) and dealt so that XCD x gets the x-th eighth of that ranking, in ascending order: each L2 serves one sector, and at any moment
// a narrow wedge of it (L2 hit rate of the correspondence launches 73 % -> 88 %).  Sectors rather than slabs:
// they are statistically alike, so the eight XCDs finish together (x- or y-slabs measured 5 % SLOWER: the outer slabs hold the
// expensive far-field blocks).  Keys by one thread per block, then one workgroup: 1024-bin counting sort in LDS; the order inside a bin is arbitrary and does not
// matter (results are addressed by block id).
constexpr int kSectorBins = 1024;
__global__ __launch_bounds__(256) void k_xcd_keys(const BlockDesc* __restrict__ blocks, int n_blocks, const Segment* __restrict__ segs,
                                                  const GridIndex* __restrict__ grids, const ItemState* __restrict__ items,
                                                  const float4* __restrict__ sorted_all, int by_target, int* __restrict__ keys)
{
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= n_blocks) return;
    const BlockDesc bd = blocks[b];
    const Segment sg = segs[bd.seg];
    if (by_target) {
        // a batch whose registrations have targets of their own (loop-closure candidate pairs): the blocks are ranked by target slot, slots
        // x, x + 8, ... first for x = 0 .. 7, so that the x-th eighth of the ranking — XCD x's share — holds whole targets: each L2 then
        // serves one target (3-6 MB with its cell table) at a time instead of a slice of every target in flight
        const int slot = sg.target >> 1;
        keys[b] = ((slot & 7) << 7) | min(slot >> 3, 127);
        return;
    }
    const GridIndex g = grids[sg.target];
    const float* M = items[bd.item].M;
    const int e = bd.start + bd.count / 2;
    const float4 q = sorted_all ? sorted_all[sg.flat_base + e] : sg.src[e];
    const float x = M[0] * q.x + M[1] * q.y + M[2] * q.z + M[3] - (g.ox + 0.5f * (float)g.nx * g.cell);
    const float y = M[4] * q.x + M[5] * q.y + M[6] * q.z + M[7] - (g.oy + 0.5f * (float)g.ny * g.cell);
    const float a = atan2f(y, x) * (0.5f * (float)kSectorBins / 3.14159265f) + 0.5f * (float)kSectorBins;
    const int k = (a == a) ? (int)a : 0;                     // NaN poses / points: bin 0
    keys[b] = min(max(k, 0), kSectorBins - 1);
}

__global__ __launch_bounds__(1024) void k_xcd_order(int n_blocks, const int* __restrict__ keys, int* __restrict__ order)
{
    __shared__ int s_cnt[kSectorBins], s_tmp[kSectorBins];
    const int tid = threadIdx.x;
    s_cnt[tid] = 0;
    __syncthreads();
    // kU independent key loads in flight per thread: a single workgroup is bound by the round trips, not by bandwidth
    constexpr int kU = 32;
    for (int base = 0; base < n_blocks; base += kU * 1024) {
        int k[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) { const int b = base + u * 1024 + tid; k[u] = b < n_blocks ? keys[b] : -1; }
#pragma unroll
        for (int u = 0; u < kU; ++u) if (k[u] >= 0) atomicAdd(&s_cnt[k[u]], 1);
    }
    __syncthreads();
    // exclusive scan of the 1024 bins: shuffles inside each of the 16 waves, the 16 wave totals through LDS (two barriers, not twenty)
    const int lane = tid & 63, wave = tid >> 6;
    const int mine = s_cnt[tid];
    int v = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(v, d); if (lane >= d) v += o; }
    if (lane == 63) s_tmp[wave] = v;
    __syncthreads();
    if (tid < 16) {
        int w = s_tmp[tid];
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) { const int o = __shfl_up(w, d); if (tid >= d) w += o; }
        s_tmp[16 + tid] = w - s_tmp[tid];                    // exclusive offset of wave `tid`
    }
    __syncthreads();
    v += s_tmp[16 + wave];
    s_cnt[tid] = v - mine;                                   // cursor = first rank of the bin
    __syncthreads();
    for (int base = 0; base < n_blocks; base += kU * 1024) {
        int k[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) { const int b = base + u * 1024 + tid; k[u] = b < n_blocks ? keys[b] : -1; }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            if (k[u] < 0) continue;
            const int r = atomicAdd(&s_cnt[k[u]], 1);
            order[r] = base + u * 1024 + tid;                // ranks of a wave's equal keys are consecutive: near-coalesced stores
        }
    }
}

#endif

}  // namespace

#if !LISREG_EXACT
void launch_xcd_order(const BlockDesc* blocks, int n_blocks, const Segment* segs, const GridIndex* grids, const ItemState* items,
                      const float4* sorted_all, bool by_target, int* keys, int* order, hipStream_t st)
{
    if (n_blocks <= 0) return;
    k_xcd_keys<<<(n_blocks + 255) / 256, 256, 0, st>>>(blocks, n_blocks, segs, grids, items, sorted_all, by_target ? 1 : 0, keys);
    k_xcd_order<<<1, 1024, 0, st>>>(n_blocks, keys, order);
}
#endif

// launch_assoc (production arithmetic) / launch_assoc_exact (the reference's arithmetic, equal distances always resolved canonically):
// same arguments, same front-ends
template <bool kTies>
static void launch_assoc_impl(const BlockDesc* blocks, int n_blocks, const Segment* segs, const GridIndex* grids,
                              const ItemState* items, DevParams prm, const float4* sorted_all, double* partials,
                              int mode, int* nn, int n_elems, float first_pass_r2, bool wide, int graph_hops,
                              unsigned long long* counters, int* dbg_nn, int lanes_q,
                              const BlockDesc* blocks_q, int n_blocks_q, float4* coef, int* coef_ok, const int* xcd_order, hipStream_t st)
{
    using namespace LISREG_ASSOC_NS;
    if (mode == 1 && lanes_q == 8) {
        const bool share = n_blocks_q <= 2048;              // <= 8192 wavefronts: two generations at four waves per SIMD
        if (wide && share)
            k_assoc_walk<true, 0, 8, kTies, true><<<n_blocks_q, kBlockQ, 0, st>>>(blocks_q, segs, grids, items, prm, sorted_all, nn, n_elems,
                                                                                      first_pass_r2, graph_hops, counters, dbg_nn, coef, coef_ok, partials, nullptr);
        else if (share)
            k_assoc_walk<false, 0, 8, kTies, true><<<n_blocks_q, kBlockQ, 0, st>>>(blocks_q, segs, grids, items, prm, sorted_all, nn, n_elems,
                                                                                       first_pass_r2, graph_hops, counters, dbg_nn, coef, coef_ok, partials, nullptr);
        else if (wide)
            k_assoc_walk<true, 0, 8, kTies><<<n_blocks_q, kBlockQ, 0, st>>>(blocks_q, segs, grids, items, prm, sorted_all, nn, n_elems,
                                                                                first_pass_r2, graph_hops, counters, dbg_nn, coef, coef_ok, partials, nullptr);
        else
            k_assoc_walk<false, 0, 8, kTies><<<n_blocks_q, kBlockQ, 0, st>>>(blocks_q, segs, grids, items, prm, sorted_all, nn, n_elems,
                                                                                 first_pass_r2, graph_hops, counters, dbg_nn, coef, coef_ok, partials, nullptr);
        k_rows_reduce<<<n_blocks, kBlockQ, 0, st>>>(blocks, segs, grids, items, prm, sorted_all, coef, coef_ok, partials);
        return;
    }
    if (mode == 0)
        k_assoc_staged<<<n_blocks, kBlockQ, 0, st>>>(blocks, segs, grids, items, prm, sorted_all, partials);
    else if (mode == 1) {
        if (wide)
            k_assoc_walk<true, 0, 1, kTies><<<n_blocks, kBlockQ, 0, st>>>(blocks, segs, grids, items, prm, sorted_all, nn, n_elems,
                                                                              first_pass_r2, graph_hops, counters, dbg_nn, coef, coef_ok, partials, xcd_order);
        else
            k_assoc_walk<false, 0, 1, kTies><<<n_blocks, kBlockQ, 0, st>>>(blocks, segs, grids, items, prm, sorted_all, nn, n_elems,
                                                                               first_pass_r2, graph_hops, counters, dbg_nn, coef, coef_ok, partials, xcd_order);
    } else if (mode == 5) {
        if (wide)
            k_assoc_walk<true, 2, 1, kTies><<<n_blocks, kBlockQ, 0, st>>>(blocks, segs, grids, items, prm, sorted_all, nn, n_elems,
                                                                          first_pass_r2, graph_hops, counters, dbg_nn, coef, coef_ok, partials, xcd_order);
        else
            k_assoc_walk<false, 2, 1, kTies><<<n_blocks, kBlockQ, 0, st>>>(blocks, segs, grids, items, prm, sorted_all, nn, n_elems,
                                                                           first_pass_r2, graph_hops, counters, dbg_nn, coef, coef_ok, partials, xcd_order);
    } else {
        if (wide)
            k_assoc_walk<true, 1, 1, kTies><<<n_blocks, kBlockQ, 0, st>>>(blocks, segs, grids, items, prm, sorted_all, nn, n_elems,
                                                                          first_pass_r2, graph_hops, counters, dbg_nn, coef, coef_ok, partials, xcd_order);
        else
            k_assoc_walk<false, 1, 1, kTies><<<n_blocks, kBlockQ, 0, st>>>(blocks, segs, grids, items, prm, sorted_all, nn, n_elems,
                                                                           first_pass_r2, graph_hops, counters, dbg_nn, coef, coef_ok, partials, xcd_order);
    }
}

// Test hook (lisreg_test_fit_models): the DEVICE functions of the two residual models on caller-given neighbourhoods, one thread per
// case — corner_model / corner_eval (kind 0: odomEstimationNode.cpp:664-739) or surf_model / surf_eval (kind 1: :776-821; in the
// production arithmetic that is plane5_closed with its hand-over to the column-pivoted QR).  out[i] = 10 floats:
//   kind 1: pa, pb, pc, pd (NaN: the plane failed the |n.p + d| <= plane_tol test), 1 = closed form used / 0 = QR, cf[0..3], accepted
//   kind 0: the line model m0.xyz (centroid), m1.xyz (direction; m1.x NaN: lambda-ratio test failed) packed as 6 floats -> out[0..5] is
//           not the reference's representation, so only cf[0..3] (out[5..8]) and `accepted` (out[9]) are compared; out[0..4] = m0.xyz, m1.x, m1.y
void LISREG_LAUNCH_TEST_FIT(int kind, int n, const float* nb15, const float* q3, DevParams prm, float* out, hipStream_t st)
{
    using namespace LISREG_ASSOC_NS;
    if (n > 0) k_test_fit_models<<<(n + 63) / 64, 64, 0, st>>>(kind, n, nb15, q3, prm, out);
}

void LISREG_LAUNCH_ASSOC(const BlockDesc* blocks, int n_blocks, const Segment* segs, const GridIndex* grids,
                         const ItemState* items, DevParams prm, const float4* sorted_all, double* partials,
                         int mode, int* nn, int n_elems, float first_pass_r2, bool wide, int graph_hops,
                         unsigned long long* counters, int* dbg_nn, int lanes_q,
                         const BlockDesc* blocks_q, int n_blocks_q, float4* coef, int* coef_ok, const int* xcd_order, hipStream_t st)
{
    if (n_blocks <= 0) return;
#if LISREG_EXACT
    launch_assoc_impl<true>(blocks, n_blocks, segs, grids, items, prm, sorted_all, partials, mode, nn, n_elems, first_pass_r2, wide, graph_hops,
                            counters, dbg_nn, lanes_q, blocks_q, n_blocks_q, coef, coef_ok, xcd_order, st);
#else
    if (prm.ties)
        launch_assoc_impl<true>(blocks, n_blocks, segs, grids, items, prm, sorted_all, partials, mode, nn, n_elems, first_pass_r2, wide, graph_hops,
                                counters, dbg_nn, lanes_q, blocks_q, n_blocks_q, coef, coef_ok, xcd_order, st);
    else
        launch_assoc_impl<false>(blocks, n_blocks, segs, grids, items, prm, sorted_all, partials, mode, nn, n_elems, first_pass_r2, wide, graph_hops,
                                 counters, dbg_nn, lanes_q, blocks_q, n_blocks_q, coef, coef_ok, xcd_order, st);
#endif
}

}  // namespace lisreg
