// lisreg_solve.hip — per-registration Gauss-Newton step, kept on the device so the GN loop never syncs the host.
//
// One 64-lane workgroup per registration:
//   * fixed-order fp64 sum of the workgroup partials written by lisreg_assoc.hip -> AtA, AtB (rounded to float
//     exactly once, like cv's CV_32F GEMM with double accumulators), n_corr;
//   * LMOptimization's tail: /root/reference/src/node/odomEstimationNode.cpp:869-872 (`< 50` no-op), :921
//     cv::solve(DECOMP_QR), :923-946 iteration-0 degeneracy analysis (cv::eigen 6x6, matV.inv()*matV2),
//     :948-953 projection incl. the local-matP shadowing quirk, :955-973 update + convergence;
//   * finalize: transformUpdate (:976-1006) and the 12-float result record.
// 6x6 algebra is tiny and strictly sequential: lane 0 runs it on LDS-resident matrices.
#include "lisreg_internal.hpp"

namespace lisreg {

namespace {

__device__ float hyp(float a, float b)
{
    a = fabsf(a); b = fabsf(b);
    if (a > b) { b /= a; return a * sqrtf(1.f + b * b); }
    if (b > 0.f) { a /= b; return b * sqrtf(1.f + a * a); }
    return 0.f;
}

// cv::eigen on a symmetric 6x6 float matrix: classical Jacobi (largest off-diagonal pivot), eigenvalues
// descending, eigenvectors in rows.  A is destroyed.
__device__ void eigen_sym6(float* A, float* W, float* V, int* indR, int* indC)
{
    const int n = 6;
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) V[i * n + j] = (i == j) ? 1.f : 0.f;
    for (int k = 0; k < n; ++k) {
        W[k] = A[k * n + k];
        if (k < n - 1) {
            int m = k + 1; float mv = fabsf(A[k * n + m]);
            for (int i = k + 2; i < n; ++i) { float v = fabsf(A[k * n + i]); if (mv < v) { mv = v; m = i; } }
            indR[k] = m;
        }
        if (k > 0) {
            int m = 0; float mv = fabsf(A[k]);
            for (int i = 1; i < k; ++i) { float v = fabsf(A[i * n + k]); if (mv < v) { mv = v; m = i; } }
            indC[k] = m;
        }
    }
    for (int it = 0; it < n * n * 30; ++it) {
        int k = 0, l; float mv = fabsf(A[indR[0]]);
        for (int i = 1; i < n - 1; ++i) { float v = fabsf(A[i * n + indR[i]]); if (mv < v) { mv = v; k = i; } }
        l = indR[k];
        for (int i = 1; i < n; ++i) { float v = fabsf(A[indC[i] * n + i]); if (mv < v) { mv = v; k = indC[i]; l = i; } }
        const float p = A[k * n + l];
        if (fabsf(p) <= 1.1920929e-7f) break;
        const float y = (W[l] - W[k]) * 0.5f;
        float t = fabsf(y) + hyp(p, y);
        float s = hyp(p, t);
        const float c = t / s;
        s = p / s; t = (p / t) * p;
        if (y < 0.f) { s = -s; t = -t; }
        A[k * n + l] = 0.f;
        W[k] -= t; W[l] += t;
#define LR(v0, v1) do { const float a0_ = (v0), b0_ = (v1); (v0) = a0_ * c - b0_ * s; (v1) = a0_ * s + b0_ * c; } while (0)
        for (int i = 0; i < k; ++i)      LR(A[i * n + k], A[i * n + l]);
        for (int i = k + 1; i < l; ++i)  LR(A[k * n + i], A[i * n + l]);
        for (int i = l + 1; i < n; ++i)  LR(A[k * n + i], A[l * n + i]);
        for (int i = 0; i < n; ++i)      LR(V[k * n + i], V[l * n + i]);
#undef LR
        for (int j = 0; j < 2; ++j) {
            const int idx = j == 0 ? k : l;
            if (idx < n - 1) {
                int m = idx + 1; float mv2 = fabsf(A[idx * n + m]);
                for (int i = idx + 2; i < n; ++i) { float v = fabsf(A[idx * n + i]); if (mv2 < v) { mv2 = v; m = i; } }
                indR[idx] = m;
            }
            if (idx > 0) {
                int m = 0; float mv2 = fabsf(A[idx]);
                for (int i = 1; i < idx; ++i) { float v = fabsf(A[i * n + idx]); if (mv2 < v) { mv2 = v; m = i; } }
                indC[idx] = m;
            }
        }
    }
    for (int k = 0; k < n - 1; ++k) {
        int m = k;
        for (int i = k + 1; i < n; ++i) if (W[m] < W[i]) m = i;
        if (m != k) {
            float tw = W[m]; W[m] = W[k]; W[k] = tw;
            for (int i = 0; i < n; ++i) { float tv = V[m * n + i]; V[m * n + i] = V[k * n + i]; V[k * n + i] = tv; }
        }
    }
}

// cv::solve(AtA, AtB, X, DECOMP_QR): Householder QR in float; x zeroed and false returned if singular.  Everything in registers (all
// loops fully unrolled, compile-time indices): the per-iteration critical path of a single registration is this solve.
// One COLUMN of [A | b] per lane (lanes 0..5 hold column j: a[i] = A[i][j]; lane 6 holds b; the other lanes idle
// along): at step k lane k forms the Householder vector of its own column (tail sum in ascending row order, beta, v, tau as cv does), tau and v travel
// by readlane, and every later column is updated by its own lane — the same operations in the same order as the one-lane form this replaces
// (rounds 2-5: the same bits, checked by the exact build's sweeps against the oracle), a sixth of the instructions on the critical path of a
// Gauss-Newton iteration (the step is one wavefront's serial latency between two correspondence launches).
// The triangular solve runs on values read out of the lanes (uniform), so every lane ends with the same x.
__device__ __forceinline__ float lane_value(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ bool solve6_qr_lanes(float a[6], int lane, float x[6])
{
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const float c0 = a[k];
        float tail = 0.f;
#pragma unroll
        for (int i = k + 1; i < 6; ++i) tail += a[i] * a[i];
        float beta, tau, v[6];
        if (tail <= 1.17549435e-38f) {
            tau = 0.f; beta = c0;
#pragma unroll
            for (int i = 0; i < 6; ++i) v[i] = 0.f;
        } else {
            beta = sqrtf(c0 * c0 + tail);
            if (c0 >= 0.f) beta = -beta;
#pragma unroll
            for (int i = 0; i < 6; ++i) v[i] = (i > k) ? a[i] / (c0 - beta) : 0.f;
            tau = (beta - c0) / beta;
        }
        if (lane == k) a[k] = beta;
        tau = lane_value(tau, k);
#pragma unroll
        for (int i = k + 1; i < 6; ++i) v[i] = lane_value(v[i], k);
        if (tau != 0.f && lane > k) {                            // columns k + 1 .. 5 and the right-hand side
            float dot = a[k];
#pragma unroll
            for (int i = k + 1; i < 6; ++i) dot += v[i] * a[i];
            dot *= tau;
            a[k] -= dot;
#pragma unroll
            for (int i = k + 1; i < 6; ++i) a[i] -= dot * v[i];
        }
    }
    float R[6][6], c[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        c[i] = lane_value(a[i], 6);
#pragma unroll
        for (int j = i; j < 6; ++j) R[i][j] = lane_value(a[i], j);
    }
    bool singular = false;
#pragma unroll
    for (int i = 0; i < 6; ++i) singular = singular || (fabsf(R[i][i]) <= 1.17549435e-38f);
    float y[6];
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        float s_ = c[i];
#pragma unroll
        for (int j = i + 1; j < 6; ++j) s_ -= R[i][j] * y[j];
        y[i] = s_ / R[i][i];
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) x[i] = singular ? 0.f : y[i];
    return !singular;
}

// cv::Mat::inv() (DECOMP_LU): LU with partial pivoting.  A destroyed, B receives the inverse (zeros if singular).
__device__ void inv6_lu(float* A, float* B)
{
    const int N = 6;
    for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) B[i * N + j] = (i == j) ? 1.f : 0.f;
    for (int i = 0; i < N; ++i) {
        int k = i;
        for (int j = i + 1; j < N; ++j) if (fabsf(A[j * N + i]) > fabsf(A[k * N + i])) k = j;
        if (fabsf(A[k * N + i]) < 1.1920929e-7f * 100) { for (int j = 0; j < N * N; ++j) B[j] = 0.f; return; }
        if (k != i) for (int j = 0; j < N; ++j) {
            float t = A[i * N + j]; A[i * N + j] = A[k * N + j]; A[k * N + j] = t;
            t = B[i * N + j]; B[i * N + j] = B[k * N + j]; B[k * N + j] = t;
        }
        const float d = -1.f / A[i * N + i];
        for (int j = i + 1; j < N; ++j) {
            const float alpha = A[j * N + i] * d;
            for (int m = i + 1; m < N; ++m) A[j * N + m] += alpha * A[i * N + m];
            for (int m = 0; m < N; ++m) B[j * N + m] += alpha * B[i * N + m];
        }
    }
    for (int i = N - 1; i >= 0; --i)
        for (int j = 0; j < N; ++j) {
            float s = B[i * N + j];
            for (int k = i + 1; k < N; ++k) s -= A[i * N + k] * B[k * N + j];
            B[i * N + j] = s / A[i * N + i];
        }
}

// pcl::getTransformation via trans2Affine3f (common.cpp:54-57) and LMOptimization's sin/cos (:862-867), computed
// once per registration per iteration here instead of once per thread in the correspondence kernel.
// trig = {cos yaw, sin yaw, cos pitch, sin pitch, cos roll, sin roll} computed elsewhere (exact build: by the HOST's libm, see
// launch_pose_cache_host_trig), or null: the device's own cosf / sinf
__device__ void write_pose_cache(ItemState* it, const float* trig = nullptr)
{
    const float* T = it->T;
    float A, B, C, D, E, F;
    if (trig) { A = trig[0]; B = trig[1]; C = trig[2]; D = trig[3]; E = trig[4]; F = trig[5]; }
    else { A = cosf(T[2]); B = sinf(T[2]); C = cosf(T[1]); D = sinf(T[1]); E = cosf(T[0]); F = sinf(T[0]); }
    const float DE = D * E, DF = D * F;
    float* M = it->M;
    M[0] = A * C;  M[1] = A * DF - B * E;  M[2]  = B * F + A * DE;  M[3]  = T[3];
    M[4] = B * C;  M[5] = A * E + B * DF;  M[6]  = B * DE - A * F;  M[7]  = T[4];
    M[8] = -D;     M[9] = C * F;           M[10] = C * E;           M[11] = T[5];
    it->sc[0] = D; it->sc[1] = C;      // srx = sin(pitch), crx = cos(pitch)
    it->sc[2] = B; it->sc[3] = A;      // sry = sin(yaw),   cry = cos(yaw)
    it->sc[4] = F; it->sc[5] = E;      // srz = sin(roll),  crz = cos(roll)
    // LMOptimization :898-907: every coefficient of pointOri.{x,y,z} in arx / ary / arz depends on the pose only.  They are formed
    // here once per registration and iteration with the reference's own association ((a*b)*c ...), so that the correspondence kernel
    // multiplies each by the point coordinate exactly where the reference does — instead of 64 lanes redoing ~45 uniform products.
    const float srx = D, crx = C, sry = B, cry = A, srz = F, crz = E;
    float* K = it->jk;
    K[0] = crx * sry * srz;  K[1] = crx * crz * sry;  K[2] = srx * sry;
    K[3] = -srx * srz;       K[4] = crz * srx;        K[5] = crx;
    K[6] = crx * cry * srz;  K[7] = crx * cry * crz;  K[8] = cry * srx;
    K[9] = cry * srx * srz - crz * sry;    K[10] = sry * srz + cry * crz * srx;   K[11] = crx * cry;
    K[12] = -cry * crz - srx * sry * srz;  K[13] = cry * srz - crz * srx * sry;   K[14] = crx * sry;
    K[15] = crz * srx * sry - cry * srz;   K[16] = crx * crz;                      K[17] = crx * srz;
    K[18] = crz * sry - cry * srx * srz;   K[19] = 0.f;                            K[20] = 0.f;
}

// One registration back to the start of a run (member initialisers :70-71, guard :598, the pose cache of the initial pose).  The done
// counter's start value is the number of registrations that fail the guard — a constant of the prepared batch the host knows
// (DevParams::n_guard_failed) — written by ONE thread: no memset launch in front, no atomics here.
__device__ __forceinline__ void reset_item(ItemState* it, const DevParams& P)
{
    for (int k = 0; k < 6; ++k) it->T[k] = it->T_init[k];
    for (int k = 0; k < 36; ++k) it->P[k] = 0.f;
    const bool guard_ok = (it->n_sc > P.edge_min) && (it->n_ss > P.surf_min);   // odomEstimationNode.cpp:598
    it->iter = 0;
    it->guard_failed = guard_ok ? 0 : 1;
    it->done = guard_ok ? 0 : 1;
    it->iters_out = 0;
    it->deltaR = 100.f; it->deltaT = 100.f;                                      // member initialisers :70-71
    it->degenerate = it->degenerate_in;
    it->n_corr = 0;
    it->any_solved = 0;
    write_pose_cache(it);
}

__global__ __launch_bounds__(64) void k_reset_items(ItemState* __restrict__ items, int n_items, const DevParams P,
                                                    int* __restrict__ done_counter, int* __restrict__ zero_too)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i == 0) { *done_counter = P.n_guard_failed; if (zero_too) *zero_too = 0; }      // (zero_too: the batch's count of queries in cells left without rows, "row_reach")
    if (i >= n_items) return;
    reset_item(&items[i], P);
}

constexpr int kSolveThreads = 512;

__global__ __launch_bounds__(kSolveThreads) void k_solve(ItemState* __restrict__ items, const DevParams P,
                                                         const double* __restrict__ partials,
                                                         float* __restrict__ trace, int trace_cap,
                                                         int* __restrict__ done_counter)
{
    __shared__ float s_AtA[36], s_AtB[6], s_X[6], s_A[36];
    __shared__ float s_E[6], s_V[36], s_V2[36], s_Vi[36];
    __shared__ int   s_ind[12];
    __shared__ double s_part[kSolveThreads / 32][32];
    __shared__ double s_sum[kNumAcc];
    __shared__ double s_L[36];

    ItemState* it = &items[blockIdx.x];
    if (it->done) return;
    // fixed-order fp64 sum of this registration's workgroup partials: 8 row groups x 28 columns in parallel,
    // group g takes rows g, g+8, ... ; groups are then combined in index order (deterministic).
    {
        const int col = threadIdx.x & 31, grp = threadIdx.x >> 5;
        double s = 0.0;
        if (col < kNumAcc) {
            const double* p = partials + (size_t)it->blk_begin * kNumAcc + col;
            const int nb = it->blk_count;
            // the rows of a group are added in ascending order whatever the unrolling.  A launch of this kernel is a chain of load round
            // trips on one workgroup per registration (450 rows / 16 groups = 28 loads per thread): up to 32 rows are asked for at once — rows
            // past the end read as +0.0, which leaves a sum that started at +0.0 unchanged to the bit — so that the usual registration is ONE
            // round trip (it was four: chunks of 16 + 4 + 4 + 4)
            constexpr int kG = kSolveThreads / 32;
            for (int b = grp; b < nb; b += 32 * kG) {
                double v[32];
#pragma unroll
                for (int u = 0; u < 32; ++u) v[u] = (b + u * kG < nb) ? p[(size_t)(b + u * kG) * kNumAcc] : 0.0;
#pragma unroll
                for (int u = 0; u < 32; ++u) s += v[u];
            }
        }
        s_part[grp][col] = s;
    }
    __syncthreads();
    if (threadIdx.x < kNumAcc) {
        double s = 0.0;
#pragma unroll
        for (int g = 0; g < kSolveThreads / 32; ++g) s += s_part[g][threadIdx.x];
        s_sum[threadIdx.x] = s;
    }
    __syncthreads();
    // the rest is one wavefront: lane-parallel where the arithmetic allows (the matrix fill, the QR solve, the three sine / cosine pairs of
    // the pose cache), lane 0 for the strictly sequential pieces
    if (threadIdx.x >= 64) return;
    const int lane = threadIdx.x;
    const bool l0 = lane == 0;

    const int iter = it->iter;
    const int n_sel = (int)(s_sum[27] + 0.5);
    float* tr = (trace && iter < trace_cap) ? trace + ((size_t)blockIdx.x * trace_cap + iter) * kTraceStride : nullptr;
    if (l0) {
        it->n_corr = n_sel;
        if (tr) {
            for (int k = 0; k < kTraceStride; ++k) tr[k] = 0.f;
            tr[0] = (float)n_sel;
            for (int k = 0; k < 6; ++k) tr[49 + k] = it->T[k];
        }
    }
    bool finished = false;
    if (n_sel >= P.min_corr) {                                   // :870-872 (uniform: n_sel comes out of LDS)
        // packed upper triangle -> the symmetric matrix: entry (r, c) of the 36 by lane, the right-hand side by the next six
        if (lane < 36) {
            const int r_ = lane / 6, c_ = lane % 6, lo = min(r_, c_), hi = max(r_, c_);
            s_AtA[lane] = (float)s_sum[lo * 6 - (lo * (lo - 1)) / 2 + (hi - lo)];
        } else if (lane < 42) s_AtB[lane - 36] = (float)s_sum[21 + lane - 36];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        {
            float a[6], xs[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) a[i] = lane < 6 ? s_AtA[i * 6 + lane] : (lane == 6 ? s_AtB[i] : 0.f);
            solve6_qr_lanes(a, lane, xs);                        // :921
            if (l0) {
#pragma unroll
                for (int i = 0; i < 6; ++i) s_X[i] = xs[i];
            }
        }
        int isDeg = 0;
        float Tn[6] = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f };
        if (l0) {
            isDeg = it->degenerate;
            // Shortcut for the common, well-conditioned case: if AtA - shift * I is positive definite (6x6 Cholesky in double,
            // < 1 us on one lane) every eigenvalue exceeds shift = eig_thresh + 2e-5 * trace, far enough above the threshold that
            // cv::eigen's float Jacobi (absolute error ~ 1e-7 * norm) cannot report one below it: not degenerate, matP = V^-1 V = I.
            // Anything closer to the threshold takes the full restatement of cv::eigen below (78 us at this size).
            // (the factor lives in LDS: as a private array under these data-dependent loops it went to scratch memory, and the first
            //  iteration's step took 26 us instead of 10)
            bool well_conditioned = false;
            if (iter == 0) {
                double* L = s_L;
                double trace = 0.0;
                for (int i = 0; i < 6; ++i) trace += (double)s_AtA[i * 6 + i];
                const double shift = (double)P.eig_thresh + 2e-5 * trace;
                well_conditioned = trace > 0.0;
                for (int r = 0; r < 6 && well_conditioned; ++r)
                    for (int c = 0; c <= r; ++c) {
                        double v = (double)s_AtA[r * 6 + c] - (r == c ? shift : 0.0);
                        for (int k = 0; k < c; ++k) v -= L[r * 6 + k] * L[c * 6 + k];
                        if (r == c) { if (!(v > 0.0)) { well_conditioned = false; break; } L[r * 6 + r] = sqrt(v); }
                        else L[r * 6 + c] = v / L[c * 6 + c];
                    }
            }
            if (iter == 0 && well_conditioned) {
                isDeg = 0;
                for (int i = 0; i < 36; ++i) it->P[i] = (i % 7 == 0) ? 1.f : 0.f;
            } else if (iter == 0) {                                  // :923-946
                for (int i = 0; i < 36; ++i) s_A[i] = s_AtA[i];
                eigen_sym6(s_A, s_E, s_V, s_ind, s_ind + 6);
                for (int i = 0; i < 36; ++i) s_V2[i] = s_V[i];
                isDeg = 0;
                for (int i = 5; i >= 0; --i) {
                    if (s_E[i] < P.eig_thresh) { for (int j = 0; j < 6; ++j) s_V2[i * 6 + j] = 0.f; isDeg = 1; }
                    else break;
                }
                for (int i = 0; i < 36; ++i) s_A[i] = s_V[i];
                inv6_lu(s_A, s_Vi);
                for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) {
                    double s = 0; for (int m = 0; m < 6; ++m) s += (double)s_Vi[r * 6 + m] * (double)s_V2[m * 6 + c];
                    it->P[r * 6 + c] = (float)s;
                }
            } else if (P.emulate_shadow) {
                for (int i = 0; i < 36; ++i) it->P[i] = 0.f;         // local zero cv::Mat matP (:880)
            }
            if (isDeg) {                                             // :948-953
                float X2[6];
                for (int r = 0; r < 6; ++r) X2[r] = s_X[r];
                for (int r = 0; r < 6; ++r) {
                    double s = 0; for (int m = 0; m < 6; ++m) s += (double)it->P[r * 6 + m] * (double)X2[m];
                    s_X[r] = (float)s;
                }
            }
            if (!P.freeze_pose) {
#pragma unroll
                for (int m = 0; m < 6; ++m) { Tn[m] = it->T[m] + s_X[m]; it->T[m] = Tn[m]; }      // :955-960
            }
        }
        if (!P.freeze_pose) {
            // the pose cache's three sine / cosine pairs (yaw, pitch, roll) on three lanes at once; lane 0 assembles the cache from them
            const float ang = lane == 0 ? lane_value(Tn[2], 0) : (lane == 1 ? lane_value(Tn[1], 0) : lane_value(Tn[0], 0));
            const float cs = cosf(ang), sn = sinf(ang);
            const float trig[6] = { lane_value(cs, 0), lane_value(sn, 0), lane_value(cs, 1), lane_value(sn, 1), lane_value(cs, 2), lane_value(sn, 2) };
            if (l0) write_pose_cache(it, trig);
        }
        if (l0) {
            const double r0 = (double)(s_X[0] * 57.29578f), r1 = (double)(s_X[1] * 57.29578f), r2 = (double)(s_X[2] * 57.29578f);
            const double t0 = (double)(s_X[3] * 100.f), t1 = (double)(s_X[4] * 100.f), t2 = (double)(s_X[5] * 100.f);
            const float dR = (float)sqrt(r0 * r0 + r1 * r1 + r2 * r2);
            const float dT = (float)sqrt(t0 * t0 + t1 * t1 + t2 * t2);
            it->deltaR = dR; it->deltaT = dT;
            it->degenerate = isDeg;
            it->any_solved = 1;
            if (tr) {
                for (int i = 0; i < 36; ++i) tr[1 + i] = s_AtA[i];
                for (int i = 0; i < 6; ++i) { tr[37 + i] = s_AtB[i]; tr[43 + i] = s_X[i]; tr[49 + i] = it->T[i]; }
                tr[55] = 1.f;
            }
            if (dR < P.conv_deg && dT < P.conv_cm && P.fixed_iters <= 0) {   // :969-972 -> break at :617
                it->iters_out = iter;
                finished = true;
            }
        }
    }
    if (l0) {
        it->iter = iter + 1;
        if (!finished && iter + 1 >= P.bound) { it->iters_out = P.bound; finished = true; }
        if (finished) { it->done = 1; atomicAdd(done_counter, 1); }     // host-side early stop of the launch loop
    }
}

// tf::Quaternion / tf::Matrix3x3 pieces of transformUpdate (odomEstimationNode.cpp:976-1006), double like tf
struct Quat { double x, y, z, w; };
__device__ Quat q_rpy(double roll, double pitch, double yaw)
{
    const double hy = yaw * 0.5, hp = pitch * 0.5, hr = roll * 0.5;
    const double cy = cos(hy), sy = sin(hy), cp = cos(hp), sp = sin(hp), cr = cos(hr), sr = sin(hr);
    return Quat{ sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy,
                 cr * cp * cy + sr * sp * sy };
}
__device__ double q_dot(Quat a, Quat b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
__device__ Quat q_slerp(Quat a, Quat b, double t)
{
    const double s = sqrt(q_dot(a, a) * q_dot(b, b));
    const double d = q_dot(a, b);
    const double theta = (d < 0 ? acos(-d / s) * 2.0 : acos(d / s) * 2.0) / 2.0;
    if (theta != 0.0) {
        const double dd = 1.0 / sin(theta), s0 = sin((1.0 - t) * theta), s1 = sin(t * theta);
        const double sg = d < 0 ? -1.0 : 1.0;
        return Quat{ (a.x * s0 + sg * b.x * s1) * dd, (a.y * s0 + sg * b.y * s1) * dd,
                     (a.z * s0 + sg * b.z * s1) * dd, (a.w * s0 + sg * b.w * s1) * dd };
    }
    return a;
}
__device__ void q_get_rpy(Quat q, double& roll, double& pitch, double& yaw)
{
    const double d = q_dot(q, q), s = 2.0 / d;
    const double xs = q.x * s, ys = q.y * s, zs = q.z * s;
    const double wx = q.w * xs, wy = q.w * ys, wz = q.w * zs;
    const double xx = q.x * xs, xy = q.x * ys, xz = q.x * zs, yy = q.y * ys, yz = q.y * zs, zz = q.z * zs;
    const double m00 = 1.0 - (yy + zz), m10 = xy + wz, m20 = xz - wy, m21 = yz + wx, m22 = 1.0 - (xx + yy);
    if (fabs(m20) >= 1) {
        yaw = 0;
        roll = atan2(m21, m22);
        pitch = m20 < 0 ? 1.5707963267948966 : -1.5707963267948966;
    } else {
        pitch = -asin(m20);
        const double cp = cos(pitch);
        roll = atan2(m21 / cp, m22 / cp);
        yaw = atan2(m10 / cp, m00 / cp);
    }
}
__device__ float clampf(float v, float lim) { v = v < -lim ? -lim : v; return v > lim ? lim : v; }

__global__ __launch_bounds__(64) void k_finalize(ItemState* __restrict__ items, int n_items, const DevParams P,
                                                 float* __restrict__ results, int* __restrict__ reset_done_counter)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i == 0 && reset_done_counter) *reset_done_counter = P.n_guard_failed;      // (no solve is in flight any more: nobody else touches the counter)
    if (i >= n_items) return;
    ItemState* it = &items[i];
    float T[6];
    for (int k = 0; k < 6; ++k) T[k] = it->T[k];
    int status = LISREG_OK;
    if (it->guard_failed) status = LISREG_NOT_ENOUGH_FEATURES;          // :623-625, T untouched
    else {
        if (P.use_imu && it->imu.imu_available && fabsf(it->imu.imu_pitch_init) < 1.4f) {
            const double w = (double)P.imu_w;
            double r, p, y;
            q_get_rpy(q_slerp(q_rpy((double)T[0], 0, 0), q_rpy((double)it->imu.imu_roll_init, 0, 0), w), r, p, y);
            T[0] = (float)r;
            q_get_rpy(q_slerp(q_rpy(0, (double)T[1], 0), q_rpy(0, (double)it->imu.imu_pitch_init, 0), w), r, p, y);
            T[1] = (float)p;
        }
        T[0] = clampf(T[0], P.rot_tol);                                   // constraintTransformation, common.cpp:285
        T[1] = clampf(T[1], P.rot_tol);
        T[5] = clampf(T[5], P.z_tol);
        if (!it->any_solved) status = LISREG_TOO_FEW_CORRESPONDENCES;
    }
    float* r = results + (size_t)i * kResultSize;
    for (int k = 0; k < 6; ++k) r[k] = T[k];
    r[6] = (float)it->iters_out; r[7] = it->deltaR; r[8] = it->deltaT;
    r[9] = (float)it->degenerate; r[10] = (float)it->n_corr; r[11] = (float)status;
    // round 6: the run leaves its registrations reset for the next run of the prepared batch (what the head of every run used to do with a
    // memset and a launch of its own on the critical path of the index phase — or behind an event hop on a side stream)
    if (reset_done_counter) reset_item(it, P);
}

// exact build: the poses out, and the pose caches rebuilt from trig values the host computed (six per registration)
__global__ __launch_bounds__(64) void k_pose_gather(const ItemState* __restrict__ items, int n_items, float* __restrict__ T_out)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n_items) return;
    for (int k = 0; k < 6; ++k) T_out[6 * i + k] = items[i].T[k];
}
__global__ __launch_bounds__(64) void k_pose_cache_from_trig(ItemState* __restrict__ items, int n_items, const float* __restrict__ trig)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n_items) return;
    write_pose_cache(&items[i], trig + 6 * i);
}

}  // namespace

void launch_pose_gather(const ItemState* items, int n_items, float* T_out, hipStream_t st)
{
    if (n_items > 0) k_pose_gather<<<(n_items + 63) / 64, 64, 0, st>>>(items, n_items, T_out);
}
void launch_pose_cache_from_trig(ItemState* items, int n_items, const float* trig, hipStream_t st)
{
    if (n_items > 0) k_pose_cache_from_trig<<<(n_items + 63) / 64, 64, 0, st>>>(items, n_items, trig);
}

void launch_reset_items(ItemState* items, int n_items, DevParams prm, int* done_counter, hipStream_t st, int* zero_too)
{
    k_reset_items<<<(std::max(n_items, 1) + 63) / 64, 64, 0, st>>>(items, n_items, prm, done_counter, zero_too);
}

void launch_solve(ItemState* items, int n_items, DevParams prm, const double* partials, float* trace,
                  int trace_cap, int* done_counter, hipStream_t st)
{
    if (n_items > 0) k_solve<<<n_items, kSolveThreads, 0, st>>>(items, prm, partials, trace, trace_cap, done_counter);
}

void launch_finalize(ItemState* items, int n_items, DevParams prm, float* results, hipStream_t st, int* reset_done_counter)
{
    if (n_items > 0) k_finalize<<<(n_items + 63) / 64, 64, 0, st>>>(items, n_items, prm, results, reset_done_counter);
}

}  // namespace lisreg
